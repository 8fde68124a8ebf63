"""Keyframe sharding over GPUs (north_star: "keyframes shard naturally across the 8 GPUs of one
node with an RCCL all-gather over xGMI to assemble the final maps").

One process per GPU.  The session maps are replicated (a 10 M point map is 160 MB of 288 GB); every
per-keyframe stage runs on this rank's contiguous block of keyframes:

  visibility vote   -> MAX all-reduce of the M-byte label mask (the union of Removerter.cpp:589-590), after which
                       every rank performs the same deterministic partition + voxel grid (replicated, no broadcast)
  reprojection/kNN  -> the per-keyframe clouds STAY on the rank that produced them (LazyScans) as long as the next
                       consumer is another per-keyframe stage of the same keyframes (kNN on the reprojected scans,
                       votes that use them as source scans, the scan-wise merge + voxel grid); they are all-gathered
                       (sizes, then padded payload; reassembled in keyframe order) only when a stage needs every
                       keyframe: merging scans into a global map, and the final outputs.

  voxel grids of    -> the octree's Morton key space is cut into `world` contiguous ranges of equal point count; every
  (replicated) maps    rank sorts/reduces its range (ltm_voxel_centroid_shard) and the centroid lists are all-gathered in
                       rank order, which IS the single-GPU output (small clouds stay replicated).

Everything else (merge, the tiny weak->strong ND split) is replicated.  The collectives are
torch.distributed calls (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests), so the same code is exercised
on CPU with world_size 2.  `ops` is any object with the stage interface of removerter.HipOps.
"""
import os

import numpy as np
import torch


def shard_range(n, rank, world):
    """contiguous block of keyframes [kb, ke) of rank `rank`"""
    return (n * rank) // world, (n * (rank + 1)) // world


class LazyScans:
    """a scan set of n keyframes of which this rank holds [kb, ke); `full()` all-gathers it once"""

    def __init__(self, sops, local, kb, ke, n):
        self.sops, self.local, self.kb, self.ke, self.n = sops, local, kb, ke, n
        self._full = None

    def full(self):
        if self._full is None:
            self._full = self.sops._allgather_scanset(self.local, self.n)
        return self._full

    # so that code written for plain scan sets (tests, scan_outputs) keeps working
    def download(self):
        return self.full().download()

    def info(self):
        return self.full().info()


class ShardedOps:
    # Clouds below this many points are voxelised on every rank (replicated) instead of sharded + all-gathered.  The exchange moves
    # the whole OUTPUT to every rank (a 6.8 M-point map: 109 MB, ~0.5 ms over xGMI) while sharding saves only the sort of the input
    # (~0.35 ms for 6.8 M points on one MI355X), so it pays only for the big merges of makeGlobalMap (23 M points in, 6.8 M out).
    VOXEL_SHARD_MIN = 1 << 24

    # Step 1 (merge + grid, remove / revert passes, HD kNN) of the two sessions is independent work on different data (Removerter.cpp:1584-1591
    # runs one after the other): with an even world the even ranks take the central session and the odd ranks the query session, each group
    # sharding ITS session's keyframes world/2 ways.  A rank then does the replicated part (partitions, re-grids, host round trips) of one
    # session instead of two and its label all-reduces span half the ranks; the price is one swap of the finished maps between rank pairs.
    # LTM_SESSION_GROUPS (the switch the C++ host reads as well, Comm.cpp): "0" off, "1" on; unset = on for every backend that has run it under test
    # (gloo worlds 2/4/6, logical ranks) and OFF over RCCL ("nccl"), whose group + pair-swap path no multi-GPU hardware has executed yet
    SESSION_GROUPS = os.environ.get("LTM_SESSION_GROUPS")

    def _session_groups_on(self):
        if self.SESSION_GROUPS is not None:
            return str(self.SESSION_GROUPS) not in ("0", "False")
        try:
            return self.dist.get_backend() != "nccl"
        except Exception:
            return True

    def __init__(self, ops, dist, rank, world, group=None, peer=None):
        self.ops, self.dist, self.rank, self.world, self.group = ops, dist, rank, world, group
        self.peer = peer                   # global rank of this rank's partner in the other session group (set on group ops only)
        self._pose_slices = {}
        self._session_groups = None

    def session_groups(self):
        """(ops bound to this rank's session group, 0 = central / 1 = query) or None when the world does not split"""
        if self.group is not None or self.world < 2 or self.world % 2 or not self._session_groups_on():
            return None
        if self._session_groups is None:     # every rank creates both groups, in the same order (torch.distributed.new_group is collective)
            pgs = [self.dist.new_group(ranks=list(range(g, self.world, 2))) for g in (0, 1)]
            g = self.rank % 2
            gops = ShardedOps(self.ops, self.dist, self.rank // 2, self.world // 2, group=pgs[g], peer=self.rank ^ 1)
            gops.VOXEL_SHARD_MIN = self.VOXEL_SHARD_MIN
            self._session_groups = (gops, g)
        return self._session_groups

    def swap_clouds_with_peer(self, clouds):
        """this rank's clouds go to its partner rank of the other session group and the partner's come back, in list order: one message of sizes
        and one of points each way (rank pairs 2i <-> 2i+1 sit on different xGMI links, all pairs swap at the same time)"""
        assert self.peer is not None, "swap_clouds_with_peer is an operation of session-group ops"
        ts = [self.ops.cloud_to_tensor(c) for c in clouds]
        dev = ts[0].device
        mine_n = torch.tensor([t.shape[0] for t in ts], dtype=torch.int64, device=dev)
        their_n = torch.empty_like(mine_n)
        send = torch.cat(ts).contiguous() if int(mine_n.sum()) else torch.zeros((0, 4), dtype=torch.float32, device=dev)

        def swap(out, inp):
            if self.dist.get_rank() < self.peer:
                self.dist.send(inp, self.peer); self.dist.recv(out, self.peer)
            else:
                self.dist.recv(out, self.peer); self.dist.send(inp, self.peer)
        swap(their_n, mine_n)
        rows = [int(x) for x in their_n.cpu().tolist()]
        recv = torch.empty((sum(rows), 4), dtype=torch.float32, device=dev)
        if sum(rows) or send.shape[0]:
            # both sides take part as soon as either has a point (a zero-row message is still a message)
            swap(recv, send)
        out, at = [], 0
        for n in rows:
            out.append(self.ops.cloud_from_tensor(recv[at:at + n].contiguous()) if n else self.ops.empty_cloud())
            at += n
        return out

    def __getattr__(self, name):           # replicated stages are forwarded untouched
        return getattr(self.ops, name)

    # ---- helpers
    def _range(self, poses):
        return shard_range(self.ops.n_keyframes(poses), self.rank, self.world)

    def _local_poses(self, poses):
        kb, ke = self._range(poses)
        key = (id(poses), kb, ke)
        if key not in self._pose_slices:
            self._pose_slices[key] = (poses, self.ops.poses_slice(poses, kb, ke))     # keep `poses` alive with its slice
        return self._pose_slices[key][1]

    def _check(self, scans, poses):
        kb, ke = self._range(poses)
        assert (scans.kb, scans.ke, scans.n) == (kb, ke, self.ops.n_keyframes(poses)), "scan shard and pose shard disagree"
        return kb, ke

    def _allgather_scanset(self, local, n_total):
        pts, off = self.ops.scanset_to_tensors(local)
        # per-keyframe offsets of every rank as one padded int64 all-gather (all_gather_object would pickle through the host)
        n_local = len(off) - 1
        n_max = max(shard_range(n_total, r, self.world)[1] - shard_range(n_total, r, self.world)[0] for r in range(self.world))
        assert n_local <= n_max, "scan shard larger than any keyframe block"
        mine = torch.full((n_max + 2,), -1, dtype=torch.int64, device=pts.device)
        mine[0] = n_local
        mine[1: n_local + 2] = torch.as_tensor([int(x) for x in off], dtype=torch.int64)
        gathered = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(gathered, mine, group=self.group)
        counts = []
        for h in torch.stack(gathered).cpu().tolist():      # one device->host copy for all ranks' tables
            counts.append(h[1: int(h[0]) + 2])
        sizes = [c[-1] for c in counts]
        cap = max(max(sizes), 1)
        pad = torch.zeros((cap, 4), dtype=torch.float32, device=pts.device)
        pad[: pts.shape[0]] = pts
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad, group=self.group)
        parts = [self.ops.scanset_from_tensors(bufs[r][: sizes[r]].contiguous(), counts[r]) for r in range(self.world)]
        return self.ops.concat_scansets(parts)

    def _allgather_cloud(self, local):
        """concatenation, in rank order, of every rank's cloud (sizes first, then one padded all-gather)"""
        t = self.ops.cloud_to_tensor(local)
        mine = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        sizes = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(sizes, mine, group=self.group)
        sizes = [int(x) for x in torch.cat(sizes).cpu().tolist()]
        cap = max(max(sizes), 1)
        pad = torch.zeros((cap, 4), dtype=torch.float32, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad, group=self.group)
        return self.ops.cloud_from_tensor(torch.cat([bufs[r][: sizes[r]] for r in range(self.world)]).contiguous())

    # ---- voxel grid of a (replicated) cloud: each rank sorts and reduces one contiguous range of Morton keys
    def voxel(self, c, leaf):
        if self.world == 1 or self.ops.size(c) < self.VOXEL_SHARD_MIN:
            return self.ops.voxel(c, leaf)
        return self._allgather_cloud(self.ops.voxel_shard(c, leaf, self.rank, self.world))

    def voxel_batch(self, clouds, leaf):
        if self.world == 1:
            return self.ops.voxel_batch(clouds, leaf)
        out = [None] * len(clouds)
        small = [i for i, c in enumerate(clouds) if self.ops.size(c) < self.VOXEL_SHARD_MIN]
        for i, r in zip(small, self.ops.voxel_batch([clouds[i] for i in small], leaf) if small else []):
            out[i] = r
        for i, c in enumerate(clouds):
            if out[i] is None:
                out[i] = self.voxel(c, leaf)
        return out

    def materialize(self, scans):
        """full scan set on every rank (used for the final per-keyframe outputs)"""
        return scans.full() if isinstance(scans, LazyScans) else scans

    # ---- vote: local keyframes, label union across ranks, replicated partition
    def vote_partition(self, cmap, scans, poses, alpha, thr, mode):
        labels = self.ops.new_labels(self.ops.size(cmap))
        if isinstance(scans, LazyScans):
            kb, ke = self._check(scans, poses)
            self.ops.vote(cmap, scans.local, self._local_poses(poses), 0, ke - kb, alpha, thr, mode, labels)
        else:
            kb, ke = self._range(poses)
            self.ops.vote(cmap, scans, poses, kb, ke, alpha, thr, mode, labels)
        if labels.numel():
            self.dist.all_reduce(labels, op=self.dist.ReduceOp.MAX, group=self.group)
        return self.ops.partition(cmap, labels)

    # ---- per-keyframe stages: results stay rank-local
    def reproject(self, cmap, poses, alpha):
        kb, ke = self._range(poses)
        return LazyScans(self, self.ops.reproject_range(cmap, poses, alpha, kb, ke), kb, ke, self.ops.n_keyframes(poses))

    def knn_partition(self, target, scans, poses, k, thr):
        n = self.ops.n_keyframes(poses)
        if isinstance(scans, LazyScans):
            kb, ke = self._check(scans, poses)
            co, di = self.ops.knn_partition_range(target, scans.local, self._local_poses(poses), k, thr, 0, ke - kb)
        else:
            kb, ke = self._range(poses)
            co, di = self.ops.knn_partition_range(target, scans, poses, k, thr, kb, ke)
        return LazyScans(self, co, kb, ke, n), LazyScans(self, di, kb, ke, n)

    def zip_concat(self, a, b, c):
        if all(isinstance(x, LazyScans) for x in (a, b, c) if x is not None):
            loc = self.ops.zip_concat(a.local, b.local, c.local if c is not None else None)
            return LazyScans(self, loc, a.kb, a.ke, a.n)
        return self.ops.zip_concat(self.materialize(a), self.materialize(b), self.materialize(c) if c is not None else None)

    def voxel_scanset(self, s, leaf):
        if isinstance(s, LazyScans):
            return LazyScans(self, self.ops.voxel_scanset(s.local, leaf), s.kb, s.ke, s.n)
        return self.ops.voxel_scanset(s, leaf)

    supports_deferred_grid = False      # (cascade.run_cascade: the two-halves hand-over is a one-context path)

    def voxel_grid_scanset(self, s, leaf):      # per-keyframe, like voxel_scanset
        if isinstance(s, LazyScans):
            return LazyScans(self, self.ops.voxel_grid_scanset(s.local, leaf), s.kb, s.ke, s.n)
        return self.ops.voxel_grid_scanset(s, leaf)

    def preclean(self, s, radius):
        if isinstance(s, LazyScans):
            return LazyScans(self, self.ops.preclean(s.local, radius), s.kb, s.ke, s.n)
        return self.ops.preclean(s, radius)

    # ---- stages that need every keyframe
    def merge_to_global(self, scans, poses):
        return self.ops.merge_to_global(self.materialize(scans), poses)

    # ---- merge + voxel grid of a RANK-LOCAL scan set: key-range exchange instead of "all-gather the scans, grid 45 M points on every rank"
    @staticmethod
    def balanced_cuts(hist, parts):
        """bin cuts [0, ..., 4096] so that every part gets about the same number of points; identical on every rank (same histogram)"""
        total = int(hist.sum())
        cum = np.concatenate([[0], np.cumsum(hist.astype(np.int64))])
        cuts = [0]
        for r in range(1, parts):
            want = (total * r) // parts
            cuts.append(max(int(np.searchsorted(cum, want, side="left")), cuts[-1]))
        cuts.append(len(hist))
        return [min(c, len(hist)) for c in cuts]

    def _alltoall_clouds(self, parts):
        """parts[r] goes to rank r; returns what this rank received, concatenated in SOURCE-rank order (= keyframe order)"""
        ts = [self.ops.cloud_to_tensor(p) for p in parts]
        dev = ts[0].device
        send_n = torch.tensor([t.shape[0] for t in ts], dtype=torch.int64, device=dev)
        recv_n = torch.empty_like(send_n)
        self.dist.all_to_all_single(recv_n, send_n, group=self.group)
        send_rows, recv_rows = [int(x) for x in send_n.cpu().tolist()], [int(x) for x in recv_n.cpu().tolist()]
        send = torch.cat(ts).contiguous() if sum(send_rows) else torch.zeros((0, 4), dtype=torch.float32, device=dev)
        recv = torch.empty((sum(recv_rows), 4), dtype=torch.float32, device=dev)
        self.dist.all_to_all_single(recv, send, output_split_sizes=recv_rows, input_split_sizes=send_rows, group=self.group)
        return self.ops.cloud_from_tensor(recv)

    def merge_voxel(self, scans, poses, leaf):
        """voxel(merge_to_global(scans, poses), leaf) -- bit for bit -- for a scan set whose keyframes live on their ranks: every rank merges its
        own keyframes, the ranks agree on the bounding box (min / max all-reduce) and on cuts of the octree key space (summed 4096-bin histogram),
        every point moves ONCE to the rank that owns its key range (all-to-all; arrival order = rank order = keyframe order = the input order of the
        single-GPU merge), each rank grids its range under the common frame, and the centroid lists are all-gathered in rank order."""
        if self.world == 1 or not isinstance(scans, LazyScans):
            return self.voxel(self.ops.merge_to_global(self.materialize(scans), poses), leaf)
        kb, ke = self._check(scans, poses)
        loc = self.ops.merge_to_global(scans.local, self._local_poses(poses))
        mn, mx = self.ops.bbox(loc)
        dev = self.ops.cloud_to_tensor(loc).device
        box = torch.tensor([float(v) for v in mn] + [-float(v) for v in mx], dtype=torch.float32, device=dev)     # float32 min / max are exact
        self.dist.all_reduce(box, op=self.dist.ReduceOp.MIN, group=self.group)
        box = box.cpu().numpy()
        gmn, gmx = box[:3].astype(np.float32), (-box[3:]).astype(np.float32)
        if not np.isfinite(gmn).all():                   # no rank has a point
            return self.ops.empty_cloud()
        hist = torch.as_tensor(self.ops.voxel_key_histogram(loc, gmn, gmx, leaf).astype(np.int64), device=dev)
        self.dist.all_reduce(hist, op=self.dist.ReduceOp.SUM, group=self.group)
        cuts = self.balanced_cuts(hist.cpu().numpy(), self.world)
        mine = self._alltoall_clouds(self.ops.voxel_key_split(loc, gmn, gmx, leaf, cuts))
        return self._allgather_cloud(self.ops.voxel_box(mine, gmn, gmx, leaf))

    def merge_voxel_batch(self, merges, clouds, leaf):
        outs = [self.merge_voxel(s, p, leaf) for s, p in merges]
        return outs + (self.voxel_batch(list(clouds), leaf) if clouds else [])


class CommMeter:
    """What the keyframe-sharded pipeline WOULD exchange, recorded on a single-GPU run: wraps the plain stage operations and notes, for every
    stage that ShardedOps follows with a collective, the payload that collective moves (bench.py puts the totals next to the measured
    replicated / sharded kernel time, so that N-GPU scaling can be modelled from measured parts while no multi-GPU node is reachable).
      label_allreduce   one MAX all-reduce of M bytes per vote pass (ShardedOps.vote_partition)
      scans_allgather   all-gather of a rank-local scan set when a stage needs every keyframe (merge_to_global of reprojected / kNN-split scans)
      voxel_allgather   all-gather of the centroid lists of a sharded voxel grid (clouds >= ShardedOps.VOXEL_SHARD_MIN points)"""

    def __init__(self, ops):
        self.ops = ops
        self.events = {"label_allreduce": [0, 0], "scans_allgather": [0, 0], "voxel_allgather": [0, 0], "points_alltoall": [0, 0]}      # [count, payload bytes]
        self.sharded_voxel_points = 0         # input points of the voxel grids whose sort divides by the number of ranks
        # scan sets that would be rank-local (LazyScans) under ShardedOps.  The objects themselves are tagged: an id() of a collected
        # scan set can be handed to an unrelated, replicated one by CPython, which would count collectives that never happen
        self._tag = "_comm_meter_local_%x" % id(self)

    def __getattr__(self, name):
        return getattr(self.ops, name)

    def reset(self):
        for v in self.events.values():
            v[0] = v[1] = 0
        self.sharded_voxel_points = 0

    def _note(self, kind, nbytes):
        self.events[kind][0] += 1
        self.events[kind][1] += int(nbytes)

    def _mark(self, s):
        try:
            setattr(s, self._tag, True)
        except AttributeError:          # an object that cannot carry attributes is wrapped by the caller's ops already; count it as replicated
            pass
        return s

    def _is_local(self, s):
        return getattr(s, self._tag, False) is True

    def vote_partition(self, cmap, scans, poses, alpha, thr, mode):
        self._note("label_allreduce", self.ops.size(cmap))
        return self.ops.vote_partition(cmap, scans, poses, alpha, thr, mode)

    def reproject(self, cmap, poses, alpha): return self._mark(self.ops.reproject(cmap, poses, alpha))

    def knn_partition(self, target, scans, poses, k, thr):
        co, di = self.ops.knn_partition(target, scans, poses, k, thr)
        return self._mark(co), self._mark(di)

    def zip_concat(self, a, b, c): return self._mark(self.ops.zip_concat(a, b, c))
    def voxel_scanset(self, s, leaf): return self._mark(self.ops.voxel_scanset(s, leaf))
    def voxel_grid_scanset(self, s, leaf): return self._mark(self.ops.voxel_grid_scanset(s, leaf)) if self._is_local(s) else self.ops.voxel_grid_scanset(s, leaf)

    def voxel_grid_scanset_begin(self, s, leaf): return (self.ops.voxel_grid_scanset_begin(s, leaf), self._is_local(s))

    def voxel_grid_scanset_end(self, ticket):
        out = self.ops.voxel_grid_scanset_end(ticket[0])
        return self._mark(out) if ticket[1] else out
    def preclean(self, s, radius): return self._mark(self.ops.preclean(s, radius)) if self._is_local(s) else self.ops.preclean(s, radius)

    def merge_to_global(self, scans, poses):
        if self._is_local(scans):
            n_kf, n_pts = scans.info()
            self._note("scans_allgather", 16 * n_pts + 8 * (n_kf + 2))
        return self.ops.merge_to_global(scans, poses)

    def merge_voxel_batch(self, merges, clouds, leaf):
        """what ShardedOps.merge_voxel exchanges for a rank-local scan set: two tiny all-reduces (box, histogram), ONE all-to-all of the points
        (every point crosses the fabric once: a rank receives 1/N of them) and the all-gather of the centroid list"""
        if hasattr(self.ops, "merge_voxel_batch"):
            outs = self.ops.merge_voxel_batch(merges, clouds, leaf)
        else:
            outs = self.ops.voxel_batch([self.ops.merge_to_global(s, p) for s, p in merges] + list(clouds), leaf)
        for (scans, _), o in zip(merges, outs):
            if self._is_local(scans):
                _, n_pts = scans.info()
                self._note("points_alltoall", 16 * n_pts)
                self._note("voxel_allgather", 16 * self.ops.size(o))
                self.sharded_voxel_points += n_pts
            elif self.ops.size(o) and scans.info()[1] >= ShardedOps.VOXEL_SHARD_MIN:
                self._note("voxel_allgather", 16 * self.ops.size(o))
                self.sharded_voxel_points += scans.info()[1]
        for c, o in zip(clouds, outs[len(merges):]):
            if self.ops.size(c) >= ShardedOps.VOXEL_SHARD_MIN:
                self._note("voxel_allgather", 16 * self.ops.size(o))
                self.sharded_voxel_points += self.ops.size(c)
        return outs

    def voxel(self, c, leaf):
        out = self.ops.voxel(c, leaf)
        if self.ops.size(c) >= ShardedOps.VOXEL_SHARD_MIN:
            self._note("voxel_allgather", 16 * self.ops.size(out))
            self.sharded_voxel_points += self.ops.size(c)
        return out

    def voxel_batch(self, clouds, leaf):
        outs = self.ops.voxel_batch(clouds, leaf)
        for c, o in zip(clouds, outs):
            if self.ops.size(c) >= ShardedOps.VOXEL_SHARD_MIN:
                self._note("voxel_allgather", 16 * self.ops.size(o))
                self.sharded_voxel_points += self.ops.size(c)
        return outs


# kernel classes (ltm_profile_read) whose work divides by the number of ranks under keyframe sharding; everything else is replicated
SHARDED_CLASSES = ("vote_map_cull", "vote_map_exact", "vote_scan", "vote_compare", "vote_fill", "reproject_map", "reproject_gather", "knn_query",
                   "knn_query_p2", "voxel_scanset", "voxel_grid_scanset")


def _sharded_ms(class_ms, sharded_voxel_fraction):
    """kernel time of a set of classes that divides by the number of ranks"""
    sharded = sum(v for k, v in class_ms.items() if k in SHARDED_CLASSES)
    # the voxel grids whose input is split over the ranks (key-range exchange of rank-local scans, voxel_centroid_shard of large replicated
    # clouds): their share of the `voxel` class is taken as their share of its input POINTS (sort and tail are linear in them), and the merges
    # in front of them (`merge` class) run on rank-local keyframes
    return sharded + sharded_voxel_fraction * class_ms.get("voxel", 0.0) + (class_ms.get("merge", 0.0) if sharded_voxel_fraction > 0 else 0.0)


def _comm_ms(events, n, link_gbs, latency_us):
    """ring collectives among n ranks at `link_gbs` per direction; the all-to-all spreads over the point-to-point links"""
    if n < 2:
        return 0.0
    ms = 0.0
    for kind, (cnt, nbytes) in events.items():
        if kind == "points_alltoall":
            factor = (n - 1) / float(n * n) / min(n - 1, 7)
        else:
            factor = 2.0 * (n - 1) / n if kind == "label_allreduce" else (n - 1) / n
        ms += 1e3 * factor * nbytes / (link_gbs * 1e9) + cnt * latency_us * 1e-3 * (2 if kind in ("scans_allgather", "points_alltoall") else 1)
    return ms


def scaling_model(class_ms_per_step, step_ms, events_per_step, ranks=(2, 4, 8), link_gbs=150.0, latency_us=25.0, sharded_voxel_fraction=0.0, step1=None):
    """strong-scaling estimate of ONE pair run from single-GPU measurements: T(N) = replicated + sharded / N + collectives(N), with the kernel
    time of the sharded classes measured with HIP events, replicated = the rest of the step (replicated kernels + host gaps), and ring
    collectives at `link_gbs` per direction and `latency_us` each (assumptions, stated in the output: no multi-GPU node was reachable).

    `step1` = {"class_ms": ..., "wall_ms": ..., "events": ..., "sharded_voxel_fraction": ..., "swap_bytes": ...}: the same quantities for the part
    of the step that ShardedOps.session_groups() runs on two rank groups (makeGlobalMap + Step 1: one session per group).  For an even N a rank
    then does the replicated Step-1 work of ONE session (half), shards its session N/2 ways (the same sharded time per rank), exchanges only
    its session's collectives among N/2 ranks and swaps the finished maps with its partner rank."""
    sharded = _sharded_ms(class_ms_per_step, sharded_voxel_fraction)
    replicated = max(step_ms - sharded, 0.0)
    out = {"replicated_ms": round(replicated, 3), "sharded_ms": round(sharded, 3), "comm_events_per_step": {k: {"count": v[0], "payload_bytes": v[1]} for k, v in events_per_step.items()},
           "comm_bytes_per_step": int(sum(v[1] for v in events_per_step.values())),
           "assumptions": {"ring_bandwidth_GB_s_per_direction": link_gbs, "latency_us_per_collective": latency_us,
                           "all_reduce_bytes_on_the_wire_per_rank": "2 (N-1)/N x payload", "all_gather": "(N-1)/N x payload",
                           "all_to_all": "a rank sends and receives (N-1)/N^2 x payload, spread over min(N-1, 7) point-to-point xGMI links",
                           "sharded_voxel_fraction_of_the_voxel_class": round(sharded_voxel_fraction, 4)},
           "status": "MODEL from single-GPU measurements -- unmeasured on multi-GPU hardware", "ranks": {}}
    g = None
    if step1:
        s1 = _sharded_ms(step1["class_ms"], step1.get("sharded_voxel_fraction", 0.0))
        r1 = max(step1["wall_ms"] - s1, 0.0)
        ev2 = {k: (v[0] - step1["events"].get(k, (0, 0))[0], v[1] - step1["events"].get(k, (0, 0))[1]) for k, v in events_per_step.items()}
        ev1_one_session = {k: (v[0] / 2.0, v[1] / 2.0) for k, v in step1["events"].items()}
        g = {"replicated_ms": r1, "sharded_ms": s1, "rest_replicated_ms": max(replicated - r1, 0.0), "rest_sharded_ms": max(sharded - s1, 0.0)}
        out["session_groups"] = {"what": "makeGlobalMap + Step 1 on two rank groups, one session each (ShardedOps.session_groups; even N)",
                                 "step1_wall_ms": round(step1["wall_ms"], 3), "step1_replicated_ms": round(r1, 3), "step1_sharded_ms": round(s1, 3),
                                 "swap_bytes_each_way": int(step1.get("swap_bytes", 0)),
                                 "step1_comm_events_per_step": {k: {"count": v[0], "payload_bytes": v[1]} for k, v in step1["events"].items()}}
    for n in ranks:
        comm_ms = _comm_ms(events_per_step, n, link_gbs, latency_us)
        received = sum((nbytes / n if kind == "points_alltoall" else nbytes) for kind, (cnt, nbytes) in events_per_step.items())
        t = replicated + sharded / n + comm_ms
        row = {"comm_ms": round(comm_ms, 3), "step_ms": round(t, 3), "speedup": round(step_ms / t, 3), "bytes_received_per_rank_per_step": int(received)}
        if g and n % 2 == 0:
            swap_ms = 1e3 * step1.get("swap_bytes", 0) / (link_gbs * 1e9) + 2 * latency_us * 1e-3
            comm1 = _comm_ms(ev1_one_session, n // 2, link_gbs, latency_us) + swap_ms
            comm2 = _comm_ms(ev2, n, link_gbs, latency_us)
            tg = g["replicated_ms"] / 2.0 + g["sharded_ms"] / n + comm1 + g["rest_replicated_ms"] + g["rest_sharded_ms"] / n + comm2
            rec = (sum((b / (n // 2) if k == "points_alltoall" else b) for k, (c, b) in ev1_one_session.items()) + step1.get("swap_bytes", 0) +
                   sum((b / n if k == "points_alltoall" else b) for k, (c, b) in ev2.items()))
            row = {"comm_ms": round(comm1 + comm2, 3), "step_ms": round(tg, 3), "speedup": round(step_ms / tg, 3), "bytes_received_per_rank_per_step": int(rec),
                   "without_session_groups": row}
        out["ranks"][str(n)] = row
    return out
