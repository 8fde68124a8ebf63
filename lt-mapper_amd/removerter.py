"""Python mirror of the reference's `Removerter` / `Session` pipeline over the C ABI (include/ltm.h).

Method and member names follow ltremovert/include/removert/{Removerter.h,Session.h} so that the parity
tests read like the reference; every method cites the reference lines it stands for.  All heavy work is
a call into libltm_hip.so through an `ops` object (HipOps below); nothing here computes on points.

The orchestration is written against a small ops interface so the keyframe-sharded multi-GPU variant
(dist.ShardedOps) wraps the same pipeline: per-keyframe stages run on this rank's block of keyframes and
label masks / per-keyframe clouds are exchanged with torch.distributed (RCCL on GPUs).
"""
import time
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class Params:
    """the removert/ ROS parameters the hot path reads (RosParamServer.cpp:7-59)"""
    sequence_vfov: float = 50.0
    sequence_hfov: float = 360.0
    remove_resolution_list: List[float] = field(default_factory=lambda: [2.5])
    num_nn_points_within: int = 2            # yaml value; code default 3
    dist_nn_points_within: float = 0.01      # yaml value; code default 0.1
    downsample_voxel_size: float = 0.05
    repeat_removert_iter: int = 1
    ExtrinsicLiDARtoPoseBase: Optional[np.ndarray] = None
    # new optional keys (prefix gpu_): off = exactly the shipped behaviour
    gpu_use_self_removert: bool = False      # enable the selfRemovert() call that is commented out at Removerter.cpp:1582,1586
    gpu_skip_hd_knn: bool = False            # skip the visualisation-only HD kNN (Removerter.cpp:1590-1601)
    gather_scan_outputs: bool = False        # multi-GPU: also assemble the five per-keyframe outputs on every rank (default: rank-local)


class HipOps:
    """stage operations on one GPU: thin forwarding to capi.Context"""

    def __init__(self, ctx):
        self.ctx = ctx

    # clouds
    def clone(self, c): return c.clone()
    def concat(self, cs): return self.ctx.concat(cs)
    def size(self, c): return len(c)
    def empty_cloud(self): return self.ctx.upload(np.zeros((0, 4), np.float32))
    # stages
    def merge_to_global(self, scans, poses): return self.ctx.merge_to_global(scans, poses)
    def voxel(self, c, leaf): return self.ctx.voxel_centroid(c, leaf)
    def voxel_batch(self, clouds, leaf): return self.ctx.voxel_centroid_batch(clouds, [leaf] * len(clouds))   # independent grids, two host round trips in all
    def voxel_shard(self, c, leaf, shard, n_shards): return self.ctx.voxel_centroid_shard(c, leaf, shard, n_shards)
    # key-range exchange pieces (dist.ShardedOps.merge_voxel)
    def bbox(self, c): return self.ctx.bbox(c)
    def voxel_key_histogram(self, c, mn, mx, leaf): return self.ctx.voxel_key_histogram(c, mn, mx, leaf)
    def voxel_key_split(self, c, mn, mx, leaf, cuts): return self.ctx.voxel_key_split(c, mn, mx, leaf, cuts)
    def voxel_box(self, c, mn, mx, leaf): return self.ctx.voxel_centroid_box(c, mn, mx, leaf)

    def merge_voxel_batch(self, merges, clouds, leaf):
        """voxel grids of [merge_to_global(scans, poses) for (scans, poses) in merges] + clouds, as ONE batch (single GPU: nothing to exchange)"""
        return self.voxel_batch([self.merge_to_global(s, p) for s, p in merges] + list(clouds), leaf)
    def voxel_scanset(self, s, leaf): return self.ctx.voxel_centroid_scanset(s, leaf)
    def voxel_grid_scanset(self, s, leaf): return self.ctx.voxel_grid_scanset(s, leaf)      # the loader's pcl::VoxelGrid, Session.cpp:284-289
    supports_deferred_grid = True

    def voxel_grid_scanset_begin(self, s, leaf): return self.ctx.voxel_grid_scanset_begin(s, leaf)      # ... in two halves (cascade hand-over)
    def voxel_grid_scanset_end(self, ticket): return self.ctx.voxel_grid_scanset_end(ticket)
    def preclean(self, s, radius): return self.ctx.preclean(s, radius)                      # Session.cpp:506-533
    def vote_partition(self, cmap, scans, poses, alpha, thr, mode): return self.ctx.visibility_partition(cmap, scans, poses, alpha, thr, mode)
    def prepare_scan_images(self, scans, alphas): self.ctx.prepare_scan_images(scans, alphas)
    def reproject(self, cmap, poses, alpha): return self.ctx.reproject(cmap, poses, alpha)
    def knn_partition(self, target, scans, poses, k, thr): return self.ctx.knn_partition(target, scans, poses, k, thr)
    def knn_split(self, target, query, k, thr): return self.ctx.knn_split_cloud(target, query, k, thr)
    def zip_concat(self, a, b, c): return self.ctx.zip_concat(a, b, c)
    def sync(self): self.ctx.synchronize()
    def materialize(self, scans): return scans          # single GPU: scan sets are always whole
    # lanes (include/ltm.h "lanes"): a second context on the same device, driven by a second host thread
    def lane(self): return HipOps(self.ctx.lane())
    def poses_like(self, poses): return self.ctx.poses(poses.host_poses, poses.host_inv)      # the same keyframe poses as a handle of THIS context

    # ---- pieces used by dist.ShardedOps (keyframe ranges + tensors for the collectives)
    def n_keyframes(self, poses): return poses.n
    def poses_slice(self, poses, kb, ke): return self.ctx.poses(poses.host_poses[kb:ke], poses.host_inv[kb:ke])

    def new_labels(self, n):
        import torch
        t = torch.zeros(max(n, 1), dtype=torch.uint8, device=f"cuda:{self.ctx.device}")[:n]
        # the zero-fill runs on torch's stream, the vote writes the buffer on the context's own (non-blocking) stream: order them
        torch.cuda.current_stream().synchronize()
        return t

    def vote(self, cmap, scans, poses, kb, ke, alpha, thr, mode, labels):
        if labels.numel():
            self.ctx.visibility_vote(cmap, scans, poses, kb, ke, alpha, thr, mode, labels.data_ptr())

    def partition(self, cmap, labels):
        import torch
        torch.cuda.current_stream().synchronize()      # the all-reduce ran on torch's stream
        return self.ctx.partition_by_labels(cmap, labels.data_ptr() if labels.numel() else None)

    def reproject_range(self, cmap, poses, alpha, kb, ke): return self.ctx.reproject(cmap, poses, alpha, kb, ke)
    def knn_partition_range(self, target, scans, poses, k, thr, kb, ke): return self.ctx.knn_partition(target, scans, poses, k, thr, kb, ke)
    def concat_scansets(self, sets): return self.ctx.concat_scansets(sets)

    def scanset_to_tensors(self, ss):
        import torch
        n_kf, n = ss.info()
        off = ss.offsets()
        self.ctx.synchronize()
        if n == 0:
            return torch.zeros((0, 4), dtype=torch.float32, device=f"cuda:{self.ctx.device}"), off
        view = _DeviceArray(ss.device_ptr(), (n, 4), "<f4")
        return torch.as_tensor(view, device=f"cuda:{self.ctx.device}").clone(), off

    def cloud_to_tensor(self, c):
        """zero-copy (n, 4) float32 view of a device cloud; valid while `c` lives"""
        import torch
        n = len(c)
        self.ctx.synchronize()
        if n == 0:
            return torch.zeros((0, 4), dtype=torch.float32, device=f"cuda:{self.ctx.device}")
        return torch.as_tensor(_DeviceArray(c.device_ptr(), (n, 4), "<f4"), device=f"cuda:{self.ctx.device}")

    def cloud_from_tensor(self, t):
        import torch
        torch.cuda.current_stream().synchronize()
        if t.shape[0] == 0:
            return self.empty_cloud()
        return self.ctx.cloud_from_device(t.data_ptr(), t.shape[0])

    def scanset_from_tensors(self, pts, offsets):
        import torch
        torch.cuda.current_stream().synchronize()
        return self.ctx.scans_from_device(pts.data_ptr() if pts.numel() else None, np.asarray(offsets, dtype=np.uint64))


class _DeviceArray:
    """zero-copy view of library-owned device memory for torch (CUDA array interface v2)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class Session:
    """ltremovert::Session state (Session.h:9-136): everything is a device handle"""

    def __init__(self, sess_type, scans, poses):
        self.sess_type_ = sess_type
        self.keyframe_scans_ = scans            # scan set (after load + pre-clean)
        self.keyframe_poses = poses             # poses handle (keyframe_poses_ + keyframe_inverse_poses_)
        self.map_global_orig_ = None
        self.map_global_curr_ = None
        self.map_global_curr_static_ = None
        self.map_global_curr_dynamic_ = None
        self.keyframe_scans_static_projected_ = None
        self.keyframe_scans_dynamic_ = None
        self.scans_knn_coexist_ = None
        self.scans_knn_diff_ = None
        self.map_global_nd_ = self.map_global_nd_strong_ = self.map_global_nd_weak_ = None
        self.map_global_pd_ = self.map_global_pd_orig_ = self.map_global_pd_strong_ = self.map_global_pd_weak_ = None
        self.map_global_updated_ = self.map_global_updated_strong_ = None
        self.keyframe_scans_updated_ = self.keyframe_scans_updated_strong_ = None
        self.keyframe_scans_pd_ = self.keyframe_scans_strong_pd_ = None
        self.keyframe_scans_strong_nd_ = self.keyframe_scans_weak_nd_ = None


class Removerter:
    kReprojectionAlpha = 3.0     # Session.h:13

    def __init__(self, ops, params: Params, central: Session, query: Session, query_side=None, lane_ops=None):
        """`lane_ops` = ops of a LANE of `ops`'s context (HipOps.lane()): the independent chains of run() -- the two sessions' makeGlobalMap + Step 1, the two
        directions of the LD kNN diff, the ND against the PD filter, the reprojections of Step 3 -- run side by side, the query / PD / "strong" halves on the
        lane from a second host thread (run_two_lanes).  Same clouds as the one-lane order, bit for bit: every kernel sees the same inputs.

        `query_side` = (ops2, query session as ops2 sees it): an experiment, off by default -- the merge + grid and the Step-1 chain of the
        query session run on a second device context (own stream, own pool) from a second host thread while this one does the central
        session's; the two chains share nothing (Removerter.cpp:1584-1591 runs them one after the other), their results are the same clouds."""
        self.ops, self.P = ops, params
        self.central_sess_, self.query_sess_ = central, query
        self.query_side = query_side
        self.lane_ops = lane_ops
        # cascade.run_cascade: a callable that delivers the central session's scans (the previous run's scans_updated, re-gridded: its host half is still
        # running when this run starts).  While it is set, run() takes the query session's makeGlobalMap + Step-1 chain first -- it does not depend on the
        # central scans (Removerter.cpp:1584-1591 runs the two sessions one after the other, either order gives the same clouds) -- and asks for the scans then
        self.central_scans_future = None
        self.on_stage = None
        self.outputs = {}        # name -> cloud handle, the *.pcd maps of the output protocol
        self.timings = {}

    # ------------------------------------------------------------------ helpers
    def _tick(self, name, t0):
        self.ops.sync()
        self.timings[name] = self.timings.get(name, 0.0) + (time.perf_counter() - t0)
        if self.on_stage is not None:      # bench.py: per-stage snapshots of the kernel-class profile and of the comm meter
            self.on_stage(name)

    def octreeDownsampling(self, cloud, leaf):          # utility.cpp:204-219
        return self.ops.voxel(cloud, leaf)

    def octreeDownsamplingBatch(self, clouds, leaf):    # the same for several independent clouds (consecutive calls in the reference)
        return self.ops.voxel_batch(list(clouds), leaf)

    def mergeVoxelBatch(self, merges, clouds, leaf):
        """octreeDownsampling(mergeScansWithinGlobalCoordUtil(scans, poses)) for every (scans, poses) of `merges` (utility.cpp:170-192 + :204-219) and
        octreeDownsampling of every cloud of `clouds`, as one batch.  One GPU: merges, then all grids together.  Keyframe-sharded: the merges of
        rank-local scan sets become key-range exchanges (dist.ShardedOps.merge_voxel) instead of all-gathers of the scans."""
        if hasattr(self.ops, "merge_voxel_batch"):
            return self.ops.merge_voxel_batch(list(merges), list(clouds), leaf)
        return self.ops.voxel_batch([self.ops.merge_to_global(s, p) for s, p in merges] + list(clouds), leaf)

    def _append(self, a, b, ops=None):                  # `*a += *b`
        ops = ops or self.ops
        if a is None:
            return ops.clone(b)
        return ops.concat([a, b])

    # ------------------------------------------------------------------ Step 0
    def makeGlobalMap(self):                            # Removerter.cpp:213-252 (+ Session.cpp:186-202)
        t0 = time.perf_counter()
        sess = (self.central_sess_, self.query_sess_)
        for s in sess:
            s.map_global_orig_ = self.ops.merge_to_global(s.keyframe_scans_, s.keyframe_poses)
        for s, m in zip(sess, self.octreeDownsamplingBatch([s.map_global_orig_ for s in sess], self.P.downsample_voxel_size)):
            s.map_global_curr_ = m
            self.outputs["OriginalNoisy" + s.sess_type_ + "MapGlobal"] = s.map_global_curr_
            s.map_global_orig_ = None   # only ever read by makeGlobalMap
        self._tick("make_global_map", t0)

    # ------------------------------------------------------------------ Step 1
    def removeOnce(self, target, source, res_alpha, ops=None):    # Removerter.cpp:882-905
        ops = ops or self.ops
        static_tt, dynamic_tt = ops.vote_partition(target.map_global_curr_, source.keyframe_scans_, source.keyframe_poses, res_alpha, 0.1, 0)
        target.map_global_curr_static_, target.map_global_curr_dynamic_ = ops.voxel_batch(
            [static_tt, self._append(target.map_global_curr_dynamic_, dynamic_tt, ops)], 0.05)      # :896, :903 -- independent of each other
        target.map_global_curr_ = target.map_global_curr_static_

    def revertOnce(self, target, source, res_alpha, ops=None):    # Removerter.cpp:908-931
        ops = ops or self.ops
        static_tt, dynamic_tt = ops.vote_partition(target.map_global_curr_, source.keyframe_scans_, source.keyframe_poses, res_alpha, 0.1, 0)
        target.map_global_curr_dynamic_, target.map_global_curr_static_ = ops.voxel_batch(
            [dynamic_tt, self._append(target.map_global_curr_static_, static_tt, ops)], 0.05)       # :921, :928
        target.map_global_curr_ = target.map_global_curr_dynamic_

    def selfRemovert(self, sess, repeat=1, ops=None):   # Removerter.cpp:1378-1393
        prepare = getattr(ops or self.ops, "prepare_scan_images", None)
        if prepare is not None and repeat > 0:      # every scan image the passes below will ask for, in one pass over the scans
            rs = [float(np.float32(r)) for r in self.P.remove_resolution_list]
            prepare(sess.keyframe_scans_, rs + [float(np.float32(0.95 * r)) for r in rs])
        for res in self.P.remove_resolution_list:
            res = float(np.float32(res))
            for _ in range(repeat):                                            # `i < _repeat`, Removerter.cpp:1381: repeat 0 runs nothing
                self.removeOnce(sess, sess, res, ops)
                sess.map_global_curr_ = sess.map_global_curr_dynamic_          # resetCurrrentMapAsDynamic :714-737
                self.revertOnce(sess, sess, float(np.float32(0.95 * res)), ops)   # :1385 double product narrowed to float
                sess.map_global_curr_ = sess.map_global_curr_static_           # resetCurrrentMapAsStatic
                self.removeOnce(sess, sess, res, ops)

    def _removeHighDynamicOf(self, sess, ops=None):     # one session's share of Removerter.cpp:1584-1591
        if self.P.gpu_use_self_removert and len(self.P.remove_resolution_list) > 0:
            self.selfRemovert(sess, self.P.repeat_removert_iter, ops)
        else:
            self.removeOnce(sess, sess, 2.5, ops)

    def _sessionsSideBySide(self):
        """makeGlobalMap + the Step-1 chain of both sessions, the query session's on `query_side`'s context from a second thread (see __init__).
        Afterwards the query session's three maps are copied into the main context (a device copy of ~0.3 GB) where everything else runs."""
        import threading
        ops2, Q2 = self.query_side
        C, Q = self.central_sess_, self.query_sess_
        err = []

        def chain(s, ops):
            try:
                s.map_global_curr_ = ops.voxel(ops.merge_to_global(s.keyframe_scans_, s.keyframe_poses), self.P.downsample_voxel_size)
                s.map_global_orig_noisy_ = s.map_global_curr_
                self._removeHighDynamicOf(s, ops)
                ops.sync()
            except BaseException as e:      # noqa: BLE001 -- re-raised on the calling thread
                err.append(e)
        self.ops.sync()                      # nothing of the previous step is in flight when the second stream starts to reuse its blocks
        th = threading.Thread(target=chain, args=(Q2, ops2))
        th.start()
        chain(C, self.ops)
        th.join()
        if err:
            raise err[0]
        for name in ("map_global_orig_noisy_", "map_global_curr_static_", "map_global_curr_dynamic_"):
            src = getattr(Q2, name)
            setattr(Q, name, self.ops.cloud_from_tensor(ops2.cloud_to_tensor(src)) if src is not None else None)
        Q.map_global_curr_ = Q.map_global_curr_static_
        for s in (C, Q):
            self.outputs["OriginalNoisy" + s.sess_type_ + "MapGlobal"] = s.map_global_orig_noisy_

    def _rankGroups(self):
        """dist.ShardedOps.session_groups(): (group ops, which session this rank's group works on) when the ranks split into a central and a query group"""
        f = getattr(self.ops, "session_groups", None)
        return f() if f is not None and self.query_side is None else None

    def _sessionsOnRankGroups(self, gops, g):
        """makeGlobalMap + all of Step 1 (remove / revert passes and the HD kNN map) of ONE session on this rank's group, the other session's on the
        other group at the same time; then rank pairs swap the finished maps, after which every rank holds what the plain order leaves behind."""
        t0 = time.perf_counter()
        C, Q = self.central_sess_, self.query_sess_
        mine, other = (C, Q) if g == 0 else (Q, C)
        mine.map_global_curr_ = gops.voxel(gops.merge_to_global(mine.keyframe_scans_, mine.keyframe_poses), self.P.downsample_voxel_size)
        mine.map_global_orig_noisy_ = mine.map_global_curr_
        self._removeHighDynamicOf(mine, gops)
        give = [mine.map_global_orig_noisy_, mine.map_global_curr_static_, mine.map_global_curr_dynamic_]
        if not self.P.gpu_skip_hd_knn:
            k, thr = self.P.num_nn_points_within, self.P.dist_nn_points_within
            _, mine.keyframe_scans_dynamic_ = gops.knn_partition(mine.map_global_curr_static_, mine.keyframe_scans_, mine.keyframe_poses, k, thr)  # Session.cpp:487-504
            give.append(gops.merge_voxel_batch([(mine.keyframe_scans_dynamic_, mine.keyframe_poses)], [], 0.05)[0])
        got = gops.swap_clouds_with_peer(give)
        other.map_global_orig_noisy_, other.map_global_curr_static_, other.map_global_curr_dynamic_ = got[:3]
        other.map_global_curr_ = other.map_global_curr_static_
        for s in (C, Q):
            self.outputs["OriginalNoisy" + s.sess_type_ + "MapGlobal"] = s.map_global_orig_noisy_
        if not self.P.gpu_skip_hd_knn:
            hd = {id(mine): give[3], id(other): got[3]}
            self.outputs["central_sess_high_dyn"], self.outputs["query_sess_high_dyn"] = hd[id(C)], hd[id(Q)]
        self._tick("remove_high_dynamic", t0)

    def _queryThenCentral(self):
        """makeGlobalMap + the Step-1 chains with the query session first; the central scans are asked for when the query session's chain has been issued"""
        C, Q = self.central_sess_, self.query_sess_
        for s in (Q, C):
            if s is C:
                s.keyframe_scans_ = self.central_scans_future()
                self.central_scans_future = None
            s.map_global_curr_ = self.octreeDownsampling(self.ops.merge_to_global(s.keyframe_scans_, s.keyframe_poses), self.P.downsample_voxel_size)
            self.outputs["OriginalNoisy" + s.sess_type_ + "MapGlobal"] = s.map_global_curr_
            self._removeHighDynamicOf(s)

    def removeHighDynamicPoints(self):                  # Removerter.cpp:1580-1604
        groups = self._rankGroups()
        if groups is not None:
            return self._sessionsOnRankGroups(*groups)
        t0 = time.perf_counter()
        C, Q = self.central_sess_, self.query_sess_
        if self.central_scans_future is not None:
            self._queryThenCentral()
        elif self.query_side is not None:
            self._sessionsSideBySide()
        else:
            self._removeHighDynamicOf(C)
            self._removeHighDynamicOf(Q)
        self._tick("remove_high_dynamic", t0)
        if not self.P.gpu_skip_hd_knn:
            t0 = time.perf_counter()
            k, thr = self.P.num_nn_points_within, self.P.dist_nn_points_within
            for s in (C, Q):
                _, s.keyframe_scans_dynamic_ = self.ops.knn_partition(s.map_global_curr_static_, s.keyframe_scans_, s.keyframe_poses, k, thr)  # Session.cpp:487-504
            self.outputs["central_sess_high_dyn"], self.outputs["query_sess_high_dyn"] = self.mergeVoxelBatch(
                [(s.keyframe_scans_dynamic_, s.keyframe_poses) for s in (C, Q)], [], 0.05)
            self._tick("hd_knn", t0)

    def parseStaticScansViaProjection(self):            # Removerter.cpp:1527-1538, Session.cpp:305-309
        t0 = time.perf_counter()
        for s in (self.central_sess_, self.query_sess_):
            s.keyframe_scans_static_projected_ = self.ops.reproject(s.map_global_curr_, s.keyframe_poses, self.kReprojectionAlpha)
        self._tick("reproject_static", t0)

    # ------------------------------------------------------------------ Step 2
    def _removeOnceLD(self, target_maps, source, res_alpha, mode, ops=None):   # iremoveOnceForND :831-854 / removeOnceForPD :856-880
        ops = ops or self.ops
        cur, strong, weak = target_maps
        static_tt, dynamic_tt = ops.vote_partition(cur, source.keyframe_scans_static_projected_, source.keyframe_poses, res_alpha, 0.1, mode)
        strong, weak = ops.voxel_batch([static_tt, self._append(weak, dynamic_tt, ops)], 0.05)
        cur = strong
        return cur, strong, weak

    def detectLowDynamicPoints(self):                   # Removerter.cpp:1413-1481
        C, Q = self.central_sess_, self.query_sess_
        k, thr = self.P.num_nn_points_within, self.P.dist_nn_points_within
        t0 = time.perf_counter()
        # Session.cpp:393-427 (the 0.4 m octree for the disabled ICP at :401 has no observable effect and is not run)
        C.scans_knn_coexist_, C.scans_knn_diff_ = self.ops.knn_partition(Q.map_global_curr_static_, C.keyframe_scans_static_projected_, C.keyframe_poses, k, thr)
        Q.scans_knn_coexist_, Q.scans_knn_diff_ = self.ops.knn_partition(C.map_global_curr_static_, Q.keyframe_scans_static_projected_, Q.keyframe_poses, k, thr)
        self._tick("ld_knn", t0)
        t0 = time.perf_counter()
        # strong ND: constructGlobalNDMap (Session.cpp:430-435), filterStrongND (:1403-1411), weak->strong propagation (Session.cpp:452-484)
        C.map_global_nd_ = self.mergeVoxelBatch([(C.scans_knn_diff_, C.keyframe_poses)], [], 0.05)[0]
        maps = (C.map_global_nd_, None, None)
        for _ in range(3):
            maps = self._removeOnceLD(maps, Q, 2.5, 1)
        C.map_global_nd_, C.map_global_nd_strong_, C.map_global_nd_weak_ = maps
        if self.ops.size(C.map_global_nd_strong_) != 0:
            add, new_weak = self.ops.knn_split(C.map_global_nd_strong_, C.map_global_nd_weak_, 2, 1.0)
            C.map_global_nd_strong_ = self.ops.concat([C.map_global_nd_strong_, add])
            C.map_global_nd_weak_ = new_weak
        # strong PD: constructGlobalPDMap (Session.cpp:437-445), filterStrongPD (:1395-1401)
        Q.map_global_pd_ = self.mergeVoxelBatch([(Q.scans_knn_diff_, Q.keyframe_poses)], [], 0.05)[0]
        Q.map_global_pd_orig_ = Q.map_global_pd_
        maps = (Q.map_global_pd_, None, None)
        for _ in range(3):
            maps = self._removeOnceLD(maps, C, 2.5, 0)
        Q.map_global_pd_, Q.map_global_pd_strong_, Q.map_global_pd_weak_ = maps
        C.map_global_pd_, C.map_global_pd_orig_, C.map_global_pd_strong_ = Q.map_global_pd_, Q.map_global_pd_orig_, Q.map_global_pd_strong_   # :1435-1437
        self._tick("ld_filter", t0)
        t0 = time.perf_counter()
        # :1443-1480 merged maps "for visual debug" -- several of these re-voxelise state that Step 3 reads
        o = self.outputs
        merges = [(Q.scans_knn_coexist_, Q.keyframe_poses), (C.scans_knn_coexist_, C.keyframe_poses),
                  (Q.scans_knn_diff_, Q.keyframe_poses), (C.scans_knn_diff_, C.keyframe_poses)]
        ins = [C.map_global_nd_weak_, Q.map_global_pd_strong_, Q.map_global_pd_weak_]
        has_strong_nd = self.ops.size(C.map_global_nd_strong_) != 0
        if has_strong_nd:
            ins.append(C.map_global_nd_strong_)
        res = self.mergeVoxelBatch(merges, ins, 0.05)      # eight independent grids (:1445-1476), one batch
        self._union_q = o["union_map_queryside"] = res[0]
        self._union_c = o["union_map_centralside"] = res[1]
        o["pd_map"], o["nd_map"] = res[2], res[3]
        C.map_global_nd_weak_ = o["weak_nd_map"] = res[4]
        Q.map_global_pd_strong_ = o["strong_pd_map"] = res[5]
        Q.map_global_pd_weak_ = o["weak_pd_map"] = res[6]
        if has_strong_nd:
            C.map_global_nd_strong_ = o["strong_nd_map"] = res[7]
        self._tick("ld_maps", t0)

    # ------------------------------------------------------------------ Step 3
    def updateCurrentMap(self):                         # Removerter.cpp:1483-1524
        t0 = time.perf_counter()
        C = self.central_sess_
        # the two union maps are recomputed at :1489-1493 from unchanged inputs: identical to the ones of :1445-1451
        updated = self.ops.concat([self._union_q, self._union_c, C.map_global_nd_weak_])
        C.map_global_updated_strong_, C.map_global_updated_ = self.octreeDownsamplingBatch(
            [self.ops.concat([updated, C.map_global_pd_strong_]), self.ops.concat([updated, C.map_global_pd_orig_])], 0.05)
        self.outputs["updated_map"] = C.map_global_updated_
        self.outputs["updated_map_strong"] = C.map_global_updated_strong_
        self._tick("update_map", t0)

    def parseUpdatedStaticScansViaProjection(self):     # Removerter.cpp:1551-1562
        t0 = time.perf_counter()
        C = self.central_sess_
        C.keyframe_scans_updated_ = self.ops.reproject(C.map_global_updated_, C.keyframe_poses, self.kReprojectionAlpha)
        C.keyframe_scans_updated_strong_ = self.ops.reproject(C.map_global_updated_strong_, C.keyframe_poses, self.kReprojectionAlpha)
        self._tick("reproject_updated", t0)

    def parseLDScansViaProjection(self):                # Removerter.cpp:1564-1577
        t0 = time.perf_counter()
        C = self.central_sess_
        C.keyframe_scans_pd_ = self.ops.reproject(C.map_global_pd_orig_, C.keyframe_poses, self.kReprojectionAlpha)
        C.keyframe_scans_strong_pd_ = self.ops.reproject(C.map_global_pd_strong_, C.keyframe_poses, self.kReprojectionAlpha)
        C.keyframe_scans_weak_nd_ = self.ops.reproject(C.map_global_nd_weak_, C.keyframe_poses, self.kReprojectionAlpha)
        nd_strong = C.map_global_nd_strong_ if C.map_global_nd_strong_ is not None else self.ops.empty_cloud()
        C.keyframe_scans_strong_nd_ = self.ops.reproject(nd_strong, C.keyframe_poses, self.kReprojectionAlpha)
        self._tick("reproject_ld", t0)

    def updateScansScanwise(self):                      # Removerter.cpp:1540-1549, Session.cpp:362-380
        t0 = time.perf_counter()
        C = self.central_sess_
        merged = self.ops.zip_concat(C.keyframe_scans_updated_, C.keyframe_scans_weak_nd_, C.keyframe_scans_pd_)
        C.keyframe_scans_updated_ = self.ops.voxel_scanset(merged, 0.05)
        self._tick("update_scans", t0)

    def scan_outputs(self):
        """the five per-keyframe output directories (Removerter.cpp:1607-1650)"""
        C = self.central_sess_
        return {"scans_updated": C.keyframe_scans_updated_, "scans_updated_strong": C.keyframe_scans_updated_strong_,
                "scans_pd": C.keyframe_scans_pd_, "scans_pd_strong": C.keyframe_scans_strong_pd_,
                "scans_nd_strong": C.keyframe_scans_strong_nd_}

    # ------------------------------------------------------------------ two lanes (include/ltm.h "lanes")
    def _fork(self, main_fn, lane_fn):
        """main_fn on this thread and context, lane_fn on a second thread and the lane context; returns when both are done (host side) and the main context's
        next work is ordered after everything the lane submitted (device side)"""
        import threading
        M, L = self.ops.ctx, self.lane_ops.ctx
        err = []

        def run():
            try:
                lane_fn()
            except BaseException as e:      # noqa: BLE001 -- re-raised on the calling thread
                err.append(e)
        M.fence(L)                           # the lane starts after what the main context has been given so far (its inputs, recycled blocks)
        th = threading.Thread(target=run)
        th.start()
        try:
            main_fn()
        finally:
            th.join()
        if err:
            raise err[0]
        L.fence(M)                           # later frees / overwrites of lent clouds on the main context run after the lane's reads

    def run_two_lanes(self):
        """run() with the independent chains side by side on the main context and its lane (Removerter.cpp:1653-1678; which stages are independent: include/ltm.h).
        Stage A: makeGlobalMap + Step-1 chain + HD kNN map + static reprojection of the central session (main) and of the query session (lane); the library
                 chains the two lanes' large projection launches, which keeps the chains in anti-phase -- one lane's partition + grids under the other's vote;
        Stage B: C -> Q kNN diff, ND filter and the central-side grids of :1445-1476 (main); Q -> C kNN diff, PD filter and the query-side grids (lane);
        Stage C: updateCurrentMap, then the reprojections of the updated / weak-ND / PD maps and updateScansScanwise (main) beside those of the
                 "strong" maps (lane)."""
        ops, lops, P = self.ops, self.lane_ops, self.P
        M, L = ops.ctx, lops.ctx
        C, Q = self.central_sess_, self.query_sess_
        k, thr = P.num_nn_points_within, P.dist_nn_points_within
        o = self.outputs
        # ---- stage A
        t0 = time.perf_counter()
        Ql = Session("Query", M.lend(Q.keyframe_scans_, L), lops.poses_like(Q.keyframe_poses))      # the lane's view of the query session
        Cl_poses = lops.poses_like(C.keyframe_poses)

        def chain(s, op):
            if s is C and self.central_scans_future is not None:      # cascade: the hand-over's host half ends here, the lane is already at work
                s.keyframe_scans_ = self.central_scans_future()
                self.central_scans_future = None
            s.map_global_curr_ = op.voxel(op.merge_to_global(s.keyframe_scans_, s.keyframe_poses), P.downsample_voxel_size)
            s.map_global_orig_noisy_ = s.map_global_curr_
            self._removeHighDynamicOf(s, op)
            if not P.gpu_skip_hd_knn:
                _, s.keyframe_scans_dynamic_ = op.knn_partition(s.map_global_curr_static_, s.keyframe_scans_, s.keyframe_poses, k, thr)  # Session.cpp:487-504
                s.high_dyn_ = op.merge_voxel_batch([(s.keyframe_scans_dynamic_, s.keyframe_poses)], [], 0.05)[0]
            s.keyframe_scans_static_projected_ = op.reproject(s.map_global_curr_, s.keyframe_poses, self.kReprojectionAlpha)      # Session.cpp:305-309
        self._fork(lambda: chain(C, ops), lambda: chain(Ql, lops))
        for name in ("map_global_orig_noisy_", "map_global_curr_static_", "map_global_curr_dynamic_"):
            setattr(Q, name, L.give(getattr(Ql, name), M))
        Q.map_global_curr_ = Q.map_global_curr_static_
        for s in (C, Q):
            o["OriginalNoisy" + s.sess_type_ + "MapGlobal"] = s.map_global_orig_noisy_
        if not P.gpu_skip_hd_knn:
            Q.keyframe_scans_dynamic_ = L.give(Ql.keyframe_scans_dynamic_, M)
            o["central_sess_high_dyn"], o["query_sess_high_dyn"] = C.high_dyn_, L.give(Ql.high_dyn_, M)
        self._tick("lanes_step1", t0)
        # ---- stage B
        t0 = time.perf_counter()
        Qproj_view = L.lend(Ql.keyframe_scans_static_projected_, M)
        Cstat_view, Cproj_view = M.lend(C.map_global_curr_static_, L), M.lend(C.keyframe_scans_static_projected_, L)
        q_src = Session("Query", None, Q.keyframe_poses); q_src.keyframe_scans_static_projected_ = Qproj_view
        c_src = Session("Central", None, Cl_poses); c_src.keyframe_scans_static_projected_ = Cproj_view

        def central_side():      # Session.cpp:393-427 C -> Q; constructGlobalNDMap :430-435, filterStrongND :1403-1411, weak -> strong :452-484; grids of :1447-1476
            C.scans_knn_coexist_, C.scans_knn_diff_ = ops.knn_partition(Q.map_global_curr_static_, C.keyframe_scans_static_projected_, C.keyframe_poses, k, thr)
            C.map_global_nd_ = ops.merge_voxel_batch([(C.scans_knn_diff_, C.keyframe_poses)], [], 0.05)[0]
            maps = (C.map_global_nd_, None, None)
            for _ in range(3):
                maps = self._removeOnceLD(maps, q_src, 2.5, 1, ops)
            C.map_global_nd_, C.map_global_nd_strong_, C.map_global_nd_weak_ = maps
            if ops.size(C.map_global_nd_strong_) != 0:
                add, new_weak = ops.knn_split(C.map_global_nd_strong_, C.map_global_nd_weak_, 2, 1.0)
                C.map_global_nd_strong_ = ops.concat([C.map_global_nd_strong_, add])
                C.map_global_nd_weak_ = new_weak
            has_strong_nd = ops.size(C.map_global_nd_strong_) != 0
            res = ops.merge_voxel_batch([(C.scans_knn_coexist_, C.keyframe_poses), (C.scans_knn_diff_, C.keyframe_poses)],
                                        [C.map_global_nd_weak_] + ([C.map_global_nd_strong_] if has_strong_nd else []), 0.05)
            self._union_c = o["union_map_centralside"] = res[0]
            o["nd_map"] = res[1]
            C.map_global_nd_weak_ = o["weak_nd_map"] = res[2]
            if has_strong_nd:
                C.map_global_nd_strong_ = o["strong_nd_map"] = res[3]

        def query_side():        # Session.cpp:393-427 Q -> C; constructGlobalPDMap :437-445, filterStrongPD :1395-1401; grids of :1445-1476
            Ql.scans_knn_coexist_, Ql.scans_knn_diff_ = lops.knn_partition(Cstat_view, Ql.keyframe_scans_static_projected_, Ql.keyframe_poses, k, thr)
            Ql.map_global_pd_ = lops.merge_voxel_batch([(Ql.scans_knn_diff_, Ql.keyframe_poses)], [], 0.05)[0]
            Ql.map_global_pd_orig_ = Ql.map_global_pd_
            maps = (Ql.map_global_pd_, None, None)
            for _ in range(3):
                maps = self._removeOnceLD(maps, c_src, 2.5, 0, lops)
            Ql.map_global_pd_, Ql.map_global_pd_strong_, Ql.map_global_pd_weak_ = maps
            res = lops.merge_voxel_batch([(Ql.scans_knn_coexist_, Ql.keyframe_poses), (Ql.scans_knn_diff_, Ql.keyframe_poses)],
                                         [Ql.map_global_pd_strong_, Ql.map_global_pd_weak_], 0.05)
            Ql.union_q_, Ql.pd_map_, Ql.map_global_pd_strong_, Ql.map_global_pd_weak_ = res
        self._fork(central_side, query_side)
        del Cstat_view, Cproj_view, Qproj_view
        q_src.keyframe_scans_static_projected_ = c_src.keyframe_scans_static_projected_ = None
        to_main = lambda x: L.give(x, M)      # noqa: E731
        Q.keyframe_scans_static_projected_ = to_main(Ql.keyframe_scans_static_projected_)
        Q.scans_knn_coexist_, Q.scans_knn_diff_ = to_main(Ql.scans_knn_coexist_), to_main(Ql.scans_knn_diff_)
        same_pd = Ql.map_global_pd_orig_ is Ql.map_global_pd_
        Q.map_global_pd_orig_ = to_main(Ql.map_global_pd_orig_)
        Q.map_global_pd_ = Q.map_global_pd_orig_ if same_pd else to_main(Ql.map_global_pd_)
        self._union_q = o["union_map_queryside"] = to_main(Ql.union_q_)
        o["pd_map"] = to_main(Ql.pd_map_)
        Q.map_global_pd_strong_ = o["strong_pd_map"] = to_main(Ql.map_global_pd_strong_)
        Q.map_global_pd_weak_ = o["weak_pd_map"] = to_main(Ql.map_global_pd_weak_)
        C.map_global_pd_, C.map_global_pd_orig_, C.map_global_pd_strong_ = Q.map_global_pd_, Q.map_global_pd_orig_, Q.map_global_pd_strong_   # :1435-1437
        self._tick("lanes_low_dynamic", t0)
        # ---- stage C
        t0 = time.perf_counter()
        self.updateCurrentMap()
        nd_strong = C.map_global_nd_strong_ if C.map_global_nd_strong_ is not None else ops.empty_cloud()
        views = [M.lend(x, L) for x in (C.map_global_pd_strong_, nd_strong, C.map_global_updated_strong_)]
        got = {}

        def main_side():         # Removerter.cpp:1551-1577 (updated, PD, weak ND) + :1540-1549
            a = self.kReprojectionAlpha
            C.keyframe_scans_updated_ = ops.reproject(C.map_global_updated_, C.keyframe_poses, a)
            C.keyframe_scans_pd_ = ops.reproject(C.map_global_pd_orig_, C.keyframe_poses, a)
            C.keyframe_scans_weak_nd_ = ops.reproject(C.map_global_nd_weak_, C.keyframe_poses, a)
            merged = ops.zip_concat(C.keyframe_scans_updated_, C.keyframe_scans_weak_nd_, C.keyframe_scans_pd_)
            C.keyframe_scans_updated_ = ops.voxel_scanset(merged, 0.05)

        def lane_side():         # the "strong" halves of :1551-1577
            a = self.kReprojectionAlpha
            got["strong_pd"] = lops.reproject(views[0], Cl_poses, a)
            got["strong_nd"] = lops.reproject(views[1], Cl_poses, a)
            got["updated_strong"] = lops.reproject(views[2], Cl_poses, a)
        self._fork(main_side, lane_side)
        del views
        C.keyframe_scans_strong_pd_, C.keyframe_scans_strong_nd_ = to_main(got["strong_pd"]), to_main(got["strong_nd"])
        C.keyframe_scans_updated_strong_ = to_main(got["updated_strong"])
        self._tick("lanes_step3", t0)
        self.outputs.update(central_map_static=C.map_global_curr_static_, central_map_dynamic=C.map_global_curr_dynamic_,
                            query_map_static=Q.map_global_curr_static_, query_map_dynamic=Q.map_global_curr_dynamic_)

    # ------------------------------------------------------------------ run()
    def run_steps_1_to_3(self):                         # Removerter.cpp:1664-1675 (file I/O excluded)
        self.removeHighDynamicPoints()
        self.parseStaticScansViaProjection()
        self.detectLowDynamicPoints()
        self.updateCurrentMap()
        self.parseUpdatedStaticScansViaProjection()
        self.parseLDScansViaProjection()
        self.updateScansScanwise()
        C, Q = self.central_sess_, self.query_sess_
        # Multi-GPU: every map is complete on every rank at this point (their inputs were all-gathered where a merge needed all
        # keyframes -- the "all-gather to assemble the final maps" of the north star).  The five per-keyframe outputs are one file
        # per keyframe (Removerter.cpp:1637-1650), so each rank keeps -- and writes -- those of its own keyframes; assembling them
        # everywhere as well is optional (and lazy: LazyScans.download() gathers on demand).  No-op on one GPU.
        if self.P.gather_scan_outputs:
            for name in ("keyframe_scans_updated_", "keyframe_scans_updated_strong_", "keyframe_scans_pd_", "keyframe_scans_strong_pd_",
                         "keyframe_scans_strong_nd_"):
                setattr(C, name, self.ops.materialize(getattr(C, name)))
        self.outputs.update(central_map_static=C.map_global_curr_static_, central_map_dynamic=C.map_global_curr_dynamic_,
                            query_map_static=Q.map_global_curr_static_, query_map_dynamic=Q.map_global_curr_dynamic_)

    def run(self):
        if self.lane_ops is not None and self.query_side is None and self._rankGroups() is None:
            return self.run_two_lanes()
        if self.central_scans_future is not None and (self.query_side is not None or self._rankGroups() is not None):
            self.central_sess_.keyframe_scans_ = self.central_scans_future()      # (the deferred hand-over is a single-context path)
            self.central_scans_future = None
        if self.query_side is None and self._rankGroups() is None and self.central_scans_future is None:
            self.makeGlobalMap()         # otherwise part of the two sessions' separate chains (removeHighDynamicPoints)
        self.run_steps_1_to_3()
