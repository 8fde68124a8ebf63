"""Keyframe sharding over GPUs (north_star: "keyframes shard naturally across the 8 GPUs of one
node with an RCCL all-gather over xGMI to assemble the final maps").

One process per GPU.  The session maps are replicated (a 10 M point map is 160 MB of 288 GB); every
per-keyframe stage runs on this rank's contiguous block of keyframes and exactly one exchange follows it:

  visibility vote   -> MAX all-reduce of the M-byte label mask (the union of Removerter.cpp:589-590), after which
                       every rank performs the same deterministic partition + voxel grid (replicated, no broadcast)
  reprojection/kNN  -> all-gather of the per-keyframe clouds (sizes, then padded payload), reassembled in
                       keyframe order so every rank holds the full scan set

Everything else (merge, voxel grids, the tiny weak->strong ND split) is replicated.  The collectives are
torch.distributed calls (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests), so the same code is exercised
on CPU with world_size 2.  `ops` is any object with the stage interface of removerter.HipOps.
"""
import torch


def shard_range(n, rank, world):
    """contiguous block of keyframes [kb, ke) of rank `rank`"""
    return (n * rank) // world, (n * (rank + 1)) // world


class ShardedOps:
    def __init__(self, ops, dist, rank, world, group=None):
        self.ops, self.dist, self.rank, self.world, self.group = ops, dist, rank, world, group

    def __getattr__(self, name):           # replicated stages are forwarded untouched
        return getattr(self.ops, name)

    # ---- vote: local keyframes, label union across ranks, replicated partition
    def vote_partition(self, cmap, scans, poses, alpha, thr, mode):
        n = self.ops.n_keyframes(poses)
        kb, ke = shard_range(n, self.rank, self.world)
        labels = self.ops.new_labels(self.ops.size(cmap))
        self.ops.vote(cmap, scans, poses, kb, ke, alpha, thr, mode, labels)
        if labels.numel():
            self.dist.all_reduce(labels, op=self.dist.ReduceOp.MAX, group=self.group)
        return self.ops.partition(cmap, labels)

    # ---- per-keyframe outputs: all-gather in keyframe order
    def _allgather_scanset(self, local):
        pts, off = self.ops.scanset_to_tensors(local)
        counts = [None] * self.world
        self.dist.all_gather_object(counts, [int(x) for x in off], group=self.group)
        sizes = [c[-1] for c in counts]
        cap = max(max(sizes), 1)
        pad = torch.zeros((cap, 4), dtype=torch.float32, device=pts.device)
        pad[: pts.shape[0]] = pts
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad, group=self.group)
        parts = [self.ops.scanset_from_tensors(bufs[r][: sizes[r]].contiguous(), counts[r]) for r in range(self.world)]
        return self.ops.concat_scansets(parts)

    def reproject(self, cmap, poses, alpha):
        kb, ke = shard_range(self.ops.n_keyframes(poses), self.rank, self.world)
        return self._allgather_scanset(self.ops.reproject_range(cmap, poses, alpha, kb, ke))

    def knn_partition(self, target, scans, poses, k, thr):
        kb, ke = shard_range(self.ops.n_keyframes(poses), self.rank, self.world)
        co, di = self.ops.knn_partition_range(target, scans, poses, k, thr, kb, ke)
        return self._allgather_scanset(co), self._allgather_scanset(di)
