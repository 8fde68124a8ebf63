"""TEST INFRASTRUCTURE: the stage interface of ltmapper_amd.removerter.HipOps implemented with the CPU oracle, so the
host logic (Removerter orchestration, dist.ShardedOps exchange) can be exercised on CPU / gloo.  Never shipped."""
import numpy as np
import torch

from oracle import oracle_py as orc


class OPoses:
    def __init__(self, poses, inv):
        self.poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
        self.inv = np.ascontiguousarray(inv, dtype=np.float64).reshape(-1, 16)
        self.n = self.poses.shape[0]


class OScans:
    def __init__(self, pts, off):
        self.pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4)
        self.off = np.ascontiguousarray(off, dtype=np.uint64)

    def download(self):
        return self.pts, self.off

    def info(self):
        return len(self.off) - 1, int(self.off[-1])


class OCloud(np.ndarray):
    def download(self):
        return np.asarray(self)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4).view(OCloud)


class OracleOps:
    def __init__(self, vfov=50.0, hfov=360.0, l2b=None, threads=1):
        self.vfov, self.hfov = vfov, hfov
        self.l2b = np.eye(4) if l2b is None else np.asarray(l2b, dtype=np.float64)
        self.b2l = np.eye(4) if l2b is None else orc.inverse4x4(self.l2b)
        self.threads = threads

    def clone(self, c): return _c(np.array(c))
    def concat(self, cs): return _c(np.concatenate([np.asarray(c).reshape(-1, 4) for c in cs]))
    def size(self, c): return len(c)
    def empty_cloud(self): return _c(np.zeros((0, 4), np.float32))
    def sync(self): pass
    def merge_to_global(self, s, p): return _c(orc.merge_to_global(s.pts, s.off, p.poses, self.l2b))
    def voxel(self, c, leaf): return _c(orc.voxel_centroid(c, leaf))
    def voxel_batch(self, cs, leaf): return [self.voxel(c, leaf) for c in cs]

    def voxel_shard(self, c, leaf, shard, n_shards):
        # any contiguous split of the full output has the property ShardedOps relies on (concatenation == unsharded)
        full = orc.voxel_centroid(c, leaf)
        return _c(full[(len(full) * shard) // n_shards:(len(full) * (shard + 1)) // n_shards])

    # ---- key-range exchange (dist.ShardedOps.merge_voxel): box, 4096-bin key histogram, order-preserving split, grid under a given box
    def bbox(self, c):
        a = np.asarray(c).reshape(-1, 4)
        if len(a) == 0:
            return np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        return a[:, :3].min(0), a[:, :3].max(0)

    def _bins(self, c, mn, mx, leaf):
        keys, depth = orc.voxel_keys_box(np.asarray(c).reshape(-1, 4), mn, mx, leaf)
        return (keys >> np.uint64(max(3 * depth - 12, 0))).astype(np.int64)        # top 12 bits of the Morton code: a prefix, so no voxel straddles a bin

    def voxel_key_histogram(self, c, mn, mx, leaf):
        if len(c) == 0:
            return np.zeros(4096, np.uint32)
        return np.bincount(self._bins(c, mn, mx, leaf), minlength=4096).astype(np.uint32)

    def voxel_key_split(self, c, mn, mx, leaf, cuts):
        a = np.asarray(c).reshape(-1, 4)
        if len(a) == 0:
            return [self.empty_cloud() for _ in range(len(cuts) - 1)]
        b = self._bins(c, mn, mx, leaf)
        return [_c(a[(b >= cuts[r]) & (b < cuts[r + 1])]) for r in range(len(cuts) - 1)]

    def voxel_box(self, c, mn, mx, leaf): return _c(orc.voxel_centroid_box(c, mn, mx, leaf)) if len(c) else self.empty_cloud()

    def cloud_to_tensor(self, c): return torch.from_numpy(np.ascontiguousarray(np.asarray(c), dtype=np.float32).reshape(-1, 4))
    def cloud_from_tensor(self, t): return _c(t.numpy())

    def voxel_scanset(self, s, leaf):
        parts = [orc.voxel_centroid(s.pts[int(s.off[k]):int(s.off[k + 1])], leaf) for k in range(len(s.off) - 1)]
        return self._pack(parts)

    def voxel_grid_scanset(self, s, leaf):      # the loader's pcl::VoxelGrid per keyframe (Session.cpp:284-289)
        return self._pack([orc.voxel_grid(s.pts[int(s.off[k]):int(s.off[k + 1])], leaf) for k in range(len(s.off) - 1)])

    # the two halves of the cascade hand-over (ltm_voxel_grid_scanset_begin / _end): here the grid simply runs in the second half, which is enough to put
    # cascade.run_cascade's deferred path and Removerter._queryThenCentral -- pure host logic -- under a CPU test
    supports_deferred_grid = True

    def voxel_grid_scanset_begin(self, s, leaf): return (s, leaf)

    def voxel_grid_scanset_end(self, ticket): return self.voxel_grid_scanset(*ticket)

    def preclean(self, s, radius):              # Session.cpp:506-533
        return self._pack([orc.preclean(s.pts[int(s.off[k]):int(s.off[k + 1])], radius) for k in range(len(s.off) - 1)])

    @staticmethod
    def _pack(parts):
        off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
        pts = np.concatenate(parts) if parts else np.zeros((0, 4), np.float32)
        return OScans(pts, off)

    # ---- sharding pieces
    def n_keyframes(self, p): return p.n
    def poses_slice(self, p, kb, ke): return OPoses(p.poses[kb:ke], p.inv[kb:ke])
    def materialize(self, scans): return scans
    def new_labels(self, n): return torch.zeros(n, dtype=torch.uint8)

    def vote(self, cmap, scans, poses, kb, ke, alpha, thr, mode, labels):
        if labels.numel():
            lab = labels.numpy()
            orc.vote_labels(cmap, scans.pts, scans.off, poses.inv, self.b2l, self.vfov, self.hfov, alpha, thr, mode, kb, ke, self.threads, lab)

    def partition(self, cmap, labels):
        lab = labels.numpy().astype(bool)
        m = np.asarray(cmap).reshape(-1, 4)
        return _c(m[~lab]), _c(m[lab])

    def vote_partition(self, cmap, scans, poses, alpha, thr, mode):
        labels = self.new_labels(len(cmap))
        self.vote(cmap, scans, poses, 0, poses.n, alpha, thr, mode, labels)
        return self.partition(cmap, labels)

    def reproject_range(self, cmap, poses, alpha, kb, ke):
        pts, off = orc.reproject(cmap, poses.inv, self.b2l, self.vfov, self.hfov, alpha, kb, ke, self.threads)
        return OScans(pts, off)

    def reproject(self, cmap, poses, alpha): return self.reproject_range(cmap, poses, alpha, 0, poses.n)

    def knn_partition_range(self, target, scans, poses, k, thr, kb, ke):
        co, loc = orc.knn_labels(target, scans.pts, scans.off, poses.poses, poses.inv, self.b2l, k, thr, kb, ke, self.threads)
        cos, dis = [], []
        for kf in range(kb, ke):
            a, b = int(scans.off[kf]), int(scans.off[kf + 1])
            m = co[a:b] == 1
            cos.append(loc[a:b][m]); dis.append(loc[a:b][~m])
        return self._pack(cos), self._pack(dis)

    def knn_partition(self, target, scans, poses, k, thr): return self.knn_partition_range(target, scans, poses, k, thr, 0, poses.n)

    def knn_split(self, target, query, k, thr):
        near = orc.knn_split(target, query, k, thr).astype(bool)
        q = np.asarray(query).reshape(-1, 4)
        return _c(q[near]), _c(q[~near])

    def zip_concat(self, a, b, c):
        parts = []
        for k in range(len(a.off) - 1):
            parts.append(np.concatenate([s.pts[int(s.off[k]):int(s.off[k + 1])] for s in (a, b, c) if s is not None]))
        return self._pack(parts)

    def concat_scansets(self, sets):
        parts = []
        for s in sets:
            parts += [s.pts[int(s.off[k]):int(s.off[k + 1])] for k in range(len(s.off) - 1)]
        return self._pack(parts)

    def scanset_to_tensors(self, s): return torch.from_numpy(np.array(s.pts)), s.off
    def scanset_from_tensors(self, pts, off): return OScans(pts.numpy(), np.asarray(off, dtype=np.uint64))
