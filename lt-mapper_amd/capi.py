"""ctypes binding of the C ABI in include/ltm.h (libltm_hip.so).

This is plumbing for tests and bench.py: every call goes straight to the HIP library.  There is no
CPU fallback -- if the library is missing or no gfx950 device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libltm_hip.so")

_lib = None

LTM_OK = 0
ERR_NAMES = {0: "LTM_OK", -1: "LTM_E_INVALID", -2: "LTM_E_DEVICE", -3: "LTM_E_NOMEM", -4: "LTM_E_UNSUPPORTED"}


class LtmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class LtmConfig(C.Structure):
    _fields_ = [("vfov", C.c_float), ("hfov", C.c_float), ("lidar2base", C.c_double * 16),
                ("device", C.c_int), ("max_kf_batch", C.c_int)]


# name -> (restype, argtypes); kept in one table so tests can check it against include/ltm.h
_vp, _sz, _u64, _i, _f = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_float
_pu64 = C.POINTER(C.c_uint64)
_psz = C.POINTER(C.c_size_t)
SIGNATURES = {
    "ltm_abi_version": (_i, []),
    "ltm_create": (_i, [C.POINTER(LtmConfig), C.POINTER(_vp)]),
    "ltm_destroy": (None, [_vp]),
    "ltm_last_error": (C.c_char_p, [_vp]),
    "ltm_synchronize": (_i, [_vp]),
    "ltm_clear_caches": (_i, [_vp]),
    "ltm_stream": (_vp, [_vp]),
    "ltm_cloud_upload": (_i, [_vp, _vp, _sz, _sz, _pu64]),
    "ltm_cloud_from_device": (_i, [_vp, _vp, _sz, _pu64]),
    "ltm_cloud_size": (_i, [_vp, _u64, _psz]),
    "ltm_cloud_download": (_i, [_vp, _u64, _vp, _sz, _sz]),
    "ltm_cloud_device_ptr": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "ltm_cloud_clone": (_i, [_vp, _u64, _pu64]),
    "ltm_cloud_concat": (_i, [_vp, _pu64, _sz, _pu64]),
    "ltm_cloud_transform": (_i, [_vp, _u64, _vp, _vp, _pu64]),
    "ltm_cloud_select": (_i, [_vp, _u64, _vp, _sz, _pu64]),
    "ltm_scanset_keyframe": (_i, [_vp, _u64, _sz, _pu64]),
    "ltm_cloud_free": (_i, [_vp, _u64]),
    "ltm_cloud_alloc": (_i, [_vp, _sz, _pu64]),
    "ltm_scanset_alloc": (_i, [_vp, _pu64, _sz, _pu64]),
    "ltm_scanset_upload_begin": (_i, [_vp, _sz, _pu64]),
    "ltm_scanset_upload_chunk": (_i, [_vp, _u64, _vp, _sz, _pu64, _sz]),
    "ltm_scanset_upload_end": (_i, [_vp, _u64, _pu64]),
    "ltm_cloud_fetch_begin": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "ltm_scanset_fetch_begin": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "ltm_fetch_wait": (_i, [_vp, C.POINTER(_vp), _psz, C.POINTER(_pu64), _psz]),
    "ltm_fetch_release": (_i, [_vp, _vp]),
    "ltm_cloud_fetch_chunks_begin": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "ltm_scanset_fetch_chunks_begin": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "ltm_fetch_info": (_i, [_vp, _psz, C.POINTER(_pu64), _psz]),
    "ltm_fetch_next_chunk": (_i, [_vp, C.POINTER(_vp), _psz, _psz, _psz, _psz]),
    "ltm_fetch_chunk_done": (_i, [_vp, _vp]),
    "ltm_buffer_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "ltm_buffer_free": (_i, [_vp, _vp]),
    "ltm_buffer_fill": (_i, [_vp, _vp, _i, _sz]),
    "ltm_buffer_copy": (_i, [_vp, _vp, _vp, _sz, _i]),
    "ltm_scanset_upload": (_i, [_vp, _vp, _sz, _pu64, _sz, _pu64]),
    "ltm_scanset_from_device": (_i, [_vp, _vp, _pu64, _sz, _pu64]),
    "ltm_scanset_info": (_i, [_vp, _u64, _psz, _psz]),
    "ltm_scanset_offsets": (_i, [_vp, _u64, _pu64]),
    "ltm_scanset_download": (_i, [_vp, _u64, _vp, _sz, _sz]),
    "ltm_scanset_device_ptr": (_i, [_vp, _u64, C.POINTER(_vp)]),
    "ltm_scanset_as_cloud": (_i, [_vp, _u64, _pu64]),
    "ltm_scanset_concat": (_i, [_vp, _pu64, _sz, _pu64]),
    "ltm_scanset_zip_concat": (_i, [_vp, _u64, _u64, _u64, _pu64]),
    "ltm_scanset_free": (_i, [_vp, _u64]),
    "ltm_poses_create": (_i, [_vp, _sz, _vp, _vp, _pu64]),
    "ltm_poses_free": (_i, [_vp, _u64]),
    "ltm_inverse4x4": (_i, [_vp, _vp]),
    "ltm_preclean": (_i, [_vp, _u64, _f, _pu64]),
    "ltm_merge_to_global": (_i, [_vp, _u64, _u64, _pu64]),
    "ltm_voxel_centroid": (_i, [_vp, _u64, _f, _pu64]),
    "ltm_voxel_centroid_shard": (_i, [_vp, _u64, _f, C.c_uint32, C.c_uint32, _pu64]),
    "ltm_voxel_centroid_batch": (_i, [_vp, _sz, _pu64, C.POINTER(_f), _pu64]),
    "ltm_cloud_bbox": (_i, [_vp, _u64, C.POINTER(_f), C.POINTER(_f)]),
    "ltm_voxel_key_histogram": (_i, [_vp, _u64, C.POINTER(_f), C.POINTER(_f), _f, C.POINTER(C.c_uint32)]),
    "ltm_voxel_key_split": (_i, [_vp, _u64, C.POINTER(_f), C.POINTER(_f), _f, C.c_uint32, C.POINTER(C.c_uint32), _pu64]),
    "ltm_voxel_centroid_box": (_i, [_vp, _u64, C.POINTER(_f), C.POINTER(_f), _f, _pu64]),
    "ltm_voxel_centroid_scanset": (_i, [_vp, _u64, _f, _pu64]),
    "ltm_voxel_grid_scanset": (_i, [_vp, _u64, _f, _pu64]),
    "ltm_voxel_grid_scanset_begin": (_i, [_vp, _u64, _f, C.POINTER(_vp)]),
    "ltm_voxel_grid_scanset_end": (_i, [_vp, _vp, _pu64]),
    "ltm_scanset_prepare_range_images": (_i, [_vp, _u64, _sz, _sz, C.POINTER(_f), _sz]),
    "ltm_visibility_vote": (_i, [_vp, _u64, _u64, _u64, _sz, _sz, _f, _f, _i, _vp]),
    "ltm_partition_by_labels": (_i, [_vp, _u64, _vp, _pu64, _pu64]),
    "ltm_visibility_partition": (_i, [_vp, _u64, _u64, _u64, _f, _f, _i, _pu64, _pu64, _vp]),
    "ltm_lane_create": (_i, [_vp, C.POINTER(_vp)]),
    "ltm_lane_fence": (_i, [_vp, _vp]),
    "ltm_event_record": (_i, [_vp, C.POINTER(_vp)]),
    "ltm_event_wait": (_i, [_vp, _vp]),
    "ltm_event_destroy": (None, [_vp]),
    "ltm_cloud_lend": (_i, [_vp, _u64, _vp, _pu64]),
    "ltm_cloud_give": (_i, [_vp, _u64, _vp, _pu64]),
    "ltm_scanset_lend": (_i, [_vp, _u64, _vp, _pu64]),
    "ltm_scanset_give": (_i, [_vp, _u64, _vp, _pu64]),
    "ltm_reproject": (_i, [_vp, _u64, _u64, _sz, _sz, _f, _pu64]),
    "ltm_knn_partition": (_i, [_vp, _u64, _u64, _u64, _sz, _sz, _i, _f, _pu64, _pu64]),
    "ltm_knn_split_cloud": (_i, [_vp, _u64, _u64, _i, _f, _pu64, _pu64]),
    "ltm_debug_range_image": (_i, [_vp, _u64, _vp, _vp, _f, _vp, _vp]),
    "ltm_debug_viz_images": (_i, [_vp, _u64, _u64, _u64, _sz, _f, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ltm_debug_project": (_i, [_vp, _vp, _sz, _f, _vp, _vp]),
    "ltm_debug_elevation_fit": (_i, [_f, C.POINTER(_f), C.POINTER(C.c_double)]),
    "ltm_debug_pcl_sort_order": (_i, [C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint32), _i, C.POINTER(C.c_uint32)]),
    "ltm_debug_voxel_key_bits": (_i, [C.POINTER(_f), C.POINTER(_f), _f, _pu64, C.POINTER(C.c_uint), C.POINTER(C.c_double)]),
    "ltm_debug_selfcheck": (_i, [_vp, _pu64, C.POINTER(_i)]),
    "ltm_debug_cull_check": (_i, [_vp, _vp, _sz, _vp, _f, _pu64]),
    "ltm_debug_cull_stats": (_i, [_vp, _pu64, _pu64, _i]),
    "ltm_debug_cull_validation": (_i, [_vp, _pu64, _pu64]),
    "ltm_debug_occlusion_stats": (_i, [_vp, _pu64, _pu64, _pu64, _i]),
    "ltm_debug_voxel_stats": (_i, [_vp, _pu64, _pu64, _i]),
    "ltm_rimg_size": (None, [_f, _f, _f, C.POINTER(_i), C.POINTER(_i)]),
    "ltm_profile_enable": (_i, [_vp, _i]),
    "ltm_profile_reset": (_i, [_vp]),
    "ltm_profile_read_compulsory": (_i, [_vp, C.POINTER(C.c_double), _i]),
    "ltm_profile_read": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), _pu64, C.POINTER(C.c_double),
                              C.POINTER(C.c_double), _i]),
}


def load_library(path=None):
    """Load libltm_hip.so and declare every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # Load order of the HIP runtime: torch ships its own libamdhip64 and this library links the system one (same soname).  Whichever is
    # mapped first serves both; if OURS comes first and torch is imported later in the same process (build() followed by smoke()),
    # torch initialises against a runtime it was not built with and ltm_create then finds no usable device.  Importing torch first --
    # the order bench.py and the tests always had -- keeps one consistent runtime.  Without torch installed nothing changes.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `make hip` (or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def voxel_key_bits(mn, mx, leaf):
    """ltm_debug_voxel_key_bits: (kept bit count, kept mask, octree depth, lattice origin) of the voxel grid's sort key"""
    a = (_f * 3)(*[float(v) for v in mn]); b = (_f * 3)(*[float(v) for v in mx])
    mask, depth, fmin = _u64(), C.c_uint(), (C.c_double * 3)()
    n = load_library().ltm_debug_voxel_key_bits(a, b, float(leaf), C.byref(mask), C.byref(depth), fmin)
    if n < 0:
        raise ValueError("unsupported box / leaf")
    return n, mask.value, depth.value, np.array(list(fmin))


def inverse4x4(m):
    """ltm_inverse4x4: the library's Eigen::Matrix4d::inverse() restatement (needs no device)"""
    a = np.ascontiguousarray(m, dtype=np.float64).reshape(16)
    out = np.empty(16, dtype=np.float64)
    if load_library().ltm_inverse4x4(a.ctypes.data, out.ctypes.data) != 0:
        raise ValueError("singular or non-finite matrix")
    return out.reshape(4, 4)


def _np_pts(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        a = a.reshape(-1, 4)
    return a


def _mat16(m):
    m = np.ascontiguousarray(m, dtype=np.float64).reshape(-1)
    assert m.size == 16
    return m


class Context:
    """One ltm_ctx: one GPU, one HIP stream, all device memory behind handles."""

    def __init__(self, vfov=50.0, hfov=360.0, lidar2base=None, device=0, max_kf_batch=0):
        self.lib = load_library()
        cfg = LtmConfig()
        cfg.vfov, cfg.hfov, cfg.device, cfg.max_kf_batch = vfov, hfov, device, max_kf_batch
        l2b = _mat16(np.eye(4) if lidar2base is None else lidar2base)
        for i in range(16):
            cfg.lidar2base[i] = l2b[i]
        self.vfov, self.hfov, self.device = float(vfov), float(hfov), int(device)
        h = _vp()
        rc = self.lib.ltm_create(C.byref(cfg), C.byref(h))
        if rc != LTM_OK:
            raise LtmError(rc, "ltm_create failed (no usable gfx950 device?)")
        self.h = h

    def lane(self):
        """ltm_lane_create: a second context on this one's device (own stream, own pool, same configuration) for a second host thread; clouds and
        scan sets pass between the two without copies (lend / give)"""
        other = Context.__new__(Context)
        other.lib, other.vfov, other.hfov, other.device = self.lib, self.vfov, self.hfov, self.device
        h = _vp()
        rc = self.lib.ltm_lane_create(self.h, C.byref(h))
        if rc != LTM_OK:
            raise LtmError(rc, "ltm_lane_create failed")
        other.h = h
        return other

    def _pass(self, obj, to, give):
        kind = "cloud" if isinstance(obj, Cloud) else "scanset"
        out = _u64()
        self._ck(getattr(self.lib, f"ltm_{kind}_{'give' if give else 'lend'}")(self.h, obj.h, to.h, C.byref(out)))
        new = (Cloud if kind == "cloud" else ScanSet)(to, out.value)
        if give:
            obj.h = 0
        else:
            new._lender = obj      # keeps the owner's Python handle (and so the memory) alive as long as the view exists
        return new

    def lend(self, obj, to):
        """a borrowed view of `obj` (Cloud or ScanSet of this context) in context `to`, no copy; this context keeps ownership"""
        return self._pass(obj, to, False)

    def give(self, obj, to):
        """moves `obj` into context `to` (no copy); `obj` is invalid afterwards"""
        return self._pass(obj, to, True)

    def fence(self, before_next_of):
        """what is submitted to `before_next_of` from now on runs after everything submitted to this context so far"""
        self._ck(self.lib.ltm_lane_fence(self.h, before_next_of.h))

    def event_record(self):
        ev = _vp()
        self._ck(self.lib.ltm_event_record(self.h, C.byref(ev)))
        return Event(self.lib, ev)

    def event_wait(self, ev):
        self._ck(self.lib.ltm_event_wait(self.h, ev.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.ltm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != LTM_OK:
            raise LtmError(rc, self.lib.ltm_last_error(self.h).decode())

    # ---- clouds
    def upload(self, pts):
        a = _np_pts(pts)
        out = _u64()
        self._ck(self.lib.ltm_cloud_upload(self.h, a.ctypes.data, a.shape[0], 16, C.byref(out)))
        return Cloud(self, out.value)

    def cloud_from_device(self, dev_ptr, n):
        out = _u64()
        self._ck(self.lib.ltm_cloud_from_device(self.h, dev_ptr, n, C.byref(out)))
        return Cloud(self, out.value)

    def transform(self, cloud, T1=None, T2=None):
        """pcl::transformPointCloud once or twice (row-major 4x4 doubles)"""
        t1 = None if T1 is None else np.ascontiguousarray(T1, dtype=np.float64).reshape(16)
        t2 = None if T2 is None else np.ascontiguousarray(T2, dtype=np.float64).reshape(16)
        out = _u64()
        self._ck(self.lib.ltm_cloud_transform(self.h, cloud.h, None if t1 is None else t1.ctypes.data, None if t2 is None else t2.ctypes.data, C.byref(out)))
        return Cloud(self, out.value)

    def select(self, cloud, idx):
        """pcl::ExtractIndices (parsePointcloudSubsetUsingPtIdx, Removerter.cpp:933-946): out[j] = cloud[idx[j]]"""
        a = np.ascontiguousarray(idx, dtype=np.int32)
        out = _u64()
        self._ck(self.lib.ltm_cloud_select(self.h, cloud.h, a.ctypes.data, a.size, C.byref(out)))
        return Cloud(self, out.value)

    def scan_of_keyframe(self, scans, kf):
        out = _u64()
        self._ck(self.lib.ltm_scanset_keyframe(self.h, scans.h, kf, C.byref(out)))
        return Cloud(self, out.value)

    def concat(self, clouds):
        arr = (_u64 * len(clouds))(*[c.h for c in clouds])
        out = _u64()
        self._ck(self.lib.ltm_cloud_concat(self.h, arr, len(clouds), C.byref(out)))
        return Cloud(self, out.value)

    # ---- scan sets
    def upload_scans(self, pts, offsets):
        a = _np_pts(pts)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = _u64()
        self._ck(self.lib.ltm_scanset_upload(self.h, a.ctypes.data, 16, off.ctypes.data_as(_pu64), off.size - 1, C.byref(out)))
        return ScanSet(self, out.value)

    def scans_from_device(self, dev_ptr, offsets):
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = _u64()
        self._ck(self.lib.ltm_scanset_from_device(self.h, dev_ptr, off.ctypes.data_as(_pu64), off.size - 1, C.byref(out)))
        return ScanSet(self, out.value)

    def concat_scansets(self, sets):
        arr = (_u64 * len(sets))(*[s.h for s in sets])
        out = _u64()
        self._ck(self.lib.ltm_scanset_concat(self.h, arr, len(sets), C.byref(out)))
        return ScanSet(self, out.value)

    def zip_concat(self, a, b, c=None):
        out = _u64()
        self._ck(self.lib.ltm_scanset_zip_concat(self.h, a.h, b.h, c.h if c is not None else 0, C.byref(out)))
        return ScanSet(self, out.value)

    # ---- poses
    def poses(self, poses, inv=None):
        p = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
        iv = None if inv is None else np.ascontiguousarray(inv, dtype=np.float64).reshape(-1, 16)
        out = _u64()
        self._ck(self.lib.ltm_poses_create(self.h, p.shape[0], p.ctypes.data, None if iv is None else iv.ctypes.data, C.byref(out)))
        if iv is None:      # the host copy holds what the library computed for itself (ltm_inverse4x4)
            iv = np.stack([inverse4x4(m).reshape(16) for m in p]) if p.shape[0] else p.copy()
        return Poses(self, out.value, p.shape[0], p.copy(), iv.copy())

    # ---- stages
    def preclean(self, scans, radius):
        out = _u64()
        self._ck(self.lib.ltm_preclean(self.h, scans.h, radius, C.byref(out)))
        return ScanSet(self, out.value)

    def merge_to_global(self, scans, poses):
        out = _u64()
        self._ck(self.lib.ltm_merge_to_global(self.h, scans.h, poses.h, C.byref(out)))
        return Cloud(self, out.value)

    def voxel_centroid(self, cloud, leaf):
        out = _u64()
        self._ck(self.lib.ltm_voxel_centroid(self.h, cloud.h, leaf, C.byref(out)))
        return Cloud(self, out.value)

    def voxel_centroid_batch(self, clouds, leafs):
        n = len(clouds)
        ins = (_u64 * n)(*[c.h for c in clouds])
        lf = (_f * n)(*[float(x) for x in leafs])
        outs = (_u64 * n)()
        self._ck(self.lib.ltm_voxel_centroid_batch(self.h, n, ins, lf, outs))
        return [Cloud(self, outs[k]) for k in range(n)]

    # ---- key-range exchange (include/ltm.h: ltm_cloud_bbox ... ltm_voxel_centroid_box)
    @staticmethod
    def _f3(v):
        return (_f * 3)(*[float(x) for x in v])

    def bbox(self, cloud):
        mn, mx = (_f * 3)(), (_f * 3)()
        self._ck(self.lib.ltm_cloud_bbox(self.h, cloud.h, mn, mx))
        return np.array(mn[:], np.float32), np.array(mx[:], np.float32)

    def voxel_key_histogram(self, cloud, mn, mx, leaf):
        hist = (C.c_uint32 * 4096)()
        self._ck(self.lib.ltm_voxel_key_histogram(self.h, cloud.h, self._f3(mn), self._f3(mx), leaf, hist))
        return np.frombuffer(hist, dtype=np.uint32).copy()

    def voxel_key_split(self, cloud, mn, mx, leaf, cuts):
        n = len(cuts) - 1
        cb = (C.c_uint32 * (n + 1))(*[int(x) for x in cuts])
        outs = (_u64 * n)()
        self._ck(self.lib.ltm_voxel_key_split(self.h, cloud.h, self._f3(mn), self._f3(mx), leaf, n, cb, outs))
        return [Cloud(self, outs[k]) for k in range(n)]

    def voxel_centroid_box(self, cloud, mn, mx, leaf):
        out = _u64()
        self._ck(self.lib.ltm_voxel_centroid_box(self.h, cloud.h, self._f3(mn), self._f3(mx), leaf, C.byref(out)))
        return Cloud(self, out.value)

    def voxel_centroid_shard(self, cloud, leaf, shard, n_shards):
        out = _u64()
        self._ck(self.lib.ltm_voxel_centroid_shard(self.h, cloud.h, leaf, shard, n_shards, C.byref(out)))
        return Cloud(self, out.value)

    def viz_images(self, cmap, scans, poses, kf, alpha, mode=0, range_axis=(0.0, 10.0), diff_axis=(0.0, 0.5)):
        """the four RViz images of keyframe `kf` (Removerter.cpp:580-585) as (rows, cols, 3) BGR8 arrays"""
        rows, cols = self.rimg_size(alpha)
        out = {k: np.empty((rows, cols, 3), dtype=np.uint8) for k in ("scan", "map", "diff", "ptidx")}
        self._ck(self.lib.ltm_debug_viz_images(self.h, cmap.h, scans.h, poses.h, kf, alpha, mode, range_axis[0], range_axis[1],
                                               diff_axis[0], diff_axis[1], out["scan"].ctypes.data, out["map"].ctypes.data,
                                               out["diff"].ctypes.data, out["ptidx"].ctypes.data))
        return out

    def voxel_centroid_scanset(self, scans, leaf):
        out = _u64()
        self._ck(self.lib.ltm_voxel_centroid_scanset(self.h, scans.h, leaf, C.byref(out)))
        return ScanSet(self, out.value)

    def voxel_grid_scanset(self, scans, leaf):
        out = _u64()
        self._ck(self.lib.ltm_voxel_grid_scanset(self.h, scans.h, leaf, C.byref(out)))
        return ScanSet(self, out.value)

    def voxel_grid_scanset_begin(self, scans, leaf):
        """first half of voxel_grid_scanset: returns a ticket at once (keys on their way to the host threads); `scans` must stay alive until
        voxel_grid_scanset_end(ticket).  Work submitted in between runs beside the transfer and the host sort."""
        t = C.c_void_p()
        self._ck(self.lib.ltm_voxel_grid_scanset_begin(self.h, scans.h, leaf, C.byref(t)))
        return VgsTicket(self, t, scans)

    def voxel_grid_scanset_end(self, ticket):
        return ticket.end()

    def prepare_scan_images(self, scans, alphas, kf_begin=0, kf_end=None):
        """scan range images of all the listed resolutions in one pass over the scans (kept for the votes that follow)"""
        a = (_f * len(alphas))(*[float(x) for x in alphas])
        self._ck(self.lib.ltm_scanset_prepare_range_images(self.h, scans.h, kf_begin, scans.n_kf if kf_end is None else kf_end, a, len(alphas)))

    def visibility_vote(self, cmap, scans, poses, kf_begin, kf_end, alpha, thr, mode, labels_dev_ptr):
        self._ck(self.lib.ltm_visibility_vote(self.h, cmap.h, scans.h, poses.h, kf_begin, kf_end, alpha, thr, mode, labels_dev_ptr))

    def partition_by_labels(self, cmap, labels_dev_ptr):
        k, f = _u64(), _u64()
        self._ck(self.lib.ltm_partition_by_labels(self.h, cmap.h, labels_dev_ptr, C.byref(k), C.byref(f)))
        return Cloud(self, k.value), Cloud(self, f.value)

    def visibility_partition(self, cmap, scans, poses, alpha, thr=0.1, mode=0, want_labels=False):
        k, f = _u64(), _u64()
        labels = np.zeros(len(cmap), dtype=np.uint8) if want_labels else None
        self._ck(self.lib.ltm_visibility_partition(self.h, cmap.h, scans.h, poses.h, alpha, thr, mode, C.byref(k), C.byref(f),
                                                   labels.ctypes.data if want_labels else None))
        kept, flagged = Cloud(self, k.value), Cloud(self, f.value)
        return (kept, flagged, labels) if want_labels else (kept, flagged)

    def reproject(self, cmap, poses, alpha=3.0, kf_begin=0, kf_end=None):
        out = _u64()
        self._ck(self.lib.ltm_reproject(self.h, cmap.h, poses.h, kf_begin, poses.n if kf_end is None else kf_end, alpha, C.byref(out)))
        return ScanSet(self, out.value)

    def knn_partition(self, target, scans, poses, k, thr, kf_begin=0, kf_end=None):
        co, di = _u64(), _u64()
        self._ck(self.lib.ltm_knn_partition(self.h, target.h, scans.h, poses.h, kf_begin, poses.n if kf_end is None else kf_end,
                                            k, thr, C.byref(co), C.byref(di)))
        return ScanSet(self, co.value), ScanSet(self, di.value)

    def knn_split_cloud(self, target, query, k, thr):
        near, far = _u64(), _u64()
        self._ck(self.lib.ltm_knn_split_cloud(self.h, target.h, query.h, k, thr, C.byref(near), C.byref(far)))
        return Cloud(self, near.value), Cloud(self, far.value)

    # ---- debug / parity
    def rimg_size(self, alpha):
        r, c = _i(), _i()
        self.lib.ltm_rimg_size(self.vfov, self.hfov, alpha, C.byref(r), C.byref(c))
        return r.value, c.value

    def debug_range_image(self, cloud, alpha, T1=None, T2=None, want_idx=True):
        R, Cc = self.rimg_size(alpha)
        rimg = np.empty((R, Cc), dtype=np.float32)
        idx = np.empty((R, Cc), dtype=np.int32) if want_idx else None
        t1 = None if T1 is None else _mat16(T1)
        t2 = None if T2 is None else _mat16(T2)
        self._ck(self.lib.ltm_debug_range_image(self.h, cloud.h, None if t1 is None else t1.ctypes.data,
                                                None if t2 is None else t2.ctypes.data, alpha, rimg.ctypes.data,
                                                idx.ctypes.data if want_idx else None))
        return rimg, idx

    def debug_project(self, xyz, alpha):
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        sph = np.empty((a.shape[0], 3), dtype=np.float32)
        rc = np.empty((a.shape[0], 2), dtype=np.int32)
        self._ck(self.lib.ltm_debug_project(self.h, a.ctypes.data, a.shape[0], alpha, sph.ctypes.data, rc.ctypes.data))
        return sph, rc

    def cull_check(self, xyz, alpha, inv_pose=None):
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        t = None if inv_pose is None else _mat16(inv_pose)
        v = _u64()
        self._ck(self.lib.ltm_debug_cull_check(self.h, a.ctypes.data, a.shape[0], None if t is None else t.ctypes.data, alpha, C.byref(v)))
        return int(v.value)

    def occlusion_stats(self, reset=False):
        a, b, d = _u64(), _u64(), _u64()
        self._ck(self.lib.ltm_debug_occlusion_stats(self.h, C.byref(a), C.byref(b), C.byref(d), 1 if reset else 0))
        return a.value, b.value, d.value

    def voxel_stats(self, reset=False):
        a, b = _u64(), _u64()
        self._ck(self.lib.ltm_debug_voxel_stats(self.h, C.byref(a), C.byref(b), 1 if reset else 0))
        return int(a.value), int(b.value)

    def cull_validation(self):
        """(image shapes whose bounded-error projection was validated on first use, shapes that failed and fell back to the exact kernels)"""
        a, b = _u64(), _u64()
        self._ck(self.lib.ltm_debug_cull_validation(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def cull_stats(self, reset=True):
        a, b = _u64(), _u64()
        self._ck(self.lib.ltm_debug_cull_stats(self.h, C.byref(a), C.byref(b), 1 if reset else 0))
        return int(a.value), int(b.value)

    def selfcheck(self):
        m = (C.c_uint64 * 3)()
        on = _i()
        self._ck(self.lib.ltm_debug_selfcheck(self.h, m, C.byref(on)))
        return [int(x) for x in m], bool(on.value)

    # ---- measurement
    def synchronize(self):
        self._ck(self.lib.ltm_synchronize(self.h))

    def clear_caches(self):
        self._ck(self.lib.ltm_clear_caches(self.h))

    def stream(self):
        return self.lib.ltm_stream(self.h)

    def profile_enable(self, on=True):
        self._ck(self.lib.ltm_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._ck(self.lib.ltm_profile_reset(self.h))

    def profile_read(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        launches = (C.c_uint64 * cap)()
        units = (C.c_double * cap)()
        nbytes = (C.c_double * cap)()
        n = self.lib.ltm_profile_read(self.h, names, ms, launches, units, nbytes, cap)
        if n < 0:
            self._ck(n)
        nb_c = (C.c_double * cap)()
        self._ck(min(self.lib.ltm_profile_read_compulsory(self.h, nb_c, cap), 0))
        return {names[i].decode(): dict(ms=ms[i], launches=int(launches[i]), units=units[i], bytes=nbytes[i], bytes_c=nb_c[i]) for i in range(min(n, cap))}


class VgsTicket:
    """an open ltm_voxel_grid_scanset_begin: end() finishes the grid; a ticket dropped without it (an exception between the halves) is ended and its result
    freed when the object goes (and ltm_destroy joins whatever is still open), so the coordinator thread never outlives its buffers"""

    def __init__(self, ctx, t, scans):
        self.ctx, self.t, self.scans = ctx, t, scans

    def end(self):
        t, self.t = self.t, None
        if t is None:
            raise LtmError(-1, "the ticket was ended already")
        out = _u64()
        self.ctx._ck(self.ctx.lib.ltm_voxel_grid_scanset_end(self.ctx.h, t, C.byref(out)))
        self.scans = None
        return ScanSet(self.ctx, out.value)

    def __del__(self):
        try:
            if self.t is not None and self.ctx.h:
                self.end().free()
        except Exception:
            pass


class Event:
    """ltm_event: a point of one context's stream that other contexts can wait for (any number of times)"""

    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def __del__(self):
        try:
            if self.h:
                self.lib.ltm_event_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _Handle:
    _free = None

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def free(self):
        if self.h and self.ctx.h:
            getattr(self.ctx.lib, self._free)(self.ctx.h, self.h)
        self.h = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Cloud(_Handle):
    _free = "ltm_cloud_free"

    def __len__(self):
        n = C.c_size_t()
        self.ctx._ck(self.ctx.lib.ltm_cloud_size(self.ctx.h, self.h, C.byref(n)))
        return n.value

    def download(self):
        n = len(self)
        out = np.empty((n, 4), dtype=np.float32)
        self.ctx._ck(self.ctx.lib.ltm_cloud_download(self.ctx.h, self.h, out.ctypes.data, n, 16))
        return out

    def clone(self):
        out = _u64()
        self.ctx._ck(self.ctx.lib.ltm_cloud_clone(self.ctx.h, self.h, C.byref(out)))
        return Cloud(self.ctx, out.value)

    def device_ptr(self):
        p = _vp()
        self.ctx._ck(self.ctx.lib.ltm_cloud_device_ptr(self.ctx.h, self.h, C.byref(p)))
        return p.value


class ScanSet(_Handle):
    _free = "ltm_scanset_free"

    def info(self):
        a, b = C.c_size_t(), C.c_size_t()
        self.ctx._ck(self.ctx.lib.ltm_scanset_info(self.ctx.h, self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def n_kf(self):
        return self.info()[0]

    def offsets(self):
        nk, _ = self.info()
        off = np.empty(nk + 1, dtype=np.uint64)
        self.ctx._ck(self.ctx.lib.ltm_scanset_offsets(self.ctx.h, self.h, off.ctypes.data_as(_pu64)))
        return off

    def download(self):
        _, n = self.info()
        out = np.empty((n, 4), dtype=np.float32)
        self.ctx._ck(self.ctx.lib.ltm_scanset_download(self.ctx.h, self.h, out.ctypes.data, n, 16))
        return out, self.offsets()

    def as_cloud(self):
        out = _u64()
        self.ctx._ck(self.ctx.lib.ltm_scanset_as_cloud(self.ctx.h, self.h, C.byref(out)))
        return Cloud(self.ctx, out.value)

    def device_ptr(self):
        p = _vp()
        self.ctx._ck(self.ctx.lib.ltm_scanset_device_ptr(self.ctx.h, self.h, C.byref(p)))
        return p.value


class Poses(_Handle):
    _free = "ltm_poses_free"

    def __init__(self, ctx, h, n, host_poses=None, host_inv=None):
        super().__init__(ctx, h)
        self.n = n
        self.host_poses, self.host_inv = host_poses, host_inv     # kept so that keyframe shards can be sliced off
