"""TEST INFRASTRUCTURE: ctypes binding of oracle/libltm_oracle.so (the CPU restatement of the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
package (lt-mapper_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libltm_oracle.so")
_lib = None

_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float


class OrcParams(C.Structure):
    _fields_ = [("vfov", _f), ("hfov", _f), ("k", _i), ("knn_thr", _f), ("voxel", _f), ("lidar2base", C.c_double * 16),
                ("use_self_removert", _i), ("n_res", _i), ("res_list", _f * 8), ("repeat", _i), ("threads", _i),
                ("skip_hd_knn", _i), ("kf_sample_stride", _i)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libltm_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_atan2f.restype = _f
        L.orc_atan2f.argtypes = [_f, _f]
        L.orc_rad2deg.restype = _f
        L.orc_rad2deg.argtypes = [_f]
        L.orc_voxel_centroid.restype = _sz
        L.orc_voxel_centroid_box.restype = _sz
        L.orc_voxel_keys_box.restype = _i
        L.orc_reproject.restype = _sz
        L.orc_preclean.restype = _sz
        L.orc_voxel_grid.restype = _sz
        L.orc_pipeline_run.restype = _vp
        L.orc_inverse4x4.restype = _i
        L.orc_inverse4x4_variant.restype = _i
        _lib = L
    return _lib


def _pts(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)


def _m(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(-1, 16) if n is None else a.reshape(n, 16)


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def atan2f(y, x):
    y = np.ascontiguousarray(y, dtype=np.float32); x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(y)
    lib().orc_atan2f_array(_p(y), _p(x), _p(out), _sz(y.size))
    return out


def rimg_size(vfov, hfov, alpha):
    r, c = _i(), _i()
    lib().orc_rimg_size(_f(vfov), _f(hfov), _f(alpha), C.byref(r), C.byref(c))
    return r.value, c.value


def pixel(xyz, vfov, hfov, rows, cols):
    a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    rc = np.empty((a.shape[0], 2), dtype=np.int32); rng = np.empty(a.shape[0], dtype=np.float32)
    r, c, g = _i(), _i(), _f()
    for j in range(a.shape[0]):
        lib().orc_pixel(_p(a[j]), _f(vfov), _f(hfov), _i(rows), _i(cols), C.byref(r), C.byref(c), C.byref(g))
        rc[j] = (r.value, c.value); rng[j] = g.value
    return rc, rng


def cart2sph(xyz):
    a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(a)
    for j in range(a.shape[0]):
        lib().orc_cart2sph(_p(a[j]), _p(out[j]))
    return out


def transform(T, pts):
    a = _pts(pts); out = np.empty_like(a); t = _m(T, 1)
    lib().orc_transform(_p(t), _p(a), _p(out), _sz(a.shape[0]))
    return out


def inverse4x4(m, variant=0):
    """variant 0: Eigen 3.3.7 SSE2 operation order (the oracle's inverse); 1: cofactor expansion; 2: Gauss-Jordan (sensitivity test)"""
    a = _m(m, 1); out = np.empty_like(a)
    rc = lib().orc_inverse4x4_variant(_p(a), _p(out), C.c_int(int(variant)))
    assert rc == 0
    return out.reshape(4, 4)


def set_atan2f_perturbation(ppm, seed=0):
    """sensitivity experiments only: move `ppm` per million of the oracle's atan2f results by one ulp; 0 switches it off"""
    lib().orc_set_atan2f_perturbation(C.c_uint(int(ppm)), C.c_uint64(int(seed)))


def inverse_poses(poses, variant=0):
    """(n, 16) row-major poses -> (n, 16) inverses, as Session::loadSessionInfo computes them (Session.cpp:109-110)"""
    p = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)
    return np.array([inverse4x4(m, variant).reshape(16) for m in p]).reshape(-1, 16)


def range_image(pts, vfov, hfov, rows, cols, T1=None, T2=None, want_idx=True):
    a = _pts(pts)
    rimg = np.empty((rows, cols), dtype=np.float32)
    idx = np.empty((rows, cols), dtype=np.int32) if want_idx else None
    t1 = None if T1 is None else _m(T1, 1); t2 = None if T2 is None else _m(T2, 1)
    lib().orc_range_image(_p(a), _sz(a.shape[0]), _p(t1), _p(t2), _f(vfov), _f(hfov), _i(rows), _i(cols), _p(rimg), _p(idx))
    return rimg, idx


def vote_labels(cmap, scans, offsets, inv_poses, b2l, vfov, hfov, alpha, thr=0.1, mode=0, kf_begin=0, kf_end=None, threads=1, labels=None):
    m = _pts(cmap); s = _pts(scans); off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ip = _m(inv_poses); b = _m(b2l, 1)
    nkf = off.size - 1
    if labels is None:
        labels = np.zeros(m.shape[0], dtype=np.uint8)
    lib().orc_vote_labels(_p(m), _sz(m.shape[0]), _p(s), _p(off), _sz(nkf), _p(ip), _p(b), _f(vfov), _f(hfov), _f(alpha), _f(thr),
                          _i(mode), _sz(kf_begin), _sz(nkf if kf_end is None else kf_end), _i(threads), _p(labels))
    return labels


def jet_lut():
    lut = np.empty((256, 3), dtype=np.uint8)
    lib().orc_jet_lut(_p(lut))
    return lut


def colormap(img, cmin, cmax):
    """convertColorMappedImg (utility.h:114-127): float32 or int32 image -> BGR8"""
    a = np.ascontiguousarray(img)
    out = np.empty(a.shape + (3,), dtype=np.uint8)
    if a.dtype == np.int32:
        lib().orc_colormap_i32(_p(a), _sz(a.size), _f(cmin), _f(cmax), _p(out))
    else:
        a = np.ascontiguousarray(a, dtype=np.float32)
        lib().orc_colormap_f32(_p(a), _sz(a.size), _f(cmin), _f(cmax), _p(out))
    return out


def voxel_centroid(pts, leaf):
    a = _pts(pts)
    out = np.empty((max(a.shape[0], 1), 4), dtype=np.float32)
    n = lib().orc_voxel_centroid(_p(a), _sz(a.shape[0]), _f(leaf), _p(out), _sz(out.shape[0]))
    return out[:n].copy()


def voxel_centroid_box(pts, mn, mx, leaf):
    a = _pts(pts); out = np.empty_like(a)
    box = np.ascontiguousarray(np.concatenate([mn, mx]), dtype=np.float32)
    n = lib().orc_voxel_centroid_box(_p(a), _sz(a.shape[0]), _p(box), _f(leaf), _p(out), _sz(out.shape[0]))
    return out[:n].copy()


def voxel_keys_box(pts, mn, mx, leaf):
    a = _pts(pts); keys = np.empty(a.shape[0], np.uint64)
    box = np.ascontiguousarray(np.concatenate([mn, mx]), dtype=np.float32)
    depth = lib().orc_voxel_keys_box(_p(a), _sz(a.shape[0]), _p(box), _f(leaf), _p(keys))
    assert depth >= 0
    return keys, depth


def reproject(cmap, inv_poses, b2l, vfov, hfov, alpha=3.0, kf_begin=0, kf_end=None, threads=1):
    m = _pts(cmap); ip = _m(inv_poses); b = _m(b2l, 1)
    ke = ip.shape[0] if kf_end is None else kf_end
    r, c = rimg_size(vfov, hfov, alpha)
    cap = (ke - kf_begin) * r * c
    out = np.empty((max(cap, 1), 4), dtype=np.float32)
    off = np.zeros(ke - kf_begin + 1, dtype=np.uint64)
    n = lib().orc_reproject(_p(m), _sz(m.shape[0]), _p(ip), _p(b), _f(vfov), _f(hfov), _f(alpha), _sz(kf_begin), _sz(ke), _i(threads),
                            _p(out), _sz(cap), _p(off))
    return out[:n].copy(), off


def knn_labels(target, scans, offsets, poses, inv_poses, b2l, k, thr, kf_begin=0, kf_end=None, threads=1, use_kdtree=True):
    t = _pts(target); s = _pts(scans); off = np.ascontiguousarray(offsets, dtype=np.uint64)
    po = _m(poses); ip = _m(inv_poses); b = _m(b2l, 1)
    nkf = off.size - 1
    co = np.zeros(s.shape[0], dtype=np.uint8); loc = np.zeros_like(s)
    lib().orc_knn_labels(_p(t), _sz(t.shape[0]), _p(s), _p(off), _sz(nkf), _p(po), _p(ip), _p(b), _i(k), _f(thr), _sz(kf_begin),
                         _sz(nkf if kf_end is None else kf_end), _i(threads), _i(1 if use_kdtree else 0), _p(co), _p(loc))
    return co, loc


def knn_split(target, query, k, thr, use_kdtree=True):
    t = _pts(target); q = _pts(query)
    near = np.zeros(q.shape[0], dtype=np.uint8)
    lib().orc_knn_split(_p(t), _sz(t.shape[0]), _p(q), _sz(q.shape[0]), _i(k), _f(thr), _i(1 if use_kdtree else 0), _p(near))
    return near


def merge_to_global(scans, offsets, poses, l2b):
    s = _pts(scans); off = np.ascontiguousarray(offsets, dtype=np.uint64); po = _m(poses); l = _m(l2b, 1)
    out = np.empty_like(s)
    lib().orc_merge_to_global(_p(s), _p(off), _sz(off.size - 1), _p(po), _p(l), _p(out))
    return out


def preclean(pts, radius):
    a = _pts(pts); out = np.empty_like(a)
    n = lib().orc_preclean(_p(a), _sz(a.shape[0]), _f(radius), _p(out))
    return out[:n].copy()


def voxel_grid(pts, leaf, stable=False):
    """pcl::VoxelGrid as the session loader applies it to every scan (Session.cpp:284-289).  stable=False: PCL's in-voxel order
    (std::sort on the leaf index only; == oracle/_ref); stable=True: input order, which is what the DEVICE form of the cascade
    hand-over (ltm_voxel_grid_scanset) sums in -- the two differ in the last bit where a voxel holds three or more points."""
    a = _pts(pts); out = np.empty_like(a)
    lib().orc_set_voxel_grid_stable(_i(1 if stable else 0))
    try:
        n = lib().orc_voxel_grid(_p(a), _sz(a.shape[0]), _f(leaf), _p(out), _sz(out.shape[0]))
    finally:
        lib().orc_set_voxel_grid_stable(_i(0))
    return out[:n].copy()


class PipelineResult:
    def __init__(self, h):
        self.h = h

    def cloud(self, name):
        p = C.POINTER(C.c_float)(); n = _sz()
        if lib().orc_run_cloud(_vp(self.h), name.encode(), C.byref(p), C.byref(n)) != 0:
            return None
        if n.value == 0:
            return np.zeros((0, 4), dtype=np.float32)
        return np.ctypeslib.as_array(p, shape=(n.value, 4)).copy()

    def scanset(self, name):
        p = C.POINTER(C.c_float)(); o = C.POINTER(C.c_uint64)(); nk = _sz()
        if lib().orc_run_scanset(_vp(self.h), name.encode(), C.byref(p), C.byref(o), C.byref(nk)) != 0:
            return None
        off = np.ctypeslib.as_array(o, shape=(nk.value + 1,)).copy()
        n = int(off[-1])
        pts = np.ctypeslib.as_array(p, shape=(n, 4)).copy() if n else np.zeros((0, 4), dtype=np.float32)
        return pts, off

    def timings(self):
        names = (C.c_char_p * 64)(); secs = (C.c_double * 64)()
        n = lib().orc_run_timings(_vp(self.h), names, secs, 64)
        return {names[i].decode(): secs[i] for i in range(min(n, 64))}

    def free(self):
        if self.h:
            lib().orc_run_free(_vp(self.h)); self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def make_params(vfov=50.0, hfov=360.0, k=2, knn_thr=0.01, voxel=0.05, lidar2base=None, use_self_removert=False,
                res_list=(2.5,), repeat=1, threads=1, skip_hd_knn=False, kf_sample_stride=1):
    P = OrcParams()
    P.vfov, P.hfov, P.k, P.knn_thr, P.voxel = vfov, hfov, k, knn_thr, voxel
    l2b = np.eye(4).reshape(-1) if lidar2base is None else np.asarray(lidar2base, dtype=np.float64).reshape(-1)
    for i in range(16):
        P.lidar2base[i] = l2b[i]
    P.use_self_removert = 1 if use_self_removert else 0
    P.n_res = len(res_list)
    for i, r in enumerate(res_list):
        P.res_list[i] = r
    P.repeat, P.threads, P.skip_hd_knn, P.kf_sample_stride = repeat, threads, 1 if skip_hd_knn else 0, kf_sample_stride
    return P


def pipeline_run(params, central, query):
    """central/query: dicts with 'scans' (n,4) f32, 'offsets' u64, 'poses' (n,16), 'inv' (n,16)"""
    def unpack(s):
        return (_pts(s["scans"]), np.ascontiguousarray(s["offsets"], dtype=np.uint64), _m(s["poses"]), _m(s["inv"]))
    cs, co, cp, ci = unpack(central); qs, qo, qp, qi = unpack(query)
    h = lib().orc_pipeline_run(C.byref(params), _p(cs), _p(co), _sz(co.size - 1), _p(cp), _p(ci),
                               _p(qs), _p(qo), _sz(qo.size - 1), _p(qp), _p(qi))
    if not h:
        raise RuntimeError("orc_pipeline_run failed")
    return PipelineResult(h)
