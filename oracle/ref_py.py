"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libltm_ref.so -- the reference's own, unmodified sources
(/root/reference/ltremovert/src/{utility,RosParamServer,Session,Removerter}.cpp) compiled against the stand-in headers of
oracle/refshim/include, plus the flat-array entry points of oracle/refshim/ref_capi.cpp.

Only tests/, tools/ fixture generators and bench.py's cpu_baseline leg may import this module.  The library can only be BUILT where
/root/reference exists (the build container); the built .so travels to the GPU box with the snapshot like any other in-tree .so.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libltm_ref.so")
EXE_PATH = os.path.join(_HERE, "_ref", "removert_removert")
# the same process with the multi-resolution remove / revert loop switched on (oracle/refshim/removert_selfremovert_main.cpp)
EXE_SELFREMOVERT_PATH = os.path.join(_HERE, "_ref", "removert_selfremovert")
REFERENCE = "/root/reference/ltremovert"
_lib = None

_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float


class RefParams(C.Structure):
    _fields_ = [("vfov", _f), ("hfov", _f), ("k", _i), ("knn_thr", _f), ("voxel", _f), ("lidar2base", C.c_double * 16),
                ("use_self_removert", _i), ("n_res", _i), ("res_list", _f * 8), ("repeat", _i)]


def can_build():
    return os.path.isdir(os.path.join(REFERENCE, "src"))


def available():
    return os.path.exists(LIB_PATH) or can_build()


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "refshim")])


def lib():
    global _lib
    if _lib is None:
        if can_build():
            build()                      # make: a no-op when up to date
        L = C.CDLL(LIB_PATH)
        L.ref_rad2deg.restype = _f
        L.ref_rad2deg.argtypes = [_f]
        for fn in ("ref_linspace_int", "ref_splitPoseLine", "ref_parseProjectedPoints", "ref_octreeDownsampling", "ref_leaf_voxel_grid",
                   "ref_calcDescrepancy", "ref_getStaticIdxFromDynamicIdx", "ref_parsePointcloudSubsetUsingPtIdx", "ref_parseKeyframes",
                   "ref_precleaning", "ref_saved_names", "ref_vote_dynamic_idx"):
            getattr(L, fn).restype = _sz
        L.ref_rmv_create.restype = _vp
        _lib = L
    return _lib


def _pts(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)


def _m(a):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 16)


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def make_params(vfov=50.0, hfov=360.0, k=2, knn_thr=0.01, voxel=0.05, lidar2base=None, use_self_removert=False, res_list=(2.5,), repeat=1):
    p = RefParams()
    p.vfov, p.hfov, p.k, p.knn_thr, p.voxel = vfov, hfov, k, knn_thr, voxel
    l2b = np.eye(4) if lidar2base is None else np.asarray(lidar2base, dtype=np.float64)
    p.lidar2base = (C.c_double * 16)(*l2b.reshape(-1))
    p.use_self_removert = int(use_self_removert)
    p.n_res = len(res_list)
    for j, r in enumerate(res_list):
        p.res_list[j] = r
    p.repeat = repeat
    return p


# ---- free functions of utility.cpp ------------------------------------------------------------------------------------------
def rad2deg(r):
    r = np.ascontiguousarray(r, dtype=np.float32); out = np.empty_like(r)
    lib().ref_rad2deg_array(_p(r), _sz(r.size), _p(out))
    return out


def cart2sph(xyz):
    a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3); out = np.empty_like(a)
    lib().ref_cart2sph_array(_p(a), _sz(a.shape[0]), _p(out))
    return out


def rimg_size(vfov, hfov, alpha):
    r, c = _i(), _i()
    lib().ref_resetRimgSize(_f(vfov), _f(hfov), _f(alpha), C.byref(r), C.byref(c))
    return r.value, c.value


def linspace_int(a, b, n):
    out = np.empty(n, np.int32)
    m = lib().ref_linspace_int(_i(a), _i(b), _sz(n), _p(out))
    return out[:m]


def split_pose_line(line):
    out = np.empty(32, np.float64)
    n = lib().ref_splitPoseLine(line.encode(), _p(out), _sz(32))
    return out[:n].copy()


def inverse4x4(m):
    a = _m(m); out = np.empty_like(a)
    for j in range(a.shape[0]):
        lib().ref_inverse4x4(_p(a[j]), _p(out[j]))
    return out.reshape(np.asarray(m).shape)


def transform_global_map_to_local(pts, base_pose_inverse, base2lidar):
    a = _pts(pts); out = np.empty_like(a)
    lib().ref_transformGlobalMapToLocal(_p(a), _sz(a.shape[0]), _p(_m(base_pose_inverse)), _p(_m(base2lidar)), _p(out))
    return out


def local2global(pts, pose, lidar2base):
    a = _pts(pts); out = np.empty_like(a)
    lib().ref_local2global(_p(a), _sz(a.shape[0]), _p(_m(pose)), _p(_m(lidar2base)), _p(out))
    return out


def global2local(pts, pose_inverse, base2lidar):
    a = _pts(pts); out = np.empty_like(a)
    lib().ref_global2local(_p(a), _sz(a.shape[0]), _p(_m(pose_inverse)), _p(_m(base2lidar)), _p(out))
    return out


def merge_to_global(scans, offsets, poses, lidar2base):
    a = _pts(scans); off = np.ascontiguousarray(offsets, dtype=np.uint64); out = np.empty_like(a)
    lib().ref_mergeScansWithinGlobalCoordUtil(_p(a), _p(off), _sz(len(off) - 1), _p(_m(poses)), _p(_m(lidar2base)), _p(out))
    return out


def map2range_img(pts, vfov, hfov, rows, cols):
    a = _pts(pts)
    rimg = np.empty((rows, cols), np.float32); idx = np.empty((rows, cols), np.int32)
    lib().ref_map2RangeImg(_p(a), _sz(a.shape[0]), _f(vfov), _f(hfov), _i(rows), _i(cols), _p(rimg), _p(idx))
    return rimg, idx


def parse_projected_points(pts, vfov, hfov, rows, cols):
    a = _pts(pts); out = np.empty((rows * cols, 4), np.float32)
    n = lib().ref_parseProjectedPoints(_p(a), _sz(a.shape[0]), _f(vfov), _f(hfov), _i(rows), _i(cols), _p(out), _sz(rows * cols))
    return out[:n].copy()


def octree_downsampling(pts, leaf):
    a = _pts(pts); out = np.empty_like(a)
    n = lib().ref_octreeDownsampling(_p(a), _sz(a.shape[0]), _f(leaf), _p(out), _sz(a.shape[0]))
    return out[:n].copy()


def leaf_voxel_grid(pts, leaf):
    a = _pts(pts); out = np.empty_like(a)
    n = lib().ref_leaf_voxel_grid(_p(a), _sz(a.shape[0]), _f(leaf), _p(out), _sz(a.shape[0]))
    return out[:n].copy()


def leaf_knn(target, query, k):
    t, q = _pts(target), _pts(query)
    idx = np.empty((q.shape[0], k), np.int32); sqd = np.empty((q.shape[0], k), np.float32)
    lib().ref_leaf_knn(_p(t), _sz(t.shape[0]), _p(q), _sz(q.shape[0]), _i(k), _p(idx), _p(sqd))
    return idx, sqd


# ---- a Removerter instance ------------------------------------------------------------------------------------------------------
class Removerter:
    """ltremovert::Removerter of the reference.  Its constructor creates the output directories (Removerter.cpp:26-50): they go
    to a temporary directory; clouds the reference saves are captured in memory instead of written unless write_files=True."""

    def __init__(self, params=None, save_dir=None, write_files=False):
        self.params = params or make_params()
        self._tmp = None
        if save_dir is None:
            self._tmp = tempfile.TemporaryDirectory(prefix="ltm_ref_")
            save_dir = self._tmp.name
        self.h = lib().ref_rmv_create(C.byref(self.params), str(save_dir).encode(), _i(int(write_files)))

    def close(self):
        if self.h:
            lib().ref_rmv_destroy(_vp(self.h)); self.h = None
        if self._tmp:
            self._tmp.cleanup(); self._tmp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scan2range_img(self, pts, vfov, hfov, rows, cols):
        a = _pts(pts); rimg = np.empty((rows, cols), np.float32)
        lib().ref_scan2RangeImg(_vp(self.h), _p(a), _sz(a.shape[0]), _f(vfov), _f(hfov), _i(rows), _i(cols), _p(rimg))
        return rimg

    def calc_descrepancy(self, scan_rimg, diff_rimg, ptidx, thres=0.1):
        s = np.ascontiguousarray(scan_rimg, np.float32); d = np.ascontiguousarray(diff_rimg, np.float32); pi = np.ascontiguousarray(ptidx, np.int32)
        out = np.empty(s.size, np.int32)
        n = lib().ref_calcDescrepancy(_vp(self.h), _p(s), _p(d), _p(pi), _i(s.shape[0]), _i(s.shape[1]), _f(thres), _p(out), _sz(out.size))
        return out[:n].copy()

    def static_idx_from_dynamic_idx(self, dyn, num_all):
        d = np.ascontiguousarray(dyn, np.int32); out = np.empty(max(num_all + 1, 1), np.int32)
        n = lib().ref_getStaticIdxFromDynamicIdx(_vp(self.h), _p(d), _sz(d.size), _i(num_all), _p(out), _sz(out.size))
        return out[:n].copy()

    def subset_by_idx(self, pts, idx):
        a = _pts(pts); ix = np.ascontiguousarray(idx, np.int32); out = np.empty((ix.size, 4), np.float32)
        n = lib().ref_parsePointcloudSubsetUsingPtIdx(_vp(self.h), _p(a), _sz(a.shape[0]), _p(ix), _sz(ix.size), _p(out), _sz(ix.size))
        return out[:n].copy()

    def parse_keyframes(self, n_scans, start, end, gap=1):
        out = np.empty(max(n_scans, 1), np.int32)
        n = lib().ref_parseKeyframes(_vp(self.h), _i(n_scans), _i(start), _i(end), _i(gap), _p(out), _sz(out.size))
        return [int(v) for v in out[:n]]

    def precleaning(self, pts, radius):
        a = _pts(pts); out = np.empty_like(a)
        n = lib().ref_precleaning(_vp(self.h), _p(a), _sz(a.shape[0]), _f(radius), _p(out), _sz(a.shape[0]))
        return out[:n].copy()

    def vote_labels(self, cmap, scans, offsets, poses, alpha, which=0):
        """one vote pass (Removerter.cpp:542-593 / :485-540 ND / :429-482 PD) as a label array over the map, like the oracle's vote_labels"""
        m = _pts(cmap); sc = _pts(scans); off = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = np.empty(max(len(m), 1), np.int32)
        n = lib().ref_vote_dynamic_idx(_vp(self.h), _i(which), _p(m), _sz(len(m)), _p(sc), _p(off), _sz(len(off) - 1), _p(_m(poses)), _f(alpha), _p(out), _sz(out.size))
        lab = np.zeros(len(m), np.uint8)
        lab[out[:n]] = 1
        return lab

    def weak_strong_split(self, strong, weak):
        """Session::removeWeakNDMapPointsHavingStrongNDInNear (k = 2, thr = 1.0): 1 where a weak point moves to the strong map"""
        s, w = _pts(strong), _pts(weak); near = np.zeros(w.shape[0], np.uint8)
        lib().ref_weakStrongSplit(_vp(self.h), _p(s), _sz(s.shape[0]), _p(w), _sz(w.shape[0]), _p(near))
        return near

    # ---- Removerter::run() from makeGlobalMap() on, on in-memory sessions (dicts with scans / offsets / poses)
    def pipeline_run(self, central, query):
        def unpack(s):
            return _pts(s["scans"]), np.ascontiguousarray(s["offsets"], dtype=np.uint64), _m(s["poses"])
        cs, co, cp = unpack(central); qs, qo, qp = unpack(query)
        self.n_central = len(co) - 1
        rc = lib().ref_pipeline_run(_vp(self.h), C.byref(self.params), _p(cs), _p(co), _sz(len(co) - 1), _p(cp), _p(qs), _p(qo), _sz(len(qo) - 1), _p(qp))
        assert rc == 0
        return self

    def saved_names(self):
        n = lib().ref_saved_names(_vp(self.h), None, _sz(0))
        buf = C.create_string_buffer(n + 1)
        lib().ref_saved_names(_vp(self.h), buf, _sz(n + 1))
        return [s for s in buf.value.decode().split("\n") if s]

    def saved(self, rel):
        """a cloud the reference wrote with pcl::io::savePCDFileBinary, by its path relative to save_pcd_directory; None if never saved"""
        p = C.POINTER(C.c_float)(); n = _sz(); w = C.c_uint32(); h = C.c_uint32()
        if lib().ref_saved_cloud(_vp(self.h), rel.encode(), C.byref(p), C.byref(n), C.byref(w), C.byref(h)) != 0:
            return None
        if n.value == 0:
            return np.empty((0, 4), np.float32)
        return np.ctypeslib.as_array(p, shape=(n.value, 4)).copy()

    _STATE = {"central_map_static": (0, "static"), "central_map_dynamic": (0, "dynamic"), "query_map_static": (1, "static"), "query_map_dynamic": (1, "dynamic")}

    def cloud(self, name):
        """same names as the oracle's PipelineResult.cloud(): a map the reference saved, or (the four remove / revert results it only
        saves from selfRemovert) the session member itself"""
        if name in self._STATE:
            return self.session_map(*self._STATE[name])
        return self.saved(name + ".pcd")

    def scanset(self, name):
        """same names as the oracle's PipelineResult.scanset(): the per-keyframe files of one output directory, concatenated"""
        pts, off = [], [0]
        for k in range(self.n_central):
            c = self.saved(f"{name}/{k:06d}.pcd")
            assert c is not None, f"{name}/{k:06d}.pcd was not saved"
            pts.append(c); off.append(off[-1] + len(c))
        return (np.concatenate(pts) if pts else np.empty((0, 4), np.float32)), np.array(off, np.uint64)

    def session_scans(self, query, which, n_kf):
        pts, off = [], [0]
        buf = np.empty((1 << 22, 4), np.float32)
        for k in range(n_kf):
            n = _sz()
            rc = lib().ref_session_scans(_vp(self.h), _i(int(query)), which.encode(), _sz(k), _p(buf), _sz(buf.shape[0]), C.byref(n))
            assert rc == 0 and n.value <= buf.shape[0]
            pts.append(buf[:n.value].copy()); off.append(off[-1] + n.value)
        return np.concatenate(pts), np.array(off, np.uint64)

    def session_map(self, query, which, cap=1 << 24):
        buf = np.empty((cap, 4), np.float32); n = _sz()
        rc = lib().ref_session_map(_vp(self.h), _i(int(query)), which.encode(), _p(buf), _sz(cap), C.byref(n))
        assert rc == 0 and n.value <= cap
        return buf[:n.value].copy()


def run_process(yaml_path, timeout=3600, self_removert=False):
    """the reference's process (removert_main.cpp + everything) on a params_ltmapper.yaml-style file: files in, files out.
    self_removert: the second main, whose removeHighDynamicPoints runs the reference's own selfRemovert (Removerter.cpp:1582,1586 un-commented)"""
    env = dict(os.environ, REFSHIM_PARAMS=str(yaml_path))
    return subprocess.run([EXE_SELFREMOVERT_PATH if self_removert else EXE_PATH], env=env, capture_output=True, text=True, timeout=timeout)
