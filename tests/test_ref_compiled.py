"""The oracle against REFERENCE-COMPILED code (oracle/_ref, round 4).

oracle/refshim/Makefile compiles the reference's own, unmodified sources -- /root/reference/ltremovert/src/{utility,RosParamServer,
Session,Removerter,removert_main}.cpp with include/removert/*.h -- against stand-in headers for ROS / Eigen / OpenCV / PCL
(oracle/refshim/include; none of those libraries is in the image), serial (no OpenMP: the reference's parallel min-update is a
documented race, utility.cpp:127-138).  Everything the reference ITSELF computes is therefore pinned to the reference's text:
cart2sph / rad2deg / resetRimgSize / map2RangeImg / scan2RangeImg / parseProjectedPoints / calcDescrepancy... / the std::set union
and complement / linspace / removeOnce, revertOnce, selfRemovert, filterStrongND/PD / the k-NN label rule / weak->strong ND /
detectLowDynamicPoints / updateCurrentMap / updateScansScanwise / parseKeyframes (quirk Q6) / precleaningKeyframes / the whole run().
What stays "parity unpinned" are the library leaves the stand-ins restate (pcl::transformPointCloud, OctreePointCloudVoxelCentroid,
VoxelGrid, KdTreeFLANN, ExtractIndices, PCD I/O, Eigen's Matrix4d::inverse()): there the oracle is compared with a SECOND, literal
derivation (pointer octree, PCL's index sort, a plain kd-tree) -- which is how round 4 found that pcl::VoxelGrid's in-voxel order is
std::sort's, not input order.

These tests need /root/reference (to build) or a previously built oracle/_ref/libltm_ref.so; otherwise they are skipped."""
import os

import numpy as np
import pytest

from conftest import assert_clouds_equal

from oracle import ref_py

pytestmark = pytest.mark.skipif(not ref_py.available(), reason="needs /root/reference (build container) or a prebuilt oracle/_ref")

VFOV, HFOV = 50.0, 360.0
RES = [2.5, 2.375, 2.0, 1.9, 1.5, 1.425, 3.0, 1.0, 0.7]


@pytest.fixture(scope="module")
def ref():
    ref_py.lib()
    return ref_py


@pytest.fixture(scope="module")
def rmv(ref):
    r = ref.Removerter()
    yield r
    r.close()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _se3(rng, max_t=200.0, max_tilt_deg=8.0, origin=(0.0, 0.0, 0.0)):
    """a keyframe pose as LT-SLAM writes them: any yaw, a few degrees of roll / pitch, z drift, optionally a far session origin"""
    yaw = rng.uniform(-np.pi, np.pi); roll, pitch = np.deg2rad(rng.normal(0, max_tilt_deg / 2.5, 2))
    cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = np.asarray(origin) + rng.uniform(-max_t, max_t, 3) * [1, 1, 0.02]
    return T


def _adversarial_xyz(rng, n):
    a = (rng.normal(0, 30, (n, 3)) * [1, 1, 0.15]).astype(np.float32)
    special = np.array([[1, 0, 0], [-1, 0.0, 0], [-1, -0.0, 0], [0, 0, 1], [0, 0, -1], [0, 0, 0], [0.0, -0.0, 0], [-0.0, 0.0, 0], [1e-30, 1e-38, 0],
                        [-3, 1e-7, 0.2], [-3, -1e-7, 0.2], [5, 5, 5], [1e4, 1, 1], [0, 1, 0], [0, -1, 0], [1e-20, -1e-20, 1e-20], [7, 0, 3.2643],
                        [np.float32(2.0) ** -126, 0, 0], [3e38, 3e38, 0], [1, 1, 1e-45]], np.float32)
    m = min(n, len(special))
    a[:m] = special[:m]
    return a


# ------------------------------------------------------------------------------------------------- scalar numerics (a3)
def test_cart2sph_rad2deg_bitwise(ref, orc):
    rng = np.random.default_rng(1)
    xyz = _adversarial_xyz(rng, 400_000)
    got, want = ref.cart2sph(xyz), orc.cart2sph(xyz)
    assert (_bits(got) == _bits(want)).all(), "cart2sph (utility.cpp:38-51: std::atan2 / std::sqrt float overloads of the host libm)"
    r = np.concatenate([got[:, 0], got[:, 1], rng.uniform(-4, 4, 100_000).astype(np.float32), np.array([0.0, -0.0, np.pi, -np.pi, 1e-40], np.float32)])
    want_deg = np.array([orc.lib().orc_rad2deg(float(v)) for v in r[:20000]], np.float32)
    assert (_bits(ref.rad2deg(r[:20000])) == _bits(want_deg)).all(), "rad2deg (utility.cpp:53-56)"


def test_reset_rimg_size(ref, orc):
    for fov in ((50.0, 360.0), (45.0, 360.0), (26.9, 360.0), (33.3, 180.0), (90.0, 270.0)):
        for a in RES:
            assert ref.rimg_size(fov[0], fov[1], a) == orc.rimg_size(fov[0], fov[1], a), (fov, a)
    assert ref.rimg_size(50.0, 360.0, 2.5) == (125, 900) and ref.rimg_size(50.0, 360.0, 0.95 * 2.5) == (119, 855)


def test_linspace_and_complement_quirks(ref, rmv):
    assert list(ref.linspace_int(0, 5, 5)) == [0, 1, 2, 3, 4]                   # h = 5/4 = 1 in integer arithmetic
    assert list(ref.linspace_int(0, 2, 2)) == [0, 2]                             # quirk Q5: N = 2 gives {0, 2}
    assert list(ref.linspace_int(0, 1000, 1000)) == list(range(1000))
    # getStaticIdxFromDynamicIdx (Removerter.cpp:675-687): complement of the dynamic set in linspace(0, M, M) -- ascending, unique
    got = rmv.static_idx_from_dynamic_idx([5, 3, 3, 9, 0], 12)
    assert list(got) == [1, 2, 4, 6, 7, 8, 10, 11]
    assert list(rmv.static_idx_from_dynamic_idx([], 7)) == list(range(7))


def test_split_pose_line_and_inverse(ref, orc):
    rng = np.random.default_rng(2)
    line = "0.999876 -0.0157073 0 12.5 0.0157073 0.999876 0 -3.25 0 0 1 1.9"
    v = ref.split_pose_line(line)
    assert len(v) == 12 and v[3] == 12.5 and v[7] == -3.25
    mats = np.array([_se3(rng, origin=(3e4, -2e4, 50.0)) for _ in range(300)] + [np.eye(4) + rng.normal(0, 0.3, (4, 4)) for _ in range(50)])
    got = ref.inverse4x4(mats.reshape(-1, 16))
    want = np.array([orc.inverse4x4(m) for m in mats]).reshape(-1, 16)
    assert (got.view(np.uint64) == want.view(np.uint64)).all(), "Matrix4d::inverse(): the stand-in and the oracle restate the same Eigen SSE2 order"
    assert np.abs(got.reshape(-1, 4, 4) @ mats - np.eye(4)).max() < 1e-6


# ------------------------------------------------------------------------------------------------- transforms (a2, a14)
def test_transforms_bitwise_full_se3_far_origin(ref, orc):
    rng = np.random.default_rng(3)
    pts = np.concatenate([_adversarial_xyz(rng, 50_000), np.zeros((50_000, 1), np.float32)], axis=1)
    pts[:, 3] = rng.uniform(0, 255, len(pts))
    ext = _se3(rng, max_t=1.5, max_tilt_deg=20.0)
    for origin in ((0, 0, 0), (4.1e4, -2.7e4, 120.0)):
        T = _se3(rng, origin=origin)
        Tinv, b2l = orc.inverse4x4(T), orc.inverse4x4(ext)
        g = pts.copy(); g[:, :3] += np.asarray(origin, np.float32)
        for B in (np.eye(4), b2l):
            got = ref.transform_global_map_to_local(g, Tinv, B)
            want = orc.transform(B, orc.transform(Tinv, g))
            assert_clouds_equal(got, want, "transformGlobalMapToLocal (utility.cpp:64-72)")
        assert_clouds_equal(ref.local2global(pts, T, ext), orc.transform(T, orc.transform(ext, pts)), "local2global (utility.cpp:160-168)")
        assert_clouds_equal(ref.global2local(g, Tinv, b2l), orc.transform(b2l, orc.transform(Tinv, g)), "global2local (utility.cpp:194-202)")
    off = np.array([0, 1000, 1000, 30_000, len(pts)], np.uint64)
    poses = np.array([_se3(rng) for _ in range(4)]).reshape(-1, 16)
    assert_clouds_equal(ref.merge_to_global(pts, off, poses, ext), orc.merge_to_global(pts, off, poses, ext), "mergeScansWithinGlobalCoordUtil (utility.cpp:170-192)")


# ------------------------------------------------------------------------------------------------- range images (a4, a5, a11)
def _cloud(rng, n, spread=30.0):
    p = np.empty((n, 4), np.float32)
    p[:, :3] = _adversarial_xyz(rng, n) * (spread / 30.0)
    p[:, 3] = rng.uniform(0, 255, n)
    return p


@pytest.mark.parametrize("alpha", [2.5, 2.375, 1.5, 3.0])
def test_map2rangeimg_and_scan2rangeimg_bitwise(ref, rmv, orc, alpha):
    rng = np.random.default_rng(int(alpha * 1000))
    R, C = orc.rimg_size(VFOV, HFOV, alpha)
    pts = _cloud(rng, 120_000)
    pts[1000:1200] = pts[2000:2200]                               # exact ties: the lowest index must win (utility.cpp:134-138, serial)
    pts[5000:5100, :3] *= np.float32(0.0)                         # many points in one pixel with range 0
    rimg, idx = ref.map2range_img(pts, VFOV, HFOV, R, C)
    w_rimg, w_idx = orc.range_image(pts, VFOV, HFOV, R, C)
    assert (_bits(rimg) == _bits(w_rimg)).all() and (idx == w_idx).all(), "map2RangeImg (utility.cpp:92-142)"
    s_rimg = rmv.scan2range_img(pts[:40_000], VFOV, HFOV, R, C)
    assert (_bits(s_rimg) == _bits(orc.range_image(pts[:40_000], VFOV, HFOV, R, C, want_idx=False)[0])).all(), "scan2RangeImg (Removerter.cpp:109-156)"
    # other fields of view (generic elevation / division paths on the device side rest on the same formula)
    for vf, hf in ((33.3, 360.0), (90.0, 180.0)):
        R2, C2 = orc.rimg_size(vf, hf, alpha)
        a, b = ref.map2range_img(pts[:30_000], vf, hf, R2, C2)
        c, d = orc.range_image(pts[:30_000], vf, hf, R2, C2)
        assert (_bits(a) == _bits(c)).all() and (b == d).all()


def test_parse_projected_points_drops_point_zero(ref, orc):
    rng = np.random.default_rng(5)
    pts = _cloud(rng, 60_000)
    pts[np.abs(pts[:, :3]).max(axis=1) > 1e30] = 1.0     # 3e38 overflows to inf under a pose and 0*inf = NaN in the next transform: int(NaN) indexes
    R, C = orc.rimg_size(VFOV, HFOV, 3.0)                # the image out of bounds in the reference (UB; non-finite points are unsupported)
    # Session::parseScansViaProjection (Session.cpp:348-360) always transforms first; the identity transform is NOT a no-op: it turns -0.0
    # into +0.0 ((float)(1*x + 0*y + 0*z + 0)), which moves a point on the -180 deg seam from column 0 to the last column
    T = _se3(rng)
    Tinv = orc.inverse4x4(T)
    for Ti in (np.eye(4), Tinv):
        got = ref.parse_projected_points(ref.transform_global_map_to_local(pts, Ti, np.eye(4)), VFOV, HFOV, R, C)
        want, off = orc.reproject(pts, Ti.reshape(1, 16), np.eye(4), VFOV, HFOV, 3.0)
        assert_clouds_equal(got, want, "parseProjectedPoints (utility.cpp:74-89)")
    got = ref.parse_projected_points(ref.transform_global_map_to_local(pts, np.eye(4), np.eye(4)), VFOV, HFOV, R, C)
    want, off = orc.reproject(pts, np.eye(4).reshape(1, 16), np.eye(4), VFOV, HFOV, 3.0)
    assert_clouds_equal(got, want, "parseProjectedPoints (utility.cpp:74-89)")
    two = np.array([[10, 0, 0, 1], [0, 10, 0, 2]], np.float32)    # quirk Q3: ptidx 0 doubles as "empty": map point 0 is never emitted
    assert_clouds_equal(ref.parse_projected_points(two, VFOV, HFOV, R, C), two[1:], "Q3")


def test_calc_descrepancy_rule(rmv):
    # Removerter.cpp:381-413: emit ptidx iff diff < 200 && diff > thres, row-major
    scan = np.full((3, 4), 10000.0, np.float32); diff = np.zeros((3, 4), np.float32); idx = np.arange(12, dtype=np.int32).reshape(3, 4) + 100
    diff[0, 1], diff[1, 2], diff[2, 3], diff[2, 0], diff[1, 0] = 0.1000001, 0.1, 199.99, 200.0, np.float32(0.1) + np.float32(1e-8)
    # 0.1f + 1e-8 rounds to the float after 0.1f: flagged; 0.1f itself is not (strict >); 200.0 is not (strict <)
    assert list(rmv.calc_descrepancy(scan, diff, idx, 0.1)) == [101, 104, 111]
    diff[0, 0] = np.nan
    assert list(rmv.calc_descrepancy(scan, diff, idx, 0.1)) == [101, 104, 111]


def _plain_cloud(rng, n, spread):
    p = rng.normal(0, spread, (n, 4)).astype(np.float32)
    p[:, 2] *= np.float32(0.15)
    p[:, 3] = rng.uniform(0, 255, n)
    return p


# ------------------------------------------------------------------------------------------------- PCL leaves: a second derivation
def test_octree_downsampling_second_derivation(ref, orc):
    rng = np.random.default_rng(6)
    for n, spread, leaf in ((50_000, 30.0, 0.05), (50_000, 30.0, 0.4), (20_000, 3.0, 0.05), (2, 1.0, 0.05), (1, 1.0, 0.05), (3000, 0.2, 0.05)):
        pts = _cloud(rng, n, spread)
        if n > 100:
            pts[100:200] = pts[300:400]                                      # duplicates
            pts[400:500, :3] = np.round(pts[400:500, :3] / leaf) * leaf      # lattice-aligned coordinates
        a, b = ref.octree_downsampling(pts, leaf), orc.voxel_centroid(pts, leaf)
        assert_clouds_equal(a, b, f"octreeDownsampling n={n} leaf={leaf}: literal pointer octree vs the oracle's sort-based form")
    m1 = orc.voxel_centroid(_cloud(rng, 80_000), 0.05)                       # a re-voxelised map (what every pass does)
    assert_clouds_equal(ref.octree_downsampling(m1[::2], 0.05), orc.voxel_centroid(m1[::2], 0.05), "re-voxelised")


def test_voxel_grid_in_voxel_order_is_std_sorts(ref, orc):
    """pcl::VoxelGrid sorts (leaf, point) pairs with std::sort on the leaf index only: the float sums of a voxel with >= 3 points depend on
    the order that sort leaves.  The oracle (and the C++ loader of the product) make the same call; input order is measurably different."""
    rng = np.random.default_rng(7)
    pts = _plain_cloud(rng, 60_000, 0.6)           # ~ 4 x 4 x 0.6 m: no int32 overflow at 0.05 m, many points per voxel
    got = ref.leaf_voxel_grid(pts, 0.05)
    assert 0 < len(got) < len(pts)
    assert_clouds_equal(got, orc.voxel_grid(pts, 0.05), "pcl::VoxelGrid, PCL order")
    stable = orc.voxel_grid(pts, 0.05, stable=True)
    assert stable.shape == got.shape
    n_diff = int((_bits(stable) != _bits(got)).any(axis=1).sum())
    assert n_diff > 0, "input order and std::sort order should differ somewhere on 60 k clustered points"
    assert np.abs(stable - got)[:, :3].max() < 1e-5
    big = _plain_cloud(rng, 20_000, 300.0)         # the overflow early-out: output = input
    assert_clouds_equal(ref.leaf_voxel_grid(big, 0.05), big, "early-out")
    assert_clouds_equal(orc.voxel_grid(big, 0.05), big, "early-out (oracle)")


def test_knn_rule_reference_text_over_second_kdtree(ref, rmv, orc):
    rng = np.random.default_rng(8)
    strong = _cloud(rng, 30_000, spread=10.0)
    weak = _cloud(rng, 20_000, spread=10.0)
    weak[:5000, :3] = strong[:5000, :3] + rng.normal(0, 0.3, (5000, 3)).astype(np.float32)
    near = rmv.weak_strong_split(strong, weak)                       # Session.cpp:452-484, k = 2, thr = 1.0, the reference's own arithmetic
    want = orc.knn_split(strong, weak, 2, 1.0)
    assert (near == want).all() and 0 < near.sum() < len(near)
    # strict < at the threshold: two neighbours at squared distance 1.0 each -> mean 1.0 -> NOT near; slightly closer -> near
    s2 = np.array([[0, 0, 0, 0], [2, 0, 0, 0]], np.float32)
    w2 = np.array([[1, 0, 0, 0], [1, 0.0, np.float32(1e-3), 0], [0.9995, 0, 0, 0]], np.float32)
    assert list(rmv.weak_strong_split(s2, w2)) == list(orc.knn_split(s2, w2, 2, 1.0)) == [0, 0, 0]
    s3 = np.array([[0, 0, 0, 0], [1.9, 0, 0, 0]], np.float32)
    assert list(rmv.weak_strong_split(s3, w2[:1])) == list(orc.knn_split(s3, w2[:1], 2, 1.0)) == [1]
    # the stand-in kd-tree itself against brute force (squared L2_Simple distances, ascending)
    q = weak[:300]
    idx, sqd = ref.leaf_knn(strong, q, 3)
    with np.errstate(over="ignore"):          # one adversarial target point is at 3e38: its squared distance is +inf on both sides
        d = ((q[:, None, :3] - strong[None, :, :3]) ** 2)
        brute = np.sort(((d[:, :, 0] + d[:, :, 1]) + d[:, :, 2]), axis=1)[:, :3]
    assert (_bits(sqd) == _bits(brute)).all()


# ------------------------------------------------------------------------------------------------- session-level host logic
def test_parse_keyframes_quirk_q6(rmv):
    import fileproto
    for n, start, end in ((40, 11, 39), (40, 10, 39), (50, 1, 20), (10, 0, 9), (30, 5, 100), (7, 3, 3)):
        assert rmv.parse_keyframes(n, start, end) == fileproto.parse_keyframes(n, start, end), (n, start, end)
    assert rmv.parse_keyframes(40, 11, 39)[0] == 12, "an odd start_idx skips the first in-range scan (Session.cpp:149-152)"
    assert rmv.parse_keyframes(20, 0, 19, gap=3) == [0, 3, 6, 9, 12, 15, 18]


def test_precleaning(rmv, orc):
    rng = np.random.default_rng(9)
    pts = _cloud(rng, 50_000, spread=3.0)
    assert_clouds_equal(rmv.precleaning(pts, 2.5), orc.preclean(pts, 2.5), "precleaningKeyframes (Session.cpp:506-533)")


def test_vote_passes_reference_text_all_three_forms(ref, rmv, orc):
    """calcDescrepancyAndParseDynamicPointIdxForEachScan / ...ForND / ...ForPD (Removerter.cpp:542-593, 485-540, 429-482) over a session map:
    scan image, map transform, map image, signed difference, threshold window, std::set union -- the reference's text against the oracle's labels"""
    from tools import synth
    S = synth.to_numpy(synth.make_session(1, 5, "small", tilt_deg=2.0, z_drift=0.1, origin=(2.1e4, 1.3e4, 40.0)))
    S["inv"] = orc.inverse_poses(S["poses"])
    cmap = orc.voxel_centroid(orc.merge_to_global(S["scans"], S["offsets"], S["poses"], np.eye(4)), 0.05)
    for alpha in (2.5, 2.375, 1.5):
        for which, mode in ((0, 0), (1, 1), (2, 0)):
            got = rmv.vote_labels(cmap, S["scans"], S["offsets"], S["poses"], alpha, which)
            want = orc.vote_labels(cmap, S["scans"], S["offsets"], S["inv"], np.eye(4), VFOV, HFOV, alpha, 0.1, mode)
            assert (got == want).all(), f"res {alpha} form {which}: {(got != want).sum()} labels differ"
    assert 0 < want.sum() < len(want)


# ------------------------------------------------------------------------------------------------- the pipeline
def _tilted(S, rng, origin):
    """the same scans under full SE(3) poses: a few degrees of roll / pitch, z drift, a far session origin"""
    S = dict(S)
    poses = S["poses"].reshape(-1, 4, 4).copy()
    far = np.eye(4); far[:3, 3] = origin
    out = []
    for P in poses:
        tilt = _se3(rng, max_t=0.0, max_tilt_deg=3.0); tilt[:2, :2] = np.eye(2) if False else tilt[:2, :2]
        yawless = np.eye(4); yawless[:3, :3] = tilt[:3, :3]
        # keep the trajectory, add roll / pitch about the sensor and a z drift, then move the whole session far from the origin
        Q = P @ yawless
        Q[2, 3] += rng.normal(0, 0.05)
        out.append(far @ Q)
    S["poses"] = np.array(out).reshape(-1, 16)
    return S


CASES = [
    dict(name="1res", n_kf=4, three=False, k=2, thr=0.01),
    dict(name="3res", n_kf=3, three=True, k=3, thr=0.1),
    dict(name="1res_extrinsic", n_kf=3, three=False, k=2, thr=0.01, extrinsic=True),
    dict(name="3res_se3_far_origin", n_kf=3, three=True, k=2, thr=0.02, se3=(3.2e4, -4.4e4, 80.0)),
    dict(name="1res_k1_voxel01", n_kf=3, three=False, k=1, thr=0.05, voxel=0.1),
]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_pipeline_equals_reference_compiled_run(ref, orc, case):
    """Removerter::run() from makeGlobalMap() on (Removerter.cpp:1653-1678): every cloud the reference saves (16 maps, 5 scan directories,
    the selfRemovert maps) and the session state it keeps, against the oracle's run -- bitwise"""
    from tools import synth
    rng = np.random.default_rng(11)
    C = synth.to_numpy(synth.make_session(1, case["n_kf"], "tiny")); Q = synth.to_numpy(synth.make_session(2, case["n_kf"], "tiny"))
    if case.get("se3"):
        C, Q = _tilted(C, rng, case["se3"]), _tilted(Q, rng, case["se3"])
    for S in (C, Q):
        S["inv"] = orc.inverse_poses(S["poses"])
    ext = _se3(np.random.default_rng(12), max_t=1.0, max_tilt_deg=15.0) if case.get("extrinsic") else np.eye(4)
    res = (2.5, 2.0, 1.5) if case["three"] else (2.5,)
    voxel = case.get("voxel", 0.05)
    kw = dict(k=case["k"], knn_thr=case["thr"], voxel=voxel, lidar2base=ext, use_self_removert=case["three"], res_list=res)
    R = ref.Removerter(ref.make_params(**kw)).pipeline_run(C, Q)
    O = orc.pipeline_run(orc.make_params(**kw), C, Q)
    n_checked = 0
    for m in ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_sess_high_dyn", "query_sess_high_dyn", "union_map_queryside",
              "union_map_centralside", "pd_map", "nd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map", "updated_map", "updated_map_strong"]:
        a, b = R.cloud(m), O.cloud(m)
        assert (a is None) == (b is None), f"{m}: saved by one side only"
        if a is not None:
            assert_clouds_equal(a, b, m); n_checked += len(a)
    for s in ["scans_updated", "scans_updated_strong", "scans_pd", "scans_pd_strong", "scans_nd_strong"]:
        (a, ao), (b, bo) = R.scanset(s), O.scanset(s)
        assert (ao == bo).all(), s
        assert_clouds_equal(a, b, s); n_checked += len(a)
    # state the reference keeps but does not save
    for q, tag in ((0, "central"), (1, "query")):
        assert_clouds_equal(R.session_map(q, "static"), O.cloud(f"{tag}_map_static"), f"{tag} map_global_curr_static_")
        assert_clouds_equal(R.session_map(q, "dynamic"), O.cloud(f"{tag}_map_dynamic"), f"{tag} map_global_curr_dynamic_")
        for which, name in (("static_projected", f"{tag}_static_projected"), ("knn_coexist", f"{tag}_knn_coexist"), ("knn_diff", f"{tag}_knn_diff")):
            (a, ao), (b, bo) = R.session_scans(q, which, case["n_kf"]), O.scanset(name)
            assert (ao == bo).all(), name
            assert_clouds_equal(a, b, name)
    if case["three"]:      # selfRemovert's own saves (Removerter.cpp:318-338, quirk Q9: the doubled "ResX")
        names = R.saved_names()
        assert "map_static/CentralStaticMapMapsideGlobalResX_MVMResX1.500000.pcd" in names and "map_dynamic/QueryDynamicMapMapsideGlobal_MVMResX1.500000.pcd" in names
        assert_clouds_equal(R.saved("map_static/CentralStaticMapMapsideGlobalResX_MVMResX1.500000.pcd"), O.cloud("central_map_static"), "selfRemovert static save")
    assert n_checked > 10_000
    R.close()


def test_reference_process_files_to_files(ref, orc, tmp_path):
    """the reference's PROCESS -- main(), RosParamServer, loadSessionInfo, parseKeyframes / parseKeyframesInROI, loadKeyframes (PCD reader +
    pcl::VoxelGrid), precleaningKeyframes, run(), the PCD writer -- on session directories and a params_ltmapper.yaml, against the oracle on
    the same loaded data: all 19 outputs bitwise, headers as pcl::io::savePCDFileBinary writes them"""
    import fileproto as fp
    from tools import synth
    n_kf = 40
    sess = [synth.to_numpy(synth.make_session(s, n_kf, "tiny")) for s in (1, 2)]
    dirs = fp.write_session_dirs(tmp_path, sess, ascii_scans=(13,))
    out = tmp_path / "out"
    start, end = 11, 39
    yaml = tmp_path / "params.yaml"
    yaml.write_text(fp.yaml_text(tmp_path, dirs, out, start, end))
    r = ref.run_process(yaml)
    assert r.returncode == 0, r.stderr[-2000:]
    c_kf = fp.parse_keyframes(n_kf, start, end)
    q_kf = fp.query_keyframes_in_roi(sess[0], c_kf, sess[1], n_kf)
    assert c_kf[0] == 12 and len(q_kf) > 3
    C = fp.host_load(orc, sess[0], c_kf, roundtrip_ascii=(13,)); Q = fp.host_load(orc, sess[1], q_kf, roundtrip_ascii=(13,))
    O = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01), C, Q)
    fp.compare_output_tree(out, O, [sess[0]["names"][k] for k in c_kf], assert_clouds_equal)
    assert sorted(os.listdir(out)) == sorted([m + ".pcd" for m in fp.MAP_FILES if O.cloud(m) is not None] + [d for d, _ in fp.SCAN_DIRS] + ["map_static", "map_dynamic"])


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_fuzz_tool_cases_against_reference_compiled(ref, seed):
    """tools/fuzz_ref_vs_oracle.py (240 seeded cases, 333 M points: profiles/r4_fuzz_ref_vs_oracle_240.json) stays runnable: three of its draws"""
    from tools import fuzz_ref_vs_oracle as fz
    r = fz.run_case(fz.draw(seed))
    assert r["ok"], r["differences"]
    assert r["outputs_compared"] >= 25 and r["points_compared"] > 10_000
