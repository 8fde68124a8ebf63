"""Known-answer tests that pin the CPU oracle.  Each expected value is derivable by hand from the
cited reference lines; the reference itself ships no tests or golden vectors."""
import os
import subprocess

import numpy as np
import pytest

VFOV, HFOV = 50.0, 360.0
I4 = np.eye(4)


def test_rimg_sizes(orc):   # utility.cpp:222-236, float32 round of fov*alpha
    want = {2.5: (125, 900), 2.375: (119, 855), 2.0: (100, 720), 1.9: (95, 684), 1.5: (75, 540), 1.425: (71, 513), 3.0: (150, 1080)}
    for a, rc in want.items():
        assert orc.rimg_size(VFOV, HFOV, float(np.float32(a))) == rc
    assert orc.rimg_size(VFOV, HFOV, float(np.float32(0.95 * 2.5))) == (119, 855)


def test_pixel_kats(orc):   # utility.cpp:122-123 : roundf half away, clamp not drop
    pts = [[1, 0, 0], [-1, 0.0, 0], [-1, -0.0, 0], [0, 0, 1], [0, 0, -1], [0, 1, 0], [0, -1, 0]]
    rc, rng = orc.pixel(pts, VFOV, HFOV, 125, 900)
    assert rc.tolist() == [[63, 450], [63, 899], [63, 0], [0, 450], [124, 450], [63, 675], [63, 225]]
    assert (rng == 1.0).all()
    # (0,0,0): az = el = 0, r = 0 -> centre pixel, and it wins its pixel
    rc0, r0 = orc.pixel([[0, 0, 0]], VFOV, HFOV, 125, 900)
    assert rc0.tolist() == [[63, 450]] and r0[0] == 0.0


def test_rad2deg_is_double_division_then_float(orc):   # utility.cpp:53-56
    for r in (0.5, 1.0, -2.2, 3.1415927, 1e-8, 0.0, -0.0):
        want = np.float32(np.float64(np.float32(r)) * 180.0 / np.pi)
        got = np.float32(orc.lib().orc_rad2deg(np.float32(r)))
        assert got.view(np.uint32) == want.view(np.uint32)


def test_range_image_tie_and_strict_less(orc):   # utility.cpp:134-138
    p = np.array([[10, 0, 0, 1], [10, 0, 0, 2], [9.5, 0, 0, 3], [9.5, 0, 0, 4]], np.float32)
    r, i = orc.range_image(p, VFOV, HFOV, 125, 900)
    assert r[63, 450] == np.float32(9.5) and i[63, 450] == 2          # lowest index among the minimal range
    assert (r == 10000.0).sum() == 125 * 900 - 1 and (i != 0).sum() == 1


def test_transform_is_double_left_to_right_float_store(orc):   # PCL Transformer<double>::se3
    T = np.array([[0.1, 0.7, -0.2, 1e6], [0.3, -0.9, 0.5, -1e6], [1 / 3, 1 / 7, 1 / 9, 0.125], [0, 0, 0, 1]])
    p = np.array([[1.1, 2.2, 3.3, 9]], np.float32)
    x, y, z = (np.float64(v) for v in p[0, :3])
    want = [np.float32(((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]) for r in range(3)]
    got = orc.transform(T, p)[0]
    assert [np.float32(g).view(np.uint32) for g in got[:3]] == [w.view(np.uint32) for w in want] and got[3] == 9


def test_reprojection_drops_point_zero(orc):   # utility.cpp:82,104 (quirk Q3)
    two = np.array([[5, 0, 0, 1], [0, 5, 0, 2]], np.float32)
    pts, off = orc.reproject(two, I4.reshape(1, 16), I4, VFOV, HFOV, 3.0)
    assert off.tolist() == [0, 1] and pts[0, 3] == 2.0


def test_knn_threshold_is_mean_squared_and_strict(orc):   # Session.cpp:592-599 (quirk Q4)
    q = np.array([[0, 0, 0, 0]], np.float32)
    t = np.array([[0.1, 0, 0, 0], [0, 0.1, 0, 0], [3, 3, 3, 0]], np.float32)
    d2 = np.float32(0.1) * np.float32(0.1)
    avg = np.float32(np.float32(np.float64(d2) + np.float64(d2)) / np.float32(2))
    assert orc.knn_split(t, q, 2, float(avg))[0] == 0                               # avg == thr -> diff (strict <)
    assert orc.knn_split(t, q, 2, float(np.nextafter(avg, np.float32(1))))[0] == 1
    t2 = np.array([[0.0999, 0, 0, 0], [0, 0.0999, 0, 0]], np.float32)
    assert orc.knn_split(t2, q, 2, 0.01)[0] == 1
    # k larger than the target: PCL clamps k, the divisor stays k
    assert orc.knn_split(t[:1], q, 2, 0.0051)[0] == 1 and orc.knn_split(t[:1], q, 2, 0.005)[0] == 0


def test_kdtree_equals_brute_force(orc):
    rng = np.random.default_rng(4)
    tgt = rng.normal(0, 3, (5000, 4)).astype(np.float32)
    qry = rng.normal(0, 3, (3000, 4)).astype(np.float32)
    tgt[:200] = tgt[200:400]                     # duplicates / ties
    for k, thr in ((1, 0.02), (2, 0.05), (3, 0.1), (5, 0.5)):
        assert (orc.knn_split(tgt, qry, k, thr, True) == orc.knn_split(tgt, qry, k, thr, False)).all()


def test_voxel_centroid_lattice_and_order(orc):   # utility.cpp:204-219 + PCL octree semantics
    # two points per voxel, float sums in input order, output in Morton order with x most significant
    p = np.array([[0.01, 0.01, 0.01, 1], [10.0, 0.0, 0.0, 2], [0.02, 0.02, 0.02, 3], [0.0, 10.0, 0.0, 4], [0.0, 0.0, 10.0, 5]], np.float32)
    out = orc.voxel_centroid(p, 1.0)
    assert out.shape[0] == 4
    c0 = (np.float32(0.01) + np.float32(0.02)) / np.float32(2)
    assert out[0, 0] == c0 and out[0, 3] == 2.0
    assert out[1:, 3].tolist() == [5.0, 4.0, 2.0]          # z-only, then y-only, then x-only neighbour: z is the least significant axis
    assert orc.voxel_centroid(np.zeros((0, 4), np.float32), 0.05).shape[0] == 0
    # a second pass over already-centroided data cannot create points
    rng = np.random.default_rng(7)
    cloud = rng.uniform(-5, 5, (20000, 4)).astype(np.float32)
    a = orc.voxel_centroid(cloud, 0.25)
    b = orc.voxel_centroid(a, 0.25)
    assert 0 < b.shape[0] <= a.shape[0] < cloud.shape[0]


def test_vote_union_is_order_invariant_and_flags_the_obvious(orc):
    # a wall at x=10 seen by the scan, a "ghost" object at x=5 present only in the map -> the ghost is flagged dynamic
    ys, zs = np.meshgrid(np.linspace(-2, 2, 41), np.linspace(-1, 1, 21))
    wall = np.stack([np.full(ys.size, 10.0), ys.ravel(), zs.ravel(), np.zeros(ys.size)], 1).astype(np.float32)
    ghost = np.stack([np.full(ys.size, 5.0), ys.ravel() * 0.2, zs.ravel() * 0.2, np.ones(ys.size)], 1).astype(np.float32)
    cmap = np.concatenate([wall, ghost])
    off = np.array([0, len(wall), 2 * len(wall)], dtype=np.uint64)
    scans = np.concatenate([wall, wall])
    inv = np.stack([I4, I4]).reshape(2, 16)
    lab = orc.vote_labels(cmap, scans, off, inv, I4, VFOV, HFOV, 2.5, 0.1, 0)
    assert lab[len(wall):].sum() > 0 and lab[: len(wall)].sum() == 0
    lab_rev = orc.vote_labels(cmap, scans[::-1].copy(), off, inv, I4, VFOV, HFOV, 2.5, 0.1, 0)
    assert (lab == lab_rev).all()
    # ND mode (map - scan): the wall behind a scan-only object is what gets flagged
    lab_nd = orc.vote_labels(wall, np.concatenate([ghost, ghost]), off, inv, I4, VFOV, HFOV, 2.5, 0.1, 1)
    assert lab_nd.sum() > 0


def test_preclean_rule(orc):   # Session.cpp:522-524
    p = np.array([[1, 0, 0.0, 0], [1, 0, 0.6, 0], [1, 0, -0.6, 0], [3, 0, 0, 0], [2.4, 0, 0.49, 0]], np.float32)
    assert orc.preclean(p, 2.5)[:, 2].tolist() == [np.float32(0.6), np.float32(-0.6), 0.0]


@pytest.mark.slow
def test_atan2f_restatement_matches_host_libm_bitwise():
    """the pin: exhaustive atanf (2^32 inputs) + 2^26 structured atan2f pairs against this machine's glibc"""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-s", "-C", here, "pin_atan2f"])
    out = subprocess.run([os.path.join(here, "pin_atan2f"), "26", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "atanf exhaustive 2^32 inputs: 0 mismatches" in out.stdout


def test_jet_colormap_known_entries(orc):
    """cv::COLORMAP_JET anchor colours (BGR): dark blue, blue, cyan-ish, green-ish centre, yellow-ish, red, dark red; and the
    saturating 8-bit conversion of convertColorMappedImg (utility.h:114-127)"""
    lut = orc.jet_lut()
    assert tuple(lut[0]) == (128, 0, 0) and tuple(lut[255]) == (0, 0, 128)
    assert tuple(lut[32]) == (255, 0, 0) and tuple(lut[33]) == (255, 4, 0)      # blue saturates where green starts
    assert tuple(lut[96]) == (254, 255, 2) and tuple(lut[159]) == (2, 255, 254)  # cyan -> yellow through the green plateau
    assert tuple(lut[223]) == (0, 0, 255) and tuple(lut[224]) == (0, 0, 252)     # red plateau ends
    assert (np.diff(lut[:, 2].astype(int)[:200]) >= 0).all()                     # red channel rises monotonically up to its plateau
    img = np.array([[-5.0, 0.0, 10.0, 20.0, 10000.0]], np.float32)
    out = orc.colormap(img, 0.0, 20.0)
    assert (out[0, 0] == lut[0]).all() and (out[0, 1] == lut[0]).all() and (out[0, 3] == lut[255]).all() and (out[0, 4] == lut[255]).all()
    assert (out[0, 2] == lut[128]).all()                                         # 127.5 rounds half to even -> 128
    idx = np.array([[0, 50, 100]], np.int32)
    assert (orc.colormap(idx, 0.0, 100.0)[0, 2] == lut[255]).all()
