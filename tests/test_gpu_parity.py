"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): integer labels / pixel indices / range images bit-exact; output XYZ within
1e-4 m -- in practice every float below is compared BITWISE because the kernels reproduce the reference's
non-FMA arithmetic exactly; the tolerance is only quoted where the north star states one.
"""
import numpy as np
import pytest

from conftest import assert_clouds_equal

pytestmark = pytest.mark.gpu

XYZ_TOL = 1e-4   # metres, north_star
VFOV, HFOV = 50.0, 360.0
I4 = np.eye(4)


def _random_points(n, seed, scale=40.0):
    rng = np.random.default_rng(seed)
    p = rng.normal(0, scale, size=(n, 4)).astype(np.float32)
    p[:, 2] = rng.normal(0, 3.0, size=n)
    p[:, 3] = rng.uniform(0, 255, size=n)
    return p


def _random_pose(rng):
    yaw, pitch, roll = rng.uniform(-np.pi, np.pi), rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1)
    cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-30, 30, 3)
    return T


def test_device_projection_arithmetic_matches_oracle(gpu_ctx, orc):
    """atan2f / sqrtf / rad2deg / pixel index on the device vs the (glibc-pinned) oracle, elementwise"""
    rng = np.random.default_rng(1)
    n = 2_000_000
    xyz = rng.normal(0, 30, size=(n, 3)).astype(np.float32)
    # adversarial rows: axes, zeros, signed zeros, denormals, huge ratios, interval breakpoints of atanf
    special = np.array([[1, 0, 0], [-1, 0, 0], [-1, -0.0, 0], [0, 0, 1], [0, 0, -1], [0, 0, 0], [-0.0, -0.0, 0], [0, 1, 0], [0, -1, 0],
                        [1e-42, 1, 0], [1, 1e-42, 0], [-1, 1e-42, 0], [1e-30, 1e30, 1], [1e30, 1e-30, 1], [-1e30, 1e-30, 1], [3, 3, 3],
                        [1, 0.4375, 0], [1, 0.6875, 0], [1, 1.1875, 0], [1, 2.4375, 0], [1, 1, 2 ** 0.5], [1e-20, 1e-20, 1e-20],
                        [-5, 1e-7, 0.1], [-5, -1e-7, 0.1], [100, 0, 100], [100, 0, -100]], dtype=np.float32)
    xyz[: special.shape[0]] = special
    for alpha in (2.5, 3.0):
        sph, rc = gpu_ctx.debug_project(xyz, alpha)
        R, C = orc.rimg_size(VFOV, HFOV, alpha)
        az = orc.atan2f(xyz[:, 1], xyz[:, 0])
        el = orc.atan2f(xyz[:, 2], np.sqrt((xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]).astype(np.float32)).astype(np.float32))
        assert (sph[:, 0].view(np.uint32) == az.view(np.uint32)).all(), "azimuth bits differ"
        assert (sph[:, 1].view(np.uint32) == el.view(np.uint32)).all(), "elevation bits differ"
        k = 20000
        orc_rc, orc_r = orc.pixel(xyz[:k], VFOV, HFOV, R, C)
        assert (rc[:k] == orc_rc).all(), "pixel indices differ"
        assert (sph[:k, 2].view(np.uint32) == orc_r.view(np.uint32)).all(), "range bits differ"


def test_range_image_bit_exact(gpu_ctx, orc):
    rng = np.random.default_rng(2)
    pts = _random_points(300_000, 3)
    pts[:50] = pts[50:100]          # duplicates: lowest index must win the tie
    T = _random_pose(rng)
    Tinv = np.linalg.inv(T)
    B2L = np.linalg.inv(_random_pose(np.random.default_rng(5)))
    cloud = gpu_ctx.upload(pts)
    for alpha, t1, t2 in ((2.5, None, None), (2.375, Tinv, None), (3.0, Tinv, B2L), (1.5, Tinv, I4)):
        R, C = orc.rimg_size(VFOV, HFOV, alpha)
        g_r, g_i = gpu_ctx.debug_range_image(cloud, alpha, t1, t2)
        o_r, o_i = orc.range_image(pts, VFOV, HFOV, R, C, t1, t2)
        assert g_r.shape == (R, C)
        assert (g_r.view(np.uint32) == o_r.view(np.uint32)).all(), f"range image differs at alpha={alpha}"
        assert (g_i == o_i).all(), f"arg-min image differs at alpha={alpha}"


@pytest.mark.parametrize("mode", [0, 1])
def test_vote_labels_and_partition(gpu_ctx, orc, small_pair, mode):
    C, Q = small_pair
    b2l = I4
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), 0.05)
    src = C if mode == 0 else Q
    labels_o = orc.vote_labels(cmap, src["scans"], src["offsets"], src["inv"], b2l, VFOV, HFOV, 2.5, 0.1, mode)
    g_map = gpu_ctx.upload(cmap)
    g_scans = gpu_ctx.upload_scans(src["scans"], src["offsets"])
    g_poses = gpu_ctx.poses(src["poses"], src["inv"])
    kept, flagged, labels_g = gpu_ctx.visibility_partition(g_map, g_scans, g_poses, 2.5, 0.1, mode, want_labels=True)
    assert labels_o.sum() > 0, "degenerate test: nothing flagged"
    assert (labels_g == labels_o).all(), f"{(labels_g != labels_o).sum()} labels differ"
    assert_clouds_equal(kept.download(), cmap[labels_o == 0], "kept")
    assert_clouds_equal(flagged.download(), cmap[labels_o == 1], "flagged")


def test_vote_keyframe_batches_and_shards_union(gpu_ctx, orc, small_pair, ltm):
    """sharding keyframes over ranks and OR-ing labels == one pass (the multi-GPU exchange contract)"""
    import torch
    C, _ = small_pair
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), 0.05)
    ctx2 = ltm.Context(vfov=VFOV, hfov=HFOV, device=0, max_kf_batch=2)   # force several image batches
    g_map = ctx2.upload(cmap)
    g_scans = ctx2.upload_scans(C["scans"], C["offsets"])
    g_poses = ctx2.poses(C["poses"], C["inv"])
    nk = g_poses.n
    full = torch.zeros(len(cmap), dtype=torch.uint8, device="cuda")
    ctx2.visibility_vote(g_map, g_scans, g_poses, 0, nk, 2.5, 0.1, 0, full.data_ptr())
    parts = torch.zeros(len(cmap), dtype=torch.uint8, device="cuda")
    for a, b in ((0, 1), (1, 4), (4, nk)):
        shard = torch.zeros(len(cmap), dtype=torch.uint8, device="cuda")
        ctx2.visibility_vote(g_map, g_scans, g_poses, a, b, 2.5, 0.1, 0, shard.data_ptr())
        parts = torch.maximum(parts, shard)
    labels_o = orc.vote_labels(cmap, C["scans"], C["offsets"], C["inv"], I4, VFOV, HFOV, 2.5, 0.1, 0)
    assert (full.cpu().numpy() == labels_o).all()
    assert (parts.cpu().numpy() == labels_o).all()
    kept, flagged = ctx2.partition_by_labels(g_map, full.data_ptr())
    assert len(kept) + len(flagged) == len(cmap) and len(flagged) == int(labels_o.sum())
    ctx2.close()


def test_voxel_centroid_matches_oracle(gpu_ctx, orc, small_pair):
    C, _ = small_pair
    merged = orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4)
    for leaf in (0.05, 0.4, 1.0):
        o = orc.voxel_centroid(merged, leaf)
        g = gpu_ctx.voxel_centroid(gpu_ctx.upload(merged), leaf).download()
        assert_clouds_equal(g, o, f"voxel centroid leaf={leaf}")
    # idempotence-like property at full size independence: a second pass keeps the count non-increasing
    g1 = gpu_ctx.voxel_centroid(gpu_ctx.upload(merged), 0.05)
    g2 = gpu_ctx.voxel_centroid(g1, 0.05)
    assert len(g2) <= len(g1)
    assert_clouds_equal(g2.download(), orc.voxel_centroid(g1.download(), 0.05), "second voxel pass")
    # degenerate inputs
    assert len(gpu_ctx.voxel_centroid(gpu_ctx.upload(np.zeros((0, 4), np.float32)), 0.05)) == 0
    one = np.array([[1, 2, 3, 4]], np.float32)
    assert_clouds_equal(gpu_ctx.voxel_centroid(gpu_ctx.upload(one), 0.05).download(), orc.voxel_centroid(one, 0.05), "single point")
    same = np.repeat(one, 1000, axis=0)
    assert_clouds_equal(gpu_ctx.voxel_centroid(gpu_ctx.upload(same), 0.05).download(), orc.voxel_centroid(same, 0.05), "all in one voxel")


def test_voxel_centroid_shards_concatenate_to_the_unsharded_output(gpu_ctx, orc, small_pair):
    """multi-GPU form: every shard owns a contiguous Morton-key range; shards in order must reproduce the full output exactly"""
    C, _ = small_pair
    merged = orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4)
    g_in = gpu_ctx.upload(merged)
    for leaf in (0.05, 0.4):
        want = orc.voxel_centroid(merged, leaf)
        for n_shards in (1, 2, 3, 8, 61):
            parts = [gpu_ctx.voxel_centroid_shard(g_in, leaf, r, n_shards).download() for r in range(n_shards)]
            assert_clouds_equal(np.concatenate(parts), want, f"voxel shards leaf={leaf} n={n_shards}")
            if n_shards in (2, 8) and leaf == 0.05:     # balanced: no shard more than twice the mean
                assert max(len(p) for p in parts) <= 2 * len(want) / n_shards + 16
    # degenerate: empty input, everything in one voxel (all but one shard empty)
    assert len(gpu_ctx.voxel_centroid_shard(gpu_ctx.upload(np.zeros((0, 4), np.float32)), 0.05, 1, 2)) == 0
    same = np.repeat(np.array([[1, 2, 3, 4]], np.float32), 1000, axis=0)
    parts = [gpu_ctx.voxel_centroid_shard(gpu_ctx.upload(same), 0.05, r, 4).download() for r in range(4)]
    assert_clouds_equal(np.concatenate(parts), orc.voxel_centroid(same, 0.05), "one voxel, four shards")
    with pytest.raises(Exception):
        gpu_ctx.voxel_centroid_shard(g_in, 0.05, 2, 2)


@pytest.mark.parametrize("mode", [0, 1])
def test_viz_images_match_oracle_colormap(gpu_ctx, orc, small_pair, mode):
    """the four RViz images (pubRangeImg x4, Removerter.cpp:580-585) colour-mapped on the device"""
    C, _ = small_pair
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), 0.05)
    g_map, g_scans, g_poses = gpu_ctx.upload(cmap), gpu_ctx.upload_scans(C["scans"], C["offsets"]), gpu_ctx.poses(C["poses"], C["inv"])
    kf, alpha = 3, 2.5
    rows, cols = orc.rimg_size(50.0, 360.0, alpha)
    a, b = int(C["offsets"][kf]), int(C["offsets"][kf + 1])
    scan_r, _ = orc.range_image(C["scans"][a:b], 50.0, 360.0, rows, cols, want_idx=False)
    map_r, map_i = orc.range_image(cmap, 50.0, 360.0, rows, cols, T1=C["inv"][kf], T2=I4)
    diff = (scan_r - map_r) if mode == 0 else (map_r - scan_r)
    got = gpu_ctx.viz_images(g_map, g_scans, g_poses, kf, alpha, mode=mode, range_axis=(0.0, 20.0), diff_axis=(0.0, 0.5))
    np.testing.assert_array_equal(got["scan"], orc.colormap(scan_r, 0.0, 20.0))
    np.testing.assert_array_equal(got["map"], orc.colormap(map_r, 0.0, 20.0))
    np.testing.assert_array_equal(got["diff"], orc.colormap(diff, 0.0, 0.5))
    np.testing.assert_array_equal(got["ptidx"], orc.colormap(map_i, 0.0, float(len(cmap))))
    assert len(np.unique(got["map"].reshape(-1, 3), axis=0)) > 20        # a real picture, not a constant


def test_cloud_transform_matches_pcl_semantics(gpu_ctx, orc, small_pair):
    """ltm_cloud_transform = pcl::transformPointCloud<double> once or twice (local2global / global2local / transformGlobalMapToLocal)"""
    C, _ = small_pair
    pts = C["scans"][:20000]
    g = gpu_ctx.upload(pts)
    l2b = np.eye(4); l2b[:3, :3] = [[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]; l2b[:3, 3] = [0.3, -0.1, 0.25]
    pose, inv = C["poses"][2].reshape(4, 4), C["inv"][2].reshape(4, 4)
    assert_clouds_equal(gpu_ctx.transform(g, l2b, pose).download(), orc.transform(pose, orc.transform(l2b, pts)), "local2global")
    assert_clouds_equal(gpu_ctx.transform(g, inv, np.linalg.inv(l2b)).download(), orc.transform(np.linalg.inv(l2b), orc.transform(inv, pts)), "global2local")
    assert_clouds_equal(gpu_ctx.transform(g, pose).download(), orc.transform(pose, pts), "single transform")
    assert_clouds_equal(gpu_ctx.transform(g, None, pose).download(), orc.transform(pose, pts), "second only")
    assert_clouds_equal(gpu_ctx.transform(g).download(), pts, "no transform = copy")


def test_cloud_select_and_scan_of_keyframe(gpu_ctx, small_pair):
    """pcl::ExtractIndices stand-in (parsePointcloudSubsetUsingPtIdx) and the per-keyframe slice of a scan set"""
    C, _ = small_pair
    pts = C["scans"][:5000]
    g = gpu_ctx.upload(pts)
    rng = np.random.default_rng(3)
    for idx in (np.arange(0, 5000, 7), rng.integers(0, 5000, 999), np.array([], np.int64), np.array([4999, 0, 0])):
        np.testing.assert_array_equal(gpu_ctx.select(g, idx).download(), pts[idx])
    for bad in ([5000], [-1]):
        with pytest.raises(Exception):
            gpu_ctx.select(g, bad)
    g_scans = gpu_ctx.upload_scans(C["scans"], C["offsets"])
    for kf in (0, 3):
        a, b = int(C["offsets"][kf]), int(C["offsets"][kf + 1])
        np.testing.assert_array_equal(gpu_ctx.scan_of_keyframe(g_scans, kf).download(), C["scans"][a:b])
    with pytest.raises(Exception):
        gpu_ctx.scan_of_keyframe(g_scans, len(C["offsets"]) - 1)


def test_merge_and_preclean(gpu_ctx, orc, small_pair):
    C, _ = small_pair
    g_scans = gpu_ctx.upload_scans(C["scans"], C["offsets"])
    g_poses = gpu_ctx.poses(C["poses"], C["inv"])
    assert_clouds_equal(gpu_ctx.merge_to_global(g_scans, g_poses).download(), orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), "merge")
    cleaned = gpu_ctx.preclean(g_scans, 2.5)
    pts, off = cleaned.download()
    for k in range(len(C["offsets"]) - 1):
        a, b = int(C["offsets"][k]), int(C["offsets"][k + 1])
        assert_clouds_equal(pts[int(off[k]):int(off[k + 1])], orc.preclean(C["scans"][a:b], 2.5), f"preclean kf {k}")


def test_reproject_matches_oracle(gpu_ctx, orc, small_pair):
    C, _ = small_pair
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), 0.05)
    g = gpu_ctx.reproject(gpu_ctx.upload(cmap), gpu_ctx.poses(C["poses"], C["inv"]), 3.0)
    g_pts, g_off = g.download()
    o_pts, o_off = orc.reproject(cmap, C["inv"], I4, VFOV, HFOV, 3.0)
    assert (g_off == o_off).all(), "per-keyframe counts differ"
    assert_clouds_equal(g_pts, o_pts, "reprojected scans")
    # quirk Q3: map point 0 is never emitted
    two = np.array([[5, 0, 0, 1], [0, 5, 0, 2]], np.float32)
    gp, go = gpu_ctx.reproject(gpu_ctx.upload(two), gpu_ctx.poses(I4.reshape(1, 16), I4.reshape(1, 16)), 3.0).download()
    assert gp.shape[0] == 1 and gp[0, 3] == 2.0


def test_survivor_queue_overflow_paths(gpu_ctx, orc, small_pair):
    """both projection kernels fall back to the exact path for a whole tile when their LDS survivor queue (2048 of 4096 points)
    overflows: (a) a vote whose scans lie far behind the map, so every map point 'matters'; (b) a reprojection of a sparse
    shell of points that all own their pixel, so every point survives the block-local arg-min filter"""
    C, _ = small_pair
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), 0.05)
    far = C["scans"].copy()
    far[:, :3] *= 1.6                                             # the scans now see 60 % farther than the map surface
    labels_o = orc.vote_labels(cmap, far, C["offsets"], C["inv"], I4, VFOV, HFOV, 2.5, 0.1, 0)
    assert labels_o.mean() > 0.05
    _, _, labels_g = gpu_ctx.visibility_partition(gpu_ctx.upload(cmap), gpu_ctx.upload_scans(far, C["offsets"]), gpu_ctx.poses(C["poses"], C["inv"]),
                                                  2.5, 0.1, 0, want_labels=True)
    assert (labels_g == labels_o).all(), f"{(labels_g != labels_o).sum()} labels differ"
    surv, pts = gpu_ctx.cull_stats()
    rng = np.random.default_rng(5)
    n = 40000
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:, 2] *= 0.3
    shell = np.concatenate([d * rng.uniform(20, 60, (n, 1)), rng.uniform(0, 255, (n, 1))], axis=1).astype(np.float32)
    poses = np.tile(I4.reshape(1, 16), (3, 1))
    g_pts, g_off = gpu_ctx.reproject(gpu_ctx.upload(shell), gpu_ctx.poses(poses, poses), 3.0).download()
    o_pts, o_off = orc.reproject(shell, poses, I4, VFOV, HFOV, 3.0)
    assert (g_off == o_off).all() and int(o_off[1]) > 0.8 * n * 0.5
    assert_clouds_equal(g_pts, o_pts, "reprojection of a sparse shell")


@pytest.mark.parametrize("vfov,hfov", [(30.5, 120.25), (26.9, 360.0), (90.0, 180.0)])
def test_vote_and_reproject_with_other_fovs_match_oracle(ltm, orc, small_pair, vfov, hfov):
    """most of the map falls outside a narrow FOV and is clamped into the edge rows / columns (utility.cpp:122-123, quirk Q2):
    the culled vote, the exact-image vote and the reprojection must still agree with the oracle bit for bit"""
    C, Q = small_pair
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), 0.05)
    ctx = ltm.Context(vfov=vfov, hfov=hfov, device=0)
    g_map = ctx.upload(cmap)
    for mode, src in ((0, C), (1, Q)):
        g_scans, g_poses = ctx.upload_scans(src["scans"], src["offsets"]), ctx.poses(src["poses"], src["inv"])
        for alpha in (2.5, 1.425):
            want = orc.vote_labels(cmap, src["scans"], src["offsets"], src["inv"], I4, vfov, hfov, alpha, 0.1, mode)
            _, _, got = ctx.visibility_partition(g_map, g_scans, g_poses, alpha, 0.1, mode, want_labels=True)
            assert (got == want).all(), f"fov ({vfov},{hfov}) alpha {alpha} mode {mode}: {(got != want).sum()} labels differ"
    g_pts, g_off = ctx.reproject(g_map, ctx.poses(C["poses"], C["inv"]), 3.0).download()
    o_pts, o_off = orc.reproject(cmap, C["inv"], I4, vfov, hfov, 3.0)
    assert (g_off == o_off).all()
    assert_clouds_equal(g_pts, o_pts, f"reprojected scans, fov ({vfov},{hfov})")
    ctx.close()


@pytest.mark.parametrize("k,thr", [(2, 0.01), (3, 0.1), (1, 0.05), (4, 0.02), (6, 0.05)])   # k <= 4: register specialisations, 6: generic path
def test_knn_partition_matches_oracle(gpu_ctx, orc, small_pair, k, thr):
    C, Q = small_pair
    target = orc.voxel_centroid(orc.merge_to_global(Q["scans"], Q["offsets"], Q["poses"], I4), 0.05)
    co_o, loc_o = orc.knn_labels(target, C["scans"], C["offsets"], C["poses"], C["inv"], I4, k, thr)
    g_co, g_di = gpu_ctx.knn_partition(gpu_ctx.upload(target), gpu_ctx.upload_scans(C["scans"], C["offsets"]),
                                       gpu_ctx.poses(C["poses"], C["inv"]), k, thr)
    assert 0 < co_o.sum() < co_o.size, "degenerate test"
    co_pts, co_off = g_co.download()
    di_pts, di_off = g_di.download()
    for kf in range(len(C["offsets"]) - 1):
        a, b = int(C["offsets"][kf]), int(C["offsets"][kf + 1])
        m = co_o[a:b] == 1
        assert_clouds_equal(co_pts[int(co_off[kf]):int(co_off[kf + 1])], loc_o[a:b][m], f"coexist kf {kf}")
        assert_clouds_equal(di_pts[int(di_off[kf]):int(di_off[kf + 1])], loc_o[a:b][~m], f"diff kf {kf}")


def test_knn_kat_and_small_targets(gpu_ctx, orc):
    """SURVEY Appendix B KAT: neighbours at 0.1 m and 0.1 m, thr 0.01 => avg == 0.01 => diff (strict <)"""
    q = np.array([[0, 0, 0, 0]], np.float32)
    t = np.array([[0.1, 0, 0, 0], [0, 0.1, 0, 0], [3, 3, 3, 0]], np.float32)
    for tt in (t, np.vstack([t, _random_points(500, 9, 50.0) + 100])):
        for thr in (0.01, 0.0101, 0.0099):
            near_o = orc.knn_split(tt, q, 2, thr)
            near_g, far_g = gpu_ctx.knn_split_cloud(gpu_ctx.upload(tt), gpu_ctx.upload(q), 2, thr)
            assert len(near_g) == int(near_o.sum()) and len(far_g) == 1 - int(near_o.sum())
    # k larger than the target (PCL clamps k, the divisor stays k)
    t1 = np.array([[0.05, 0, 0, 0]], np.float32)
    assert len(gpu_ctx.knn_split_cloud(gpu_ctx.upload(t1), gpu_ctx.upload(q), 2, 0.01)[0]) == int(orc.knn_split(t1, q, 2, 0.01).sum())
    # weak->strong ND parameters (k=2, thr=1.0) on random clouds, brute-force oracle
    tgt, qry = _random_points(4000, 11, 5.0), _random_points(3000, 12, 5.0)
    near_o = orc.knn_split(tgt, qry, 2, 1.0, use_kdtree=False)
    near_g, far_g = gpu_ctx.knn_split_cloud(gpu_ctx.upload(tgt), gpu_ctx.upload(qry), 2, 1.0)
    assert_clouds_equal(near_g.download(), qry[near_o == 1], "near")
    assert_clouds_equal(far_g.download(), qry[near_o == 0], "far")


def test_empty_and_ragged_inputs(gpu_ctx, orc):
    empty = np.zeros((0, 4), np.float32)
    pts = _random_points(1000, 21)
    off = np.array([0, 0, 400, 400, 1000], dtype=np.uint64)      # empty keyframes in the middle
    poses = np.stack([np.eye(4)] * 4).reshape(4, 16)
    g_scans = gpu_ctx.upload_scans(pts, off)
    g_poses = gpu_ctx.poses(poses, poses)
    cmap = _random_points(5000, 22)
    kept, flagged, lab = gpu_ctx.visibility_partition(gpu_ctx.upload(cmap), g_scans, g_poses, 2.5, 0.1, 0, want_labels=True)
    lab_o = orc.vote_labels(cmap, pts, off, poses, I4, VFOV, HFOV, 2.5, 0.1, 0)
    assert (lab == lab_o).all()
    k2, f2 = gpu_ctx.visibility_partition(gpu_ctx.upload(empty), g_scans, g_poses, 2.5)
    assert len(k2) == 0 and len(f2) == 0
    rp, ro = gpu_ctx.reproject(gpu_ctx.upload(empty), g_poses, 3.0).download()
    assert rp.shape[0] == 0 and (ro == 0).all()
    co, di = gpu_ctx.knn_partition(gpu_ctx.upload(empty), g_scans, g_poses, 2, 0.01)
    assert co.info()[1] == 0 and di.info()[1] == 1000            # nothing coexists with an empty map
    assert (di.offsets() == off).all()


def test_fast_math_selfcheck_is_exhaustive_and_clean(gpu_ctx, ltm):
    """the create-time device self-check compares the fast rad2deg / divide-by-FOV forms with plain IEEE division over
    all 2^32 binary32 inputs; the fast forms are only enabled when no input differs"""
    mism, enabled = gpu_ctx.selfcheck()
    assert mism == [0, 0, 0] and enabled, f"fast arithmetic forms rejected for fov (50,360): {mism}"
    for vfov, hfov in ((45.0, 360.0), (26.9, 360.0), (30.5, 120.25)):
        c = ltm.Context(vfov=vfov, hfov=hfov, device=0)
        m, on = c.selfcheck()
        assert on == (m == [0, 0, 0])     # whatever the verdict, it must be the one that gates the fast path
        c.close()


def test_voxel_centroid_scanset_matches_per_keyframe_oracle(gpu_ctx, orc, small_pair):
    C, _ = small_pair
    off = C["offsets"].copy()
    pts = C["scans"].copy()
    # make it ragged: an empty keyframe, a single-point keyframe
    off2 = np.array([0, 0, int(off[2]), int(off[2]) + 1, int(off[4]), int(off[-1])], dtype=np.uint64)
    g = gpu_ctx.voxel_centroid_scanset(gpu_ctx.upload_scans(pts, off2), 0.05)
    g_pts, g_off = g.download()
    at = 0
    for k in range(len(off2) - 1):
        want = orc.voxel_centroid(pts[int(off2[k]):int(off2[k + 1])], 0.05)
        assert int(g_off[k]) == at and int(g_off[k + 1]) == at + len(want)
        assert_clouds_equal(g_pts[at:at + len(want)], want, f"scanwise voxel kf {k}")
        at += len(want)


def test_cull_projection_bounds_hold(gpu_ctx):
    """the bounded-error projection that decides which points get the exact arithmetic must contain the exact pixel and
    bracket the exact range for every point (lidar-like, tiny, huge, and points sitting on pixel boundaries)"""
    rng = np.random.default_rng(31)
    n = 4_000_000
    sets = [rng.normal(0, 30, (n, 3)), rng.normal(0, 0.5, (n, 3)), rng.normal(0, 3000, (n, 3)),
            rng.normal(0, 30, (n, 3)) * np.array([1.0, 1.0, 0.02])]
    # points placed on azimuth/elevation pixel boundaries (+- a few ulp)
    k = np.arange(n) % 900
    az = np.deg2rad((k + 0.5) / 2.5 - 180.0 + rng.normal(0, 2e-5, n))
    el = np.deg2rad(25.0 - (np.arange(n) % 125 + 0.5) / 2.5 + rng.normal(0, 2e-5, n))
    r = rng.uniform(0.5, 120, n)
    sets.append(np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1))
    for alpha in (2.5, 2.375, 1.5):
        for pts in sets:
            assert gpu_ctx.cull_check(pts.astype(np.float32), alpha) == 0
    # the distrust band is proportional to the resolution (Geom::cull_eps_px): boundary points of EVERY resolution, with a noise
    # of the order of the band itself (tools/eps_sweep.py finds the first misses at 1/6 of the shipped band, at every alpha)
    from tools.eps_sweep import boundary_points
    for alpha in (1.5, 2.5, 4.0, 6.0):
        for sigma in (1e-6, 4e-6, 1.5e-5):
            for rmin, rmax in ((0.3, 3.0), (3.0, 150.0), (150.0, 9000.0)):
                pts = boundary_points(rng, 1_000_000, alpha, VFOV, HFOV, sigma, rmin, rmax)
                assert gpu_ctx.cull_check(pts, alpha) == 0, f"alpha {alpha} sigma {sigma} r {rmin}-{rmax}"
    # with the keyframe transform: exact fp64 PCL form vs the sensor-centred float-float form used by the cull test
    prng = np.random.default_rng(32)
    for trial in range(6):
        T = _random_pose(prng)
        T[:3, 3] *= (1, 1, 0.05) if trial % 2 else (30, 30, 1)        # sensors up to ~1 km from the origin
        Tinv = np.linalg.inv(T)
        for pts in (sets[0], sets[3], sets[4]):
            glob = (T[:3, :3] @ pts[:1_000_000].T).T + T[:3, 3]
            assert gpu_ctx.cull_check(glob.astype(np.float32), 2.5, Tinv) == 0


def test_cull_projection_bounds_hold_with_an_extrinsic(ltm):
    """same bound with a non-identity LiDAR->base extrinsic: the exact side is the reference's two-step transform
    (inverse pose, float store, base->lidar), the approximate side composes both in double"""
    rng = np.random.default_rng(33)
    l2b = _random_pose(rng)
    l2b[:3, 3] = [0.8, -0.3, 1.7]
    ctx = ltm.Context(vfov=VFOV, hfov=HFOV, lidar2base=l2b, device=0)
    n = 1_000_000
    local = [rng.normal(0, 30, (n, 3)), rng.normal(0, 0.7, (n, 3)), rng.normal(0, 30, (n, 3)) * np.array([1.0, 1.0, 0.02])]
    k = np.arange(n) % 900
    az = np.deg2rad((k + 0.5) / 2.5 - 180.0 + rng.normal(0, 2e-5, n))
    el = np.deg2rad(25.0 - (np.arange(n) % 125 + 0.5) / 2.5 + rng.normal(0, 2e-5, n))
    r = rng.uniform(0.5, 120, n)
    local.append(np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1))
    for trial in range(4):
        T = _random_pose(rng)
        T[:3, 3] *= (1, 1, 0.05) if trial % 2 else (30, 30, 1)
        Tinv = np.linalg.inv(T)
        for pts in local:
            base = (l2b[:3, :3] @ pts.T).T + l2b[:3, 3]
            glob = (T[:3, :3] @ base.T).T + T[:3, 3]
            assert ctx.cull_check(glob.astype(np.float32), 2.5, Tinv) == 0
    ctx.close()


def test_vote_sentinel_arithmetic_beyond_9800_m(gpu_ctx, orc):
    """Removerter.cpp:398-404 with kFlagNoPOINT = 10000 (utility.h:93): diff = scan - map with an EMPTY scan pixel is 10000 - r,
    which lies in (0.1, 200) for a map point 9800..9999.9 m away -- the reference flags it.  Points at 9850 m / 9999 m / 10050 m on
    empty and on occupied scan pixels, plus a far scan return (9040 m) in front of a map point at 8850 m (diff 190: flagged) and one
    at 8830 m (diff 210: not), through the range-culled kernel, the tile cull and both modes."""
    rng = np.random.default_rng(41)
    # scan of one keyframe at the origin: a ring of returns at 20 m in the sector az in [0, 90) deg, everything else empty,
    # plus single far returns
    az = np.deg2rad(np.arange(0, 90, 0.2)); el = np.deg2rad(np.arange(-20, 21, 1.0))
    A, E = np.meshgrid(az, el)
    ring = np.stack([20 * np.cos(E) * np.cos(A), 20 * np.cos(E) * np.sin(A), 20 * np.sin(E)], -1).reshape(-1, 3)
    def at(r, az_deg, el_deg=0.0):
        a, e = np.deg2rad(az_deg), np.deg2rad(el_deg)
        return [r * np.cos(e) * np.cos(a), r * np.cos(e) * np.sin(a), r * np.sin(e)]
    far_scan = np.array([at(9040.0, 200.0), at(9040.0, 210.0), at(9900.0, 220.0)])
    scan = np.concatenate([ring, far_scan]).astype(np.float32)
    scan = np.concatenate([scan, np.zeros((len(scan), 1), np.float32)], 1)
    # map: a filler so that tiles exist, then the probes
    probes = [at(9850.0, 150.0), at(9999.0, 160.0), at(10050.0, 170.0),            # empty scan pixels
              at(9850.0, 30.0), at(9999.0, 40.0), at(10050.0, 50.0),               # occupied (20 m) scan pixels: diff << 0
              at(9799.0, 180.0), at(9801.0, 185.0), at(9999.95, 190.0),            # around the interval ends
              at(8850.0, 200.0), at(8830.0, 210.0), at(9850.0, 220.0),             # behind far scan returns: 190 (flag), 210 (no), 50 (flag)
              at(19.0, 10.0), at(25.0, 20.0)]                                      # ordinary: flagged (diff 1), occluded
    fr, fa = rng.uniform(3, 80, 20000), np.deg2rad(rng.uniform(-100, 100, 20000))  # filler away from the probes' azimuths (150..220 deg)
    filler = np.stack([fr * np.cos(fa), fr * np.sin(fa), rng.normal(0, 2, 20000)], 1)
    far_cluster = np.array(at(9850.0, 140.0)) + rng.normal(0, 3.0, (6000, 3))      # a whole tile out there, on empty scan pixels
    m = np.concatenate([filler, np.array(probes), far_cluster]).astype(np.float32)
    cmap = np.concatenate([m, np.zeros((len(m), 1), np.float32)], 1)
    off = np.array([0, len(scan)], dtype=np.uint64)
    pose = np.eye(4).reshape(1, 16)
    g_map, g_scans, g_poses = gpu_ctx.upload(cmap), gpu_ctx.upload_scans(scan, off), gpu_ctx.poses(pose, pose)
    for mode in (0, 1):
        for alpha in (2.5, 1.5):
            want = orc.vote_labels(cmap, scan, off, pose, I4, VFOV, HFOV, alpha, 0.1, mode)
            got = gpu_ctx.visibility_partition(g_map, g_scans, g_poses, alpha, 0.1, mode, want_labels=True)[2]
            assert (got == want).all(), f"mode {mode} alpha {alpha}: labels differ at {np.nonzero(got != want)[0][:10]}"
            if mode == 0:
                p0 = len(filler)
                assert want[p0 + 0] == 1 and want[p0 + 1] == 1 and want[p0 + 2] == 0, "the oracle must show the sentinel effect"
                assert want[p0 + 9] == 1 and want[p0 + 10] == 0


def test_non_finite_map_points_never_vote_and_change_nothing(gpu_ctx, small_pair):
    """A NaN / inf coordinate gives a NaN / inf range; `r < rimg` (utility.cpp:134) is false for it, so such a point can never be
    the arg-min of a pixel in the reference (its pixel index there is undefined behaviour: not compared).  The range-culled vote
    drops it in phase 1, the exact kernels let it lose every min: either way the labels of the other points are those of the map
    without it, and its own label stays 0.  Full tiles and the partial tail are both covered (12 000 clean points + sprinkled ones)."""
    C, _ = small_pair
    rng = np.random.default_rng(41)
    clean = np.ascontiguousarray(C["scans"][:12000], dtype=np.float32).copy()
    clean[:, :3] = (C["poses"][0].reshape(4, 4)[:3, :3] @ clean[:, :3].T).T + C["poses"][0].reshape(4, 4)[:3, 3]   # into the map frame
    bad = np.tile(clean[:64], (1, 1)).copy()
    for i, v in enumerate((np.nan, np.inf, -np.inf, 3.0e38)):
        bad[i::12, i % 3] = v
        bad[i + 4::12, :3] = v
    where = np.sort(rng.choice(len(clean) + len(bad), len(bad), replace=False))
    mixed = np.empty((len(clean) + len(bad), 4), np.float32)
    is_bad = np.zeros(len(mixed), bool); is_bad[where] = True
    mixed[is_bad], mixed[~is_bad] = bad, clean
    g_scans, g_poses = gpu_ctx.upload_scans(C["scans"], C["offsets"]), gpu_ctx.poses(C["poses"], C["inv"])
    for mode in (0, 1):
        for alpha in (2.5, 1.5):
            ref = gpu_ctx.visibility_partition(gpu_ctx.upload(clean), g_scans, g_poses, alpha, 0.1, mode, want_labels=True)[2]
            got = gpu_ctx.visibility_partition(gpu_ctx.upload(mixed), g_scans, g_poses, alpha, 0.1, mode, want_labels=True)[2]
            assert (got[~is_bad] == ref).all(), f"mode {mode} alpha {alpha}: non-finite neighbours changed {(got[~is_bad] != ref).sum()} labels"
            assert not got[is_bad].any(), f"mode {mode} alpha {alpha}: a non-finite point was flagged"
            assert ref.any(), "the clean map must have flagged points for the comparison to mean something"


def test_tile_range_cull_on_a_long_street(gpu_ctx, orc):
    """a 300 m drive: most map tiles are out of reach of any single keyframe and are culled as a whole -- labels must not change"""
    from tools import synth
    S = synth.to_numpy(synth.make_session(1, 30, "small", scene="street", kf_spacing=10.0))
    cmap = orc.voxel_centroid(orc.merge_to_global(S["scans"], S["offsets"], S["poses"], I4), 0.05)
    ext = cmap[:, :2].max(0) - cmap[:, :2].min(0)
    assert ext.max() > 350.0, "the map must be much larger than the 120 m sensor range for this test to mean anything"
    want = orc.vote_labels(cmap, S["scans"], S["offsets"], S["inv"], I4, VFOV, HFOV, 2.5, 0.1, 0)
    kept, flagged, got = gpu_ctx.visibility_partition(gpu_ctx.upload(cmap), gpu_ctx.upload_scans(S["scans"], S["offsets"]),
                                                      gpu_ctx.poses(S["poses"], S["inv"]), 2.5, 0.1, 0, want_labels=True)
    assert want.sum() > 0 and (got == want).all(), f"{(got != want).sum()} labels differ"


def test_occlusion_culled_reprojection_matches_oracle(ltm, orc):
    """the occlusion-culled launch of the exact-image kernel (forced on a small map: every pair beyond 8 m is a 'far' pair) against the
    oracle's reprojection, bitwise -- street scene so that facades really hide what lies behind them"""
    import os
    from tools import synth
    S = synth.to_numpy(synth.make_session(1, 12, "small", scene="street", kf_spacing=3.0))
    cmap = orc.voxel_centroid(orc.merge_to_global(S["scans"], S["offsets"], S["poses"], np.eye(4)), 0.05)
    want_pts, want_off = orc.reproject(cmap, S["inv"], np.eye(4), 50.0, 360.0, 3.0)
    old = {k: os.environ.get(k) for k in ("LTM_OCCLUSION", "LTM_OCCLUSION_MIN_PAIRS", "LTM_OCCLUSION_RNEAR")}
    os.environ.update(LTM_OCCLUSION="1", LTM_OCCLUSION_MIN_PAIRS="0", LTM_OCCLUSION_RNEAR="8")
    try:
        ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    got_pts, got_off = ctx.reproject(ctx.upload(cmap), ctx.poses(S["poses"], S["inv"]), 3.0).download()
    assert (got_off == want_off).all()
    assert_clouds_equal(got_pts, want_pts, "occlusion-culled reprojection")
    want = orc.vote_labels(cmap, want_pts, want_off, S["inv"], np.eye(4), 50.0, 360.0, 2.5, 0.1, 1)
    scans = ctx.upload_scans(want_pts, want_off)
    labels = ctx.visibility_partition(ctx.upload(cmap), scans, ctx.poses(S["poses"], S["inv"]), 2.5, 0.1, 1, want_labels=True)[2]
    assert (labels == want).all(), "ND-mode labels with the occlusion-culled exact images differ from the oracle"
    ctx.close()


def test_occlusion_cull_with_scaled_and_sheared_poses_matches_oracle(ltm, orc):
    """the reference accepts ANY 4x4 pose (Session.cpp:102-114).  With a scale or shear above 1 a tile's bounding sphere grows in the sensor
    frame; the occlusion cull assumes rigid poses and must switch itself off for such keyframes (only a LOWER bound of the
    smallest singular value used to gate it).  Keyframes with scale 1.08, 0.9, a shear and rigid ones mixed, cull forced on, against the oracle."""
    import os
    from tools import synth
    S = synth.to_numpy(synth.make_session(1, 12, "small", scene="street", kf_spacing=3.0))
    cmap = orc.voxel_centroid(orc.merge_to_global(S["scans"], S["offsets"], S["poses"], np.eye(4)), 0.05)
    inv = S["inv"].reshape(-1, 4, 4).copy()
    for k, M in ((1, np.diag([1.08, 1.08, 1.08])), (4, np.diag([0.9, 0.9, 0.9])), (6, np.array([[1, 0.06, 0], [0, 1, 0], [0.03, 0, 1.0]])), (9, np.diag([1.0, 1.004, 1.0]))):
        inv[k, :3, :] = M @ inv[k, :3, :]            # local = M * (rigid inverse pose) * global
    inv = inv.reshape(-1, 16)
    want_pts, want_off = orc.reproject(cmap, inv, np.eye(4), 50.0, 360.0, 3.0)
    old = {k: os.environ.get(k) for k in ("LTM_OCCLUSION", "LTM_OCCLUSION_MIN_PAIRS", "LTM_OCCLUSION_RNEAR")}
    os.environ.update(LTM_OCCLUSION="1", LTM_OCCLUSION_MIN_PAIRS="0", LTM_OCCLUSION_RNEAR="8")
    try:
        ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    got_pts, got_off = ctx.reproject(ctx.upload(cmap), ctx.poses(S["poses"], inv), 3.0).download()
    assert (got_off == want_off).all()
    assert_clouds_equal(got_pts, want_pts, "occlusion-culled reprojection under non-rigid poses")
    ctx.close()


def test_repeated_votes_over_shrinking_maps_match_oracle(gpu_ctx, orc, small_pair):
    """second and third vote of the same scans at the same resolution over shrinking, re-gridded maps (the pattern of selfRemovert; the
    scan images and bound images are served from the context's cache) against the oracle"""
    C, _ = small_pair
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], np.eye(4)), 0.05)
    scans = gpu_ctx.upload_scans(C["scans"], C["offsets"])
    poses = gpu_ctx.poses(C["poses"], C["inv"])
    cur = cmap
    for it in range(3):
        assert len(cur) > 4096
        want = orc.vote_labels(cur, C["scans"], C["offsets"], C["inv"], np.eye(4), 50.0, 360.0, 2.5, 0.1, 0)
        got = gpu_ctx.visibility_partition(gpu_ctx.upload(cur), scans, poses, 2.5, 0.1, 0, want_labels=True)[2]
        assert (got == want).all(), f"vote {it}: {(got != want).sum()} labels differ"
        keep = want == 0
        keep[::97] = True          # keep a few flagged points too, drop a few unflagged ones
        keep[5::89] = False
        cur = orc.voxel_centroid(cur[keep], 0.05)


def test_voxel_identity_shortcut_and_fused_tail_equal_the_plain_path(ltm, orc, small_pair):
    """round 4: (a) the fused head-flag / scan / segment-start kernel (single pass, decoupled look-back) against the four-kernel form;
    (b) the "already gridded under this frame" shortcut taken during the bounding-box pass: kept / flagged parts of a gridded map that keep
    their octree frame must come back bit-identical to the sort + centroid path and to the oracle -- and parts that lose their bounding
    box must NOT take it"""
    import os

    def ctx_with(**env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            return ltm.Context(vfov=VFOV, hfov=HFOV, device=0)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

    C, _ = small_pair
    merged = orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4)
    rng = np.random.default_rng(4)
    big = np.concatenate([merged, merged + np.float32(0.013), merged * np.float32(1.7)])          # ~3x the points: several look-back tiles per wave of tiles
    want_big = orc.voxel_centroid(big, 0.05)
    outs = {}
    for tag, env in (("default", {}), ("plain", dict(LTM_VOXEL_FUSED_TAIL=0, LTM_VOXEL_IDENTITY=0))):
        ctx = ctx_with(**env)
        g = ctx.voxel_centroid(ctx.upload(big), 0.05)
        assert_clouds_equal(g.download(), want_big, f"{tag}: voxel grid of {len(big)} points")
        rng = np.random.default_rng(41)                       # the same labels under both configurations
        labels = (rng.uniform(size=len(g)) < 0.07).astype(np.uint8)
        pts = g.download()
        interior = np.ones(len(pts), bool)                      # keep the extreme points so that the bounding box (hence the frame) survives
        for d in range(3):
            interior[[pts[:, d].argmin(), pts[:, d].argmax()]] = False
        labels &= interior.astype(np.uint8)
        import torch
        lab_dev = torch.from_numpy(labels).cuda()
        ctx.voxel_stats(reset=True)
        kept, flagged = ctx.partition_by_labels(g, lab_dev.data_ptr())
        k2 = ctx.voxel_centroid(kept, 0.05)                     # frame survives: the shortcut applies
        f2 = ctx.voxel_centroid(flagged, 0.05)                  # a sparse subset: its box shrinks, the frame changes -> full path
        k3, f3 = ctx.voxel_centroid_batch([kept, flagged], [0.05, 0.05])
        grids, hits = ctx.voxel_stats()
        assert_clouds_equal(k2.download(), orc.voxel_centroid(pts[labels == 0], 0.05), f"{tag}: re-grid of the kept part")
        assert_clouds_equal(f2.download(), orc.voxel_centroid(pts[labels == 1], 0.05), f"{tag}: re-grid of the flagged part")
        assert_clouds_equal(k3.download(), k2.download(), f"{tag}: batch form, kept")
        assert_clouds_equal(f3.download(), f2.download(), f"{tag}: batch form, flagged")
        assert grids == 4
        assert hits == (2 if tag == "default" else 0), f"{tag}: {hits} identity hits"
        # must NOT take the shortcut: another leaf size than the one the cloud was gridded with
        ctx.voxel_stats(reset=True)
        assert_clouds_equal(ctx.voxel_centroid(kept, 0.4).download(), orc.voxel_centroid(pts[labels == 0], 0.4), f"{tag}: other leaf")
        assert ctx.voxel_stats()[1] == 0
        # a second generation: the re-gridded kept part, partitioned again, still carries the frame
        if tag == "default":
            lab2 = torch.from_numpy((rng.uniform(size=len(k2)) < 0.5).astype(np.uint8) & interior[labels == 0].astype(np.uint8)).cuda()
            kk, _ = ctx.partition_by_labels(k2, lab2.data_ptr())
            ctx.voxel_stats(reset=True)
            kk2 = ctx.voxel_centroid(kk, 0.05)
            assert ctx.voxel_stats()[1] == 1
            assert_clouds_equal(kk2.download(), orc.voxel_centroid(kk.download(), 0.05), "second-generation re-grid")
        outs[tag] = (k2.download(), f2.download())
        ctx.close()
    assert (outs["default"][0].view(np.uint32) == outs["plain"][0].view(np.uint32)).all()


def test_key_range_exchange_pieces_reassemble_the_single_gpu_grid(gpu_ctx, orc, small_pair):
    """multi-GPU building blocks (include/ltm.h: ltm_cloud_bbox, ltm_voxel_key_histogram, ltm_voxel_key_split, ltm_voxel_centroid_box): a merged cloud
    cut into "rank-local" slices, every slice split by key range under the WHOLE cloud's box, the ranges reassembled in slice order and gridded under
    that box -- concatenated they must be the grid of the whole cloud, for balanced cuts of 2, 3 and 8 ranks and for degenerate cuts"""
    from ltmapper_amd.dist import ShardedOps
    C, _ = small_pair
    merged = orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4)
    want = orc.voxel_centroid(merged, 0.05)
    n = len(merged)
    for world in (2, 3, 8):
        bounds = [(n * r) // world for r in range(world + 1)]
        bounds[1] = bounds[0]                                   # one rank without points
        slices = [gpu_ctx.upload(merged[bounds[r]:bounds[r + 1]]) for r in range(world)]
        boxes = [gpu_ctx.bbox(s) for s in slices]
        assert not np.isfinite(boxes[0][0]).any(), "an empty cloud reports the box (+inf, -inf)"
        mn = np.min([b[0] for b in boxes], axis=0); mx = np.max([b[1] for b in boxes], axis=0)
        assert (mn == merged[:, :3].min(0)).all() and (mx == merged[:, :3].max(0)).all()
        hist = sum(gpu_ctx.voxel_key_histogram(s, mn, mx, 0.05).astype(np.int64) for s in slices)
        assert hist.sum() == n
        for cuts in (ShardedOps.balanced_cuts(hist, world), [0] * world + [4096], [0] + [4096] * world):
            parts = [gpu_ctx.voxel_key_split(s, mn, mx, 0.05, cuts) for s in slices]            # parts[source][destination]
            outs = []
            for dst in range(world):
                got = np.concatenate([parts[src][dst].download() for src in range(world)])      # arrival order = source order = input order
                outs.append(gpu_ctx.voxel_centroid_box(gpu_ctx.upload(got), mn, mx, 0.05).download())
            assert_clouds_equal(np.concatenate(outs), want, f"world {world} cuts {cuts[:4]}...")
