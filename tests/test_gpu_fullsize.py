"""Full-size checks on the BASELINE configs[1] workload (2x500 keyframes, OS1-64-like sensor, ~6.8 M point maps), where the
CPU oracle is too slow to be the checker: size-independent properties and equality between independent GPU code paths."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def full(ltm):
    import torch
    from tools import synth
    S = synth.make_session(1, 500, "os1-64", device="cuda:0")
    torch.cuda.synchronize()
    return S


def _ctx(ltm, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return ltm.Context(vfov=50.0, hfov=360.0, device=0)     # the kernel-variant switches are read at context creation
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _load(ctx, S):
    scans = ctx.preclean(ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5)
    poses = ctx.poses(S["poses"], S["inv"])
    cmap = ctx.voxel_centroid(ctx.merge_to_global(scans, poses), 0.05)
    return scans, poses, cmap


def test_culled_vote_equals_exact_vote_at_full_size(ltm, full):
    """k_vote_map_cull (bounded-error pre-test) must flag exactly the points the exact-image path flags: 3.4e9 point projections"""
    import torch
    labels = {}
    for cull in (1, 0):
        ctx = _ctx(ltm, LTM_VOTE_CULL=cull)
        scans, poses, cmap = _load(ctx, full)
        M = len(cmap)
        for alpha in (2.5, 2.375):
            lab = torch.zeros(M, dtype=torch.uint8, device="cuda")
            ctx.visibility_vote(cmap, scans, poses, 0, poses.n, alpha, 0.1, 0, lab.data_ptr())
            labels[(cull, alpha)] = lab.cpu().numpy()
        if cull:
            surv, pts = ctx.cull_stats()
            assert 0 < surv < 0.3 * pts, "the cull is expected to spare most points the exact path"
        ctx.close()
    for alpha in (2.5, 2.375):
        a, b = labels[(1, alpha)], labels[(0, alpha)]
        assert a.sum() > 1000, "degenerate: nothing flagged"
        assert (a == b).all(), f"alpha={alpha}: {(a != b).sum()} of {a.size} labels differ between culled and exact vote"


def test_blockmin_image_equals_plain_lds_image_at_full_size(ltm, full):
    """the workgroup-local arg-min pre-filter must give the same reprojection as the un-filtered exact kernel"""
    out = {}
    for variant in (2, 1):
        ctx = _ctx(ltm, LTM_MAP_KERNEL=variant)
        scans, poses, cmap = _load(ctx, full)
        pts, off = ctx.reproject(cmap, poses, 3.0, 0, 64).download()
        out[variant] = (pts, off)
        ctx.close()
    assert (out[1][1] == out[2][1]).all()
    assert (out[1][0].view(np.uint32) == out[2][0].view(np.uint32)).all()
    # size-independent properties of a reprojection: at most one point per pixel, none of them is map point 0's pixel winner twice
    off = out[2][1]
    assert (np.diff(off.astype(np.int64)) <= 150 * 1080).all() and off[-1] > 0


def test_partition_and_voxel_properties_at_full_size(ltm, full):
    import torch
    ctx = _ctx(ltm)
    scans, poses, cmap = _load(ctx, full)
    M = len(cmap)
    full_lab = torch.zeros(M, dtype=torch.uint8, device="cuda")
    ctx.visibility_vote(cmap, scans, poses, 0, poses.n, 2.5, 0.1, 0, full_lab.data_ptr())
    union = torch.zeros(M, dtype=torch.uint8, device="cuda")
    for a, b in ((0, 63), (63, 250), (250, 251), (251, 500)):          # shards of unequal size, as ranks would hold them
        part = torch.zeros(M, dtype=torch.uint8, device="cuda")
        ctx.visibility_vote(cmap, scans, poses, a, b, 2.5, 0.1, 0, part.data_ptr())
        union = torch.maximum(union, part)
    assert torch.equal(union, full_lab), "OR of shard labels != labels of the full pass"
    kept, flagged = ctx.partition_by_labels(cmap, full_lab.data_ptr())
    nf = int(full_lab.sum().item())
    assert len(flagged) == nf and len(kept) == M - nf
    k = kept.download()
    src = cmap.download()
    keep_idx = np.nonzero(full_lab.cpu().numpy() == 0)[0]
    assert (k.view(np.uint32) == src[keep_idx].view(np.uint32)).all(), "partition must be an order-preserving gather"
    # voxel grid: a second pass can only merge, and every output voxel is occupied exactly once (keys strictly increasing)
    v1 = ctx.voxel_centroid(kept, 0.05)
    v2 = ctx.voxel_centroid(v1, 0.05)
    assert len(v2) <= len(v1) <= len(kept)
    ctx.close()


def test_tile_cull_equals_no_tile_cull_on_kitti_scale_street(ltm):
    """street scene, 300 keyframes x 120 k rays: whole-tile range cull on/off must give identical labels (and be much faster)"""
    import time
    import torch
    from tools import synth
    S = synth.make_session(1, 300, "hdl-64e", device="cuda:0", scene="street", kf_spacing=2.0)
    torch.cuda.synchronize()
    out, dt = {}, {}
    for tc in (1, 0, "exact"):
        ctx = _ctx(ltm, **(dict(LTM_VOTE_CULL=0) if tc == "exact" else dict(LTM_TILE_CULL=tc)))
        scans, poses, cmap = _load(ctx, S)
        lab = torch.zeros(len(cmap), dtype=torch.uint8, device="cuda")
        ctx.visibility_vote(cmap, scans, poses, 0, poses.n, 2.5, 0.1, 0, lab.data_ptr())      # warm (scan images cached)
        lab.zero_()
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.visibility_vote(cmap, scans, poses, 0, poses.n, 2.5, 0.1, 0, lab.data_ptr())
        ctx.synchronize(); dt[tc] = time.perf_counter() - t0
        out[tc] = lab.cpu().numpy()
        ctx.close()
    assert out[1].sum() > 100 and (out[1] == out[0]).all()
    assert (out[1] == out["exact"]).all(), "culled vote differs from the exact-image vote on the street scene"
    print(f"tile cull: {dt[1] * 1e3:.1f} ms vs {dt[0] * 1e3:.1f} ms without, exact-image vote {dt['exact'] * 1e3:.1f} ms")


def test_two_phase_knn_equals_exact_search_at_full_size(ltm, full):
    """phase 1 of the kNN query (quantised 64-byte cell buckets) may only ever say "certainly coexist": with it switched off
    (LTM_KNN_FAST=0, the exact search for every query) both outputs must be the same scan sets, 200 keyframes of session 02 against
    the 6.8 M-point map of session 01, at the yaml parameters and at the MLS ones (larger cell)"""
    import torch
    from tools import synth
    Q = synth.make_session(2, 200, "os1-64", device="cuda:0")
    torch.cuda.synchronize()
    out = {}
    for fast in (1, 0):
        ctx = _ctx(ltm, LTM_KNN_FAST=fast, LTM_KNN_STATS=1)
        _, _, cmap = _load(ctx, full)
        q_scans, q_poses, _ = _load(ctx, Q)
        for k, thr in ((2, 0.01), (3, 0.04), (1, 0.003)):
            co, di = ctx.knn_partition(cmap, q_scans, q_poses, k, thr)
            out[(fast, k, thr)] = (co.download(), di.download())
        ctx.close()
    for k, thr in ((2, 0.01), (3, 0.04), (1, 0.003)):
        (a_co, a_di), (b_co, b_di) = out[(1, k, thr)], out[(0, k, thr)]
        assert len(a_co[0]) > 1000 and len(a_di[0]) > 1000, "degenerate: one of the two classes is empty"
        for (ap, ao), (bp, bo) in ((a_co, b_co), (a_di, b_di)):
            assert (ao == bo).all() and (ap.view(np.uint32) == bp.view(np.uint32)).all(), f"k={k} thr={thr}: two-phase and exact kNN split differ"


def test_voxel_sort_on_compressed_keys_equals_full_keys_at_full_size(ltm, full):
    """the voxel grid's radix sort leaves out Morton bits that cannot decide a comparison inside the cloud's bounding box (fewer passes):
    same centroids, same order as the sort over all 3 * depth bits -- the 22 M-point merge of a session and its 0.05 / 0.4 m grids"""
    out = {}
    for comp in (1, 0):
        ctx = _ctx(ltm, LTM_VOXEL_KEYBITS=comp)
        scans, poses, cmap = _load(ctx, full)
        coarse = ctx.voxel_centroid(cmap, 0.4)
        again = ctx.voxel_centroid_batch([cmap, coarse], [0.05, 1.0])
        out[comp] = [cmap.download(), coarse.download(), again[0].download(), again[1].download()]
        ctx.close()
    for a, b in zip(out[1], out[0]):
        assert a.shape == b.shape and len(a) > 1000 and (a.view(np.uint32) == b.view(np.uint32)).all()


def test_occlusion_culled_exact_images_equal_plain_launch_on_street_scene(ltm):
    """large-map path of the exact-image kernel (near pairs first, coarse maximum of the partial image, far pairs covered by strictly
    nearer returns dropped): reprojections and ND-mode labels must be those of the plain launch -- street scene, hdl-64e, 300 keyframes"""
    import torch
    from tools import synth
    S = synth.make_session(1, 300, "hdl-64e", device="cuda:0", scene="street", kf_spacing=2.0)
    torch.cuda.synchronize()
    out = {}
    for tag, env in (("occl", dict(LTM_OCCLUSION=1, LTM_OCCLUSION_MIN_PAIRS=0, LTM_OCCLUSION_STATS=1)), ("occl_near20", dict(LTM_OCCLUSION=1, LTM_OCCLUSION_MIN_PAIRS=0, LTM_OCCLUSION_RNEAR=20)),
                     ("plain", dict(LTM_OCCLUSION=0))):
        ctx = _ctx(ltm, **env)
        scans, poses, cmap = _load(ctx, S)
        rep = ctx.reproject(cmap, poses, 3.0)
        lab = torch.zeros(len(cmap), dtype=torch.uint8, device="cuda")
        ctx.visibility_vote(cmap, rep, poses, 0, poses.n, 2.5, 0.1, 1, lab.data_ptr())      # mode 1 (ND): exact images
        out[tag] = (rep.download(), lab.cpu().numpy())
        ctx.close()
    for tag in ("occl", "occl_near20"):
        (a, ao), la = out[tag]
        (b, bo), lb = out["plain"]
        assert (ao == bo).all() and (a.view(np.uint32) == b.view(np.uint32)).all(), f"{tag}: reprojection differs from the plain launch"
        assert (la == lb).all(), f"{tag}: ND labels differ"
    assert out["plain"][0][1][-1] > 1_000_000



def test_occlusion_cull_drops_hidden_tiles_and_keeps_the_image_at_kitti_scale(ltm):
    """configs[3] geometry at its real extent: 2000 keyframes of the hdl-64e sensor over the street grid, 45 M-point map = 11 000 tiles.  With
    the keyframes' (tile, keyframe) pairs above the cull's threshold the DEFAULT path of ltm_reproject is the occlusion-culled one; it must
    (a) really drop a large share of the pairs -- the parity runs against the oracle are too small for buildings to hide much -- and
    (b) give exactly the reprojection of the plain launch, for keyframes at the start and in the middle of the trajectory."""
    import torch
    from tools import synth
    S = synth.make_session(1, 2000, "hdl-64e", device="cuda:0", scene="street", kf_spacing=1.0)
    torch.cuda.synchronize()
    out, stats = {}, None
    for occl in (1, 0):
        ctx = _ctx(ltm, LTM_OCCLUSION=occl)
        scans, poses, cmap = _load(ctx, S)
        assert len(cmap) > 30_000_000
        parts = []
        for a, b in ((0, 256), (1000, 1256)):
            parts.append(ctx.reproject(cmap, poses, 3.0, a, b).download())
        if occl:
            stats = ctx.occlusion_stats()
        out[occl] = parts
        ctx.close()
    pairs, first, projected = stats
    assert pairs > 4_000_000, "the culled path was not taken"
    dropped = 1.0 - projected / pairs
    print(f"occlusion cull at KITTI scale: {pairs} pairs, {first / pairs:.1%} in the first shell, {dropped:.1%} dropped")
    assert dropped > 0.25, f"only {dropped:.1%} of the (tile, keyframe) pairs were proven hidden"
    for (ap, ao), (bp, bo) in zip(out[1], out[0]):
        assert (ao == bo).all() and (ap.view(np.uint32) == bp.view(np.uint32)).all(), "occlusion-culled reprojection differs from the plain launch"
