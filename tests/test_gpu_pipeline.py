"""End-to-end parity: Removerter::run() Steps 0-3 on the GPU (through the C ABI) vs the CPU oracle, every output
of the file protocol compared.  Labels are implied exact by identical point sets; XYZ bar is 1e-4 m
(north_star) but the comparison below is bitwise."""
import numpy as np
import pytest

from conftest import assert_clouds_equal

pytestmark = pytest.mark.gpu

MAPS = ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_map_static", "central_map_dynamic",
        "query_map_static", "query_map_dynamic", "central_sess_high_dyn", "query_sess_high_dyn", "union_map_queryside",
        "union_map_centralside", "pd_map", "nd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map",
        "updated_map", "updated_map_strong"]
SCANS = ["scans_updated", "scans_updated_strong", "scans_pd", "scans_pd_strong", "scans_nd_strong"]


def _run_gpu(ltm, C, Q, **kw):
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0)
    P = Params(**kw)
    sessions = []
    for name, S in (("Central", C), ("Query", Q)):
        sessions.append(Session(name, ctx.upload_scans(S["scans"], S["offsets"]), ctx.poses(S["poses"], S["inv"])))
    rmv = Removerter(HipOps(ctx), P, sessions[0], sessions[1])
    rmv.run()
    return ctx, rmv


def _compare(rmv, ref):
    for name in MAPS:
        want = ref.cloud(name)
        got = rmv.outputs.get(name)
        if want is None:
            assert got is None, f"{name}: GPU produced a map the oracle did not"
            continue
        assert got is not None, f"{name}: missing on the GPU"
        assert_clouds_equal(got.download(), want, name)
    for name, ss in rmv.scan_outputs().items():
        w_pts, w_off = ref.scanset(name)
        g_pts, g_off = ss.download()
        assert (g_off == w_off).all(), f"{name}: per-keyframe counts differ"
        assert_clouds_equal(g_pts, w_pts, name)


def test_pipeline_single_res_as_shipped(ltm, orc, small_pair):
    C, Q = small_pair
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01), C, Q)
    ctx, rmv = _run_gpu(ltm, C, Q)
    _compare(rmv, ref)
    assert sum(len(rmv.outputs[n]) for n in ("weak_nd_map", "weak_pd_map", "central_map_dynamic")) > 0
    ctx.close()


def test_pipeline_three_res_self_removert(ltm, orc, small_pair):
    C, Q = small_pair
    res = (2.5, 2.0, 1.5)
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01, use_self_removert=True, res_list=res), C, Q)
    ctx, rmv = _run_gpu(ltm, C, Q, gpu_use_self_removert=True, remove_resolution_list=list(res))
    _compare(rmv, ref)
    ctx.close()


def test_pipeline_sessions_side_by_side_on_two_contexts(ltm, orc, small_pair):
    """Removerter(query_side=...): merge + grid and the Step-1 chain of the query session on a second context (own stream, own pool) from a
    second host thread beside the central session's; three times in a row on the same contexts so that recycled pool blocks are exercised."""
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    C, Q = small_pair
    res = (2.5, 2.0, 1.5)
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01, use_self_removert=True, res_list=res), C, Q)
    ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0)
    ctx2 = ltm.Context(vfov=50.0, hfov=360.0, device=0)
    P = Params(gpu_use_self_removert=True, remove_resolution_list=list(res))
    up = lambda c, S: (c.upload_scans(S["scans"], S["offsets"]), c.poses(S["poses"], S["inv"]))    # noqa: E731
    cs, qs, qs2 = up(ctx, C), up(ctx, Q), up(ctx2, Q)
    for _ in range(3):
        ctx.clear_caches(); ctx2.clear_caches()
        rmv = Removerter(HipOps(ctx), P, Session("Central", *cs), Session("Query", *qs), query_side=(HipOps(ctx2), Session("Query", *qs2)))
        rmv.run()
        _compare(rmv, ref)
        del rmv
    del cs, qs, qs2
    ctx2.close()
    ctx.close()


def test_pipeline_other_knn_params_and_extrinsic(ltm, orc, small_pair):
    """code-default kNN parameters (k=3, thr=0.1) and a non-identity LiDAR->base extrinsic"""
    C, Q = small_pair
    l2b = np.eye(4)
    l2b[:3, :3] = [[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]
    l2b[:3, 3] = [0.3, -0.1, 0.25]
    ref = orc.pipeline_run(orc.make_params(k=3, knn_thr=0.1, lidar2base=l2b), C, Q)
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    ctx = ltm.Context(vfov=50.0, hfov=360.0, lidar2base=l2b, device=0)
    P = Params(num_nn_points_within=3, dist_nn_points_within=0.1)
    sessions = [Session(n, ctx.upload_scans(S["scans"], S["offsets"]), ctx.poses(S["poses"], S["inv"])) for n, S in (("Central", C), ("Query", Q))]
    rmv = Removerter(HipOps(ctx), P, *sessions)
    rmv.run()
    _compare(rmv, ref)
    ctx.close()


def test_sharded_ops_world1_rccl_plumbing(ltm, orc, small_pair):
    """dist.ShardedOps with a real RCCL process group of size 1: exercises the uint8 label all-reduce, the device-array views
    and the all-gather reassembly on the GPU; result must equal the plain single-GPU pipeline (= the oracle)."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from ltmapper_amd.dist import ShardedOps
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        C, Q = small_pair
        ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0)
        sessions = [Session(n, ctx.upload_scans(S["scans"], S["offsets"]), ctx.poses(S["poses"], S["inv"])) for n, S in (("Central", C), ("Query", Q))]
        sops = ShardedOps(HipOps(ctx), dist, 0, 1)
        rmv = Removerter(sops, Params(), *sessions)
        rmv.run()
        _compare(rmv, orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01), C, Q))
        # the sharded voxel grid's exchange (bypassed by voxel() itself when world == 1)
        cmap = rmv.outputs["OriginalNoisyCentralMapGlobal"]
        via = sops._allgather_cloud(sops.ops.voxel_shard(cmap, 0.4, 0, 1))
        np.testing.assert_array_equal(via.download(), sops.ops.voxel(cmap, 0.4).download())
        ctx.close()
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, q, C, Q):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from ltmapper_amd.dist import ShardedOps
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    from staged_dist import StagedDist
    ctx = capi.Context(vfov=50.0, hfov=360.0, device=0)
    sessions = [Session(n, ctx.upload_scans(S["scans"], S["offsets"]), ctx.poses(S["poses"], S["inv"])) for n, S in (("Central", C), ("Query", Q))]
    sops = ShardedOps(HipOps(ctx), StagedDist, rank, world)
    sops.VOXEL_SHARD_MIN = 0
    rmv = Removerter(sops, Params(), *sessions)
    rmv.run()
    maps = {k: v.download() for k, v in rmv.outputs.items() if v is not None}
    scans = {k: sops.materialize(v).download() for k, v in rmv.scan_outputs().items()}
    q.put((rank, maps, scans))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_ops_logical_ranks_sharing_the_gpu(orc, small_pair, world):
    """world_size 2 / 4 with the real HIP stages (keyframe shards, label
    union, rank-local scan sets, sharded voxel grids): every rank must end with the single-GPU (= oracle) result.  Collectives are
    staged through the host over gloo because the test box has one GPU."""
    import socket
    import torch.multiprocessing as mp
    C, Q = small_pair
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_two_rank_worker, args=(r, world, port, q, C, Q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    results, deadline = [], time.time() + 400
    while len(results) < world:
        try:
            results.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = orc.pipeline_run(orc.make_params(k=2, knn_thr=0.01), C, Q)
    for rank, maps, scans in results:
        for name in MAPS:
            want = ref.cloud(name)
            if want is None:
                assert name not in maps
                continue
            assert_clouds_equal(maps[name], want, f"rank {rank} {name}")
        for name in SCANS:
            w_pts, w_off = ref.scanset(name)
            g_pts, g_off = scans[name]
            assert (g_off == w_off).all(), f"rank {rank} {name}: per-keyframe counts differ"
            assert_clouds_equal(g_pts, w_pts, f"rank {rank} {name}")


def test_cascade_feeds_updated_scans_forward(ltm, orc):
    """configs[2] in miniature: 01 -> (02, 03); run j+1 must see the scans_updated of run j as the reference's loader would re-read them:
    pcl::VoxelGrid(downsample_voxel_size) per scan (Session.cpp:284-289), then precleaningKeyframes(2.5) (Removerter.cpp:1658-1660)"""
    from ltmapper_amd.cascade import run_cascade
    from ltmapper_amd.removerter import HipOps, Params
    from tools import synth
    S = [synth.to_numpy(synth.make_session(s, 5, "small")) for s in (1, 2, 3)]
    ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0)
    up = lambda T: (ctx.upload_scans(T["scans"], T["offsets"]), ctx.poses(T["poses"], T["inv"]))
    c_scans, c_poses = up(S[0])
    runs = run_cascade(HipOps(ctx), Params(), c_scans, c_poses, [up(S[1]), up(S[2])])
    assert len(runs) == 2
    # oracle: run 1, then run 2 with central := scans_updated of run 1 through VoxelGrid + pre-clean (same central poses)
    ref1 = orc.pipeline_run(orc.make_params(), S[0], S[1])
    upd_pts, upd_off = ref1.scanset("scans_updated")
    re_pts, re_off = [], [0]
    for k in range(len(upd_off) - 1):
        p = orc.preclean(orc.voxel_grid(upd_pts[int(upd_off[k]):int(upd_off[k + 1])], 0.05), 2.5)
        re_pts.append(p); re_off.append(re_off[-1] + len(p))
    assert re_off[-1] < int(upd_off[-1]), "the re-load is expected to thin the scans (otherwise this test cannot see a missing VoxelGrid / pre-clean)"
    C2 = dict(scans=np.concatenate(re_pts), offsets=np.array(re_off, np.uint64), poses=S[0]["poses"], inv=S[0]["inv"])
    ref2 = orc.pipeline_run(orc.make_params(), C2, S[2])
    _compare(runs[0], ref1)
    _compare(runs[1], ref2)
    ctx.close()


def test_reproject_and_vote_across_small_keyframe_batches(ltm, orc, small_pair):
    """max_kf_batch smaller than the keyframe count: images are processed in several batches and stitched"""
    import numpy as np
    from conftest import assert_clouds_equal
    C, _ = small_pair
    cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], np.eye(4)), 0.05)
    ctx = ltm.Context(vfov=50.0, hfov=360.0, device=0, max_kf_batch=4)      # 6 keyframes -> batches of 4 + 2
    g = ctx.reproject(ctx.upload(cmap), ctx.poses(C["poses"], C["inv"]), 3.0)
    g_pts, g_off = g.download()
    o_pts, o_off = orc.reproject(cmap, C["inv"], np.eye(4), 50.0, 360.0, 3.0)
    assert (g_off == o_off).all()
    assert_clouds_equal(g_pts, o_pts, "reprojection stitched over batches")
    # the scan-image cache must not leak between scan sets or survive a clear
    scans = ctx.upload_scans(C["scans"], C["offsets"])
    poses = ctx.poses(C["poses"], C["inv"])
    a = ctx.visibility_partition(ctx.upload(cmap), scans, poses, 2.5, 0.1, 0, want_labels=True)[2]
    b = ctx.visibility_partition(ctx.upload(cmap), scans, poses, 2.5, 0.1, 0, want_labels=True)[2]   # served from the cache
    ctx.clear_caches()
    c3 = ctx.visibility_partition(ctx.upload(cmap), scans, poses, 2.5, 0.1, 0, want_labels=True)[2]
    want = orc.vote_labels(cmap, C["scans"], C["offsets"], C["inv"], np.eye(4), 50.0, 360.0, 2.5, 0.1, 0)
    assert (a == want).all() and (b == want).all() and (c3 == want).all()
    ctx.close()
