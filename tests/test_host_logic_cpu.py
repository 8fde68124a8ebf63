"""CPU tests of the host logic: the Python Removerter orchestration (product code, lt-mapper_amd/removerter.py) driven by
oracle-backed stage ops must reproduce the independent C++ oracle pipeline; and the keyframe-sharded exchange
(lt-mapper_amd/dist.py) over gloo with world_size 2 and 3 must reproduce the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import assert_clouds_equal

MAPS = ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_map_static", "central_map_dynamic", "query_map_static",
        "query_map_dynamic", "central_sess_high_dyn", "query_sess_high_dyn", "union_map_queryside", "union_map_centralside", "pd_map",
        "nd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map", "updated_map", "updated_map_strong"]


def _tiny_pair():
    from tools import synth
    return (synth.to_numpy(synth.make_session(1, 4, "tiny")), synth.to_numpy(synth.make_session(2, 4, "tiny")))


def _run(ops, C, Q, **kw):
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.removerter import Params, Removerter, Session
    from oracle_ops import OPoses, OScans
    sessions = [Session(n, OScans(S["scans"], S["offsets"]), OPoses(S["poses"], S["inv"])) for n, S in (("Central", C), ("Query", Q))]
    rm = Removerter(ops, Params(**kw), *sessions)
    rm.run()
    out = {k: np.asarray(v.download()) for k, v in rm.outputs.items()}
    scans = {k: v.download() for k, v in rm.scan_outputs().items()}
    return out, scans


def _check_against_oracle_pipeline(out, scans, ref):
    for name in MAPS:
        want = ref.cloud(name)
        if want is None:
            assert name not in out
            continue
        assert_clouds_equal(out[name], want, name)
    for name, (pts, off) in scans.items():
        w_pts, w_off = ref.scanset(name)
        assert (np.asarray(off) == w_off).all(), name
        assert_clouds_equal(pts, w_pts, name)


@pytest.mark.parametrize("three_res", [False, True])
def test_python_orchestration_matches_cpp_oracle_pipeline(orc, three_res):
    from oracle_ops import OracleOps
    C, Q = _tiny_pair()
    kw = dict(gpu_use_self_removert=True, remove_resolution_list=[2.5, 2.0, 1.5]) if three_res else {}
    out, scans = _run(OracleOps(), C, Q, **kw)
    ref = orc.pipeline_run(orc.make_params(use_self_removert=three_res, res_list=(2.5, 2.0, 1.5) if three_res else (2.5,)), C, Q)
    _check_against_oracle_pipeline(out, scans, ref)


@pytest.mark.parametrize("three_res", [False, True])
def test_sessions_side_by_side_on_two_op_sets_equal_the_plain_run(orc, three_res):
    """Removerter(query_side=...): the merge + grid and the Step-1 chain of the query session run from a second thread on a second set of
    stage ops (a second device context on the GPU); every output must be what the one-after-the-other run produces."""
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.removerter import Params, Removerter, Session
    from oracle_ops import OPoses, OracleOps, OScans
    C, Q = _tiny_pair()
    kw = dict(gpu_use_self_removert=True, remove_resolution_list=[2.5, 2.0, 1.5]) if three_res else {}
    want_out, want_scans = _run(OracleOps(), C, Q, **kw)
    mk = lambda n, S: Session(n, OScans(S["scans"], S["offsets"]), OPoses(S["poses"], S["inv"]))   # noqa: E731
    rm = Removerter(OracleOps(), Params(**kw), mk("Central", C), mk("Query", Q), query_side=(OracleOps(), mk("Query", Q)))
    rm.run()
    out = {k: np.asarray(v.download()) for k, v in rm.outputs.items()}
    assert sorted(out) == sorted(want_out)
    for k in want_out:
        assert_clouds_equal(out[k], want_out[k], k)
    for k, v in rm.scan_outputs().items():
        pts, off = v.download()
        assert (np.asarray(off) == want_scans[k][1]).all(), k
        assert_clouds_equal(pts, want_scans[k][0], k)


def _worker(rank, world, port, q, session_groups=True, three_res=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.dist import ShardedOps
    from oracle_ops import OracleOps
    C, Q = _tiny_pair()
    sops = ShardedOps(OracleOps(), dist, rank, world)
    sops.VOXEL_SHARD_MIN = 0          # the tiny maps of this test would otherwise stay replicated
    sops.SESSION_GROUPS = session_groups
    out, scans = _run(sops, C, Q, **(dict(gpu_use_self_removert=True, remove_resolution_list=[2.5, 2.0, 1.5]) if three_res else {}))
    assert (sops.session_groups() is not None) == (session_groups and world % 2 == 0)
    q.put((rank, out, {k: (np.asarray(p), np.asarray(o)) for k, (p, o) in scans.items()}))
    dist.barrier()
    dist.destroy_process_group()


# 4 keyframes per session: equal blocks (2 + 2) and unequal ones (2 + 1 + 1).  An even world splits into a central and a query rank group for
# Step 1 (ShardedOps.session_groups: world 2 = two groups of one rank, world 4 = two groups of two); (2, False) keeps the unsplit path covered
@pytest.mark.parametrize("world,session_groups,three_res", [(2, True, False), (3, True, False), (4, True, False), (4, True, True), (2, False, False),
                                                            (6, False, False), (6, True, True)])     # more ranks than keyframes: empty blocks
def test_keyframe_sharding_over_gloo_matches_single_process(orc, world, session_groups, three_res):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, session_groups, three_res)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    results, deadline = [], time.time() + 300
    while len(results) < world:
        try:
            results.append(q.get(timeout=2))
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    C, Q = _tiny_pair()
    ref = orc.pipeline_run(orc.make_params(use_self_removert=three_res, res_list=(2.5, 2.0, 1.5) if three_res else (2.5,)), C, Q)
    for rank, out, scans in results:       # every rank must hold the full, identical result
        _check_against_oracle_pipeline(out, scans, ref)


def test_shard_ranges_cover_and_are_disjoint():
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.dist import shard_range
    for n in (0, 1, 7, 500, 2000):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_bench_launches_itself_under_torchrun_for_several_gpus():
    """`python bench.py --gpus N` as the driver runs it (no launcher environment) must become N ranks under torch.distributed.run;
    without GPUs every rank then stops at the loud "needs a GPU" assertion -- which shows the re-launch happened with WORLD_SIZE = N"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""      # also on a GPU box: the point is the launch, not the run
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert (r.stdout + r.stderr).count("AssertionError: bench.py needs a GPU") == 2, (r.stdout + r.stderr)[-1500:]


def test_bench_roofline_helpers_on_a_synthetic_profile():
    """bench.py's traffic grouping (PMC kernel names -> profile classes) and the parity-record check, without a GPU"""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    kib = 1024.0
    pmc = {"all_kernels": {
        "void ltm::k_vote_map_cull<true, true>": {"FETCH_SIZE": {"sum": 1000 * kib, "dispatches": 21}, "WRITE_SIZE": {"sum": 100 * kib, "dispatches": 21}},
        "rocprim::radix_sort_onesweep<u64>": {"FETCH_SIZE": {"sum": 300 * kib, "dispatches": 64}, "WRITE_SIZE": {"sum": 300 * kib, "dispatches": 64}},
        "ltm::k_voxel_centroids_packed": {"FETCH_SIZE": {"sum": 100 * kib, "dispatches": 64}, "WRITE_SIZE": {"sum": 10 * kib, "dispatches": 64}},
        "void ltm::k_knn_query_scans<true, 2>": {"FETCH_SIZE": {"sum": 50 * kib, "dispatches": 5}, "WRITE_SIZE": {"sum": 5 * kib, "dispatches": 5}},
        "ltm::k_compare_flag": {"FETCH_SIZE": {"sum": 7 * kib, "dispatches": 24}, "WRITE_SIZE": {"sum": 1 * kib, "dispatches": 24}},
        "ltm::k_selfcheck": {"FETCH_SIZE": {"sum": 1e9, "dispatches": 1}, "WRITE_SIZE": {"sum": 0, "dispatches": 1}}}}
    prof = {"vote_map_cull": dict(ms=70.0, launches=21, units=1e9, bytes=2.0 * 1024 * 1024 * 1024), "voxel": dict(ms=20.0, launches=30, units=1e8, bytes=1e9),
            "knn_query": dict(ms=10.0, launches=5, units=1e8, bytes=1e8), "vote_compare": dict(ms=2.0, launches=24, units=1e8, bytes=1e7)}
    groups = {g["group"]: g for g in bench.traffic_groups(pmc, prof, steps=1)}
    g = groups["vote_map_cull"]
    assert g["traffic_bytes_per_step"] == (2 * 1000 + 100) * kib * 1024 and g["kernels_matched"] == 1
    assert abs(g["traffic_over_algorithmic"] - (2100 * kib * 1024) / (2.0 * 1024 ** 3)) < 1e-3
    sort = next(v for k, v in groups.items() if k.startswith("sort_based"))
    assert sort["kernels_matched"] == 2 and sort["traffic_bytes_per_step"] == (2 * 400 + 310) * kib * 1024
    rest = next(v for k, v in groups.items() if k.startswith("streaming rest"))
    assert rest["kernels_matched"] == 1, "the create-time self-check is not part of a step"
    assert bench.traffic_groups({}, prof, 1) == []
    st = bench.parity_fullsize_status()
    assert set(st) >= {"record", "matches_sources"}


def _oracle_cascade_chain(orc, S):
    """the reference's lifelong loop on the oracle: run j+1's central session = run j's scans_updated re-loaded (VoxelGrid + pre-clean 2.5)"""
    refs, central = [], S[0]
    for j in range(1, len(S)):
        ref = orc.pipeline_run(orc.make_params(), central, S[j])
        refs.append(ref)
        pts, off = ref.scanset("scans_updated")
        re_pts, re_off = [], [0]
        for k in range(len(off) - 1):
            p = orc.preclean(orc.voxel_grid(pts[int(off[k]):int(off[k + 1])], 0.05), 2.5)
            re_pts.append(p); re_off.append(re_off[-1] + len(p))
        central = dict(scans=np.concatenate(re_pts), offsets=np.array(re_off, np.uint64), poses=S[0]["poses"], inv=S[0]["inv"])
    return refs


def _tiny_sessions(n):
    from tools import synth
    return [synth.to_numpy(synth.make_session(s, 4, "tiny")) for s in range(1, n + 1)]


def _cascade_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.cascade import run_cascade
    from ltmapper_amd.dist import ShardedOps
    from ltmapper_amd.removerter import Params
    from oracle_ops import OPoses, OracleOps, OScans
    S = _tiny_sessions(3)
    sops = ShardedOps(OracleOps(), dist, rank, world)
    sops.VOXEL_SHARD_MIN = 0
    P = Params(gather_scan_outputs=True)
    up = [(OScans(T["scans"], T["offsets"]), OPoses(T["poses"], T["inv"])) for T in S]
    runs = run_cascade(sops, P, up[0][0], up[0][1], up[1:])
    res = []
    for rm in runs:
        res.append(({k: np.asarray(v.download()) for k, v in rm.outputs.items()},
                    {k: tuple(np.asarray(x) for x in v.download()) for k, v in rm.scan_outputs().items()}))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_cascade_over_gloo_hands_over_the_reloaded_scans(orc):
    """configs[2] in miniature on two ranks: the keyframe-sharded cascade (rank-local scans_updated go through the loader's VoxelGrid and
    the pre-clean on their own rank) must equal the oracle chain on every rank, both pair runs"""
    import queue
    import time
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cascade_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results, deadline = [], time.time() + 300
    while len(results) < world:
        try:
            results.append(q.get(timeout=2))
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail("a rank died or timed out: exit codes %s" % [p.exitcode for p in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refs = _oracle_cascade_chain(orc, _tiny_sessions(3))
    for rank, res in results:
        assert len(res) == 2
        for (out, scans), ref in zip(res, refs):
            _check_against_oracle_pipeline(out, scans, ref)


def test_comm_meter_counts_what_the_sharded_pipeline_would_exchange(orc):
    """dist.CommMeter on a single process: one label all-reduce of M bytes per vote pass, a key-range exchange (one all-to-all of the points + the
    all-gather of the centroid list) for every merge + grid of rank-local scans (round 4: no scan-set all-gather any more), none for the replicated
    session scans; the results are untouched; scaling_model turns the totals into step times"""
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.dist import CommMeter, scaling_model
    from oracle_ops import OracleOps
    C, Q = _tiny_pair()
    meter = CommMeter(OracleOps())
    out, scans = _run(meter, C, Q)
    ref = orc.pipeline_run(orc.make_params(), C, Q)
    _check_against_oracle_pipeline(out, scans, ref)
    ev = meter.events
    assert ev["label_allreduce"][0] == 2 + 3 + 3, "single-res: removeOnce x 2 sessions, 3 ND and 3 PD filter passes"
    assert ev["label_allreduce"][1] > 0
    assert ev["scans_allgather"][0] == 0, "merges of rank-local scans go through the key-range exchange"
    assert ev["points_alltoall"][0] == 2 + 2 + 4, "HD dynamic scans x 2, ND / PD diff scans, and the four merges of the debug maps"
    assert ev["voxel_allgather"][0] == ev["points_alltoall"][0], "one centroid all-gather per exchange (tiny replicated maps stay replicated)"
    assert ev["points_alltoall"][1] > 0 and meter.sharded_voxel_points * 16 == ev["points_alltoall"][1]
    m = scaling_model({"vote_map_cull": 70.0, "voxel": 20.0, "knn_query": 10.0, "merge": 4.0}, 110.0, {k: (v[0], v[1]) for k, v in ev.items()}, sharded_voxel_fraction=0.5)
    assert m["sharded_ms"] == 80.0 + 10.0 + 4.0 and m["replicated_ms"] == 16.0
    assert m["ranks"]["8"]["step_ms"] < m["ranks"]["2"]["step_ms"] < 110.0
    assert m["ranks"]["8"]["bytes_received_per_rank_per_step"] < m["ranks"]["2"]["bytes_received_per_rank_per_step"]


def test_scaling_model_prices_the_session_rank_groups():
    """scaling_model(step1=...): for an even number of ranks makeGlobalMap + Step 1 run on two rank groups -- a rank does the replicated Step-1 work
    of one session instead of two and exchanges its session's collectives among half the ranks, then swaps the finished maps with its partner"""
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.dist import scaling_model
    cls = {"vote_map_cull": 80.0, "voxel": 20.0, "partition": 4.0, "knn_query": 10.0}
    ev = {"label_allreduce": (24.0, 24 * 7e6), "voxel_allgather": (10.0, 3e8), "points_alltoall": (8.0, 1.6e9), "scans_allgather": (0.0, 0.0)}
    step1 = {"class_ms": {"vote_map_cull": 60.0, "voxel": 12.0, "partition": 3.0}, "wall_ms": 80.0, "events": {"label_allreduce": (18.0, 18 * 7e6), "voxel_allgather": (4.0, 2e8),
             "points_alltoall": (2.0, 1e8), "scans_allgather": (0.0, 0.0)}, "sharded_voxel_fraction": 0.25, "swap_bytes": 3.3e8}
    plain = scaling_model(cls, 120.0, ev, ranks=(2, 3, 4, 8), sharded_voxel_fraction=0.5)
    m = scaling_model(cls, 120.0, ev, ranks=(2, 3, 4, 8), sharded_voxel_fraction=0.5, step1=step1)
    assert m["ranks"]["3"] == plain["ranks"]["3"], "an odd world does not split"
    g = m["session_groups"]
    assert g["step1_sharded_ms"] == 60.0 + 0.25 * 12.0 and g["step1_replicated_ms"] == 80.0 - 63.0
    for n in ("2", "4", "8"):
        assert m["ranks"][n]["without_session_groups"] == plain["ranks"][n]
        assert m["ranks"][n]["step_ms"] < plain["ranks"][n]["step_ms"], "half of 17 ms of replicated Step-1 work outweighs a 2 ms swap"
    # two ranks: each group is one rank -- no Step-1 collective at all, only the swap
    r2 = m["ranks"]["2"]
    rest_comm = r2["comm_ms"] - (1e3 * 3.3e8 / 150e9 + 0.05)
    assert rest_comm > 0 and rest_comm < plain["ranks"]["2"]["comm_ms"]


def test_cascade_with_deferred_hand_over_equals_the_serial_cascade(orc):
    """cascade.run_cascade(overlap=True) starts the re-load of scans_updated when a run ends and finishes it inside the next run, after that run's query
    session has been issued (Removerter._queryThenCentral: query merge + grid + Step 1 first, then the central session's).  The order of the two
    sessions' chains must not change a single output: three sessions, both pair runs, every map and scan set against the serial form."""
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd.cascade import run_cascade
    from ltmapper_amd.removerter import Params
    from oracle_ops import OPoses, OracleOps, OScans
    S = _tiny_sessions(3)
    out = {}
    for overlap in (True, False):
        up = [(OScans(T["scans"], T["offsets"]), OPoses(T["poses"], T["inv"])) for T in S]
        ops = OracleOps()
        runs = run_cascade(ops, Params(gather_scan_outputs=True, gpu_use_self_removert=True, remove_resolution_list=[2.5, 2.0, 1.5]), up[0][0], up[0][1], up[1:], overlap=overlap, prepare_next=True)
        assert len(runs) == 2
        if overlap:
            assert runs[0].central_scans_future is None and runs[1].central_scans_future is None, "the deferred scans were collected"
        assert len(runs[-1].next_central_scans.download()[0]) > 0, "prepare_next: the hand-over after the last run is finished, not dropped"
        out[overlap] = [({k: np.asarray(v.download()) for k, v in rm.outputs.items()},
                         {k: tuple(np.asarray(x) for x in v.download()) for k, v in rm.scan_outputs().items()}) for rm in runs]
    for (ma, sa), (mb, sb) in zip(out[True], out[False]):
        assert sorted(ma) == sorted(mb) and sorted(sa) == sorted(sb)
        for k in ma:
            assert ma[k].shape == mb[k].shape and (ma[k].view(np.uint32) == mb[k].view(np.uint32)).all(), k
        for k in sa:
            assert (sa[k][1] == sb[k][1]).all() and (sa[k][0].view(np.uint32) == sb[k][0].view(np.uint32)).all(), k
    assert len(out[True][1][0]["updated_map"]) > 100
