#!/usr/bin/env python3
"""bench.py -- keyframe-pairs/sec of the LT-removert / LT-map hot path (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU over RCCL); when
the driver already launched it that way the environment is used as it is.

A *step* is one full pass of the hot path over one synthetic session pair that is already resident in HBM:
makeGlobalMap + Removerter::run() Steps 1-3 (Removerter.cpp:1653-1678) = remove/revert visibility votes, static
reprojection, inter-session kNN change detection, ND/PD filtering, LT-map composition and the final reprojections.
Default workload = BASELINE.json configs[1]: "ParkingLot 01 vs 02, 500 keyframes each, 3-res removert, 1xMI355X"
restated on the synthetic `lot` scene (tools/synth.py; no dataset is reachable offline).  `lot-cascade-6x500` is
configs[2]: five chained pair runs 01 -> 02..06 (lt-mapper_amd/cascade.py), 2500 keyframe pairs per step.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     -- the dominant kernel (k_vote_map_cull).  Its binding resource is VALU ISSUE, not HBM (the ~110 MB map stays in
                  the 256 MiB Infinity Cache and a tile is re-used from L2 by eight keyframes): `bound` = "valu", `achieved` /
                  `peak` / `frac` = VALU lane-instructions per second against the vector pipes' nominal peak (SQ_INSTS_VALU from
                  a separate rocprofv3 --pmc pass, profiles/pmc_latest.json, used only if it was collected for exactly this kernel
                  source; average launch duration measured live with HIP events on the context's stream).  Beside it:
                  `hbm_algorithmic` (SURVEY 8d algorithmic bytes per launch / launch duration against 8 TB/s -- can exceed 1,
                  it is cache-served) and `hbm_measured_frac` (`traffic` = PMC HBM bytes per launch, against 8 TB/s);
  rooflines    -- per kernel class of the step: algorithmic GB/s against the HBM peak, and (from the same PMC passes, which keep
                  every kernel incl. rocPRIM's) the measured HBM traffic of the class's kernel group and traffic / algorithmic;
  t_total      -- files -> files through the C++ host `ltm_run` (SURVEY 8d T_total) for configs[1] and configs[0], measured now;
  parity_fullsize -- whether the committed full-size (2 x 500 keyframes, 3-res) bitwise comparison with the oracle was made with
                  exactly the sources this run measures;
  cpu_baseline -- the CPU oracle (a port of the reference; the reference itself cannot be built here) timed on this
                  box on a bounded keyframe sample of the same workload: single thread (the north-star denominator) and
                  all cores; the committed full, unsampled runs are quoted next to it.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
# gfx950 issues one wave64 fp32 / integer VALU instruction per SIMD every 2 cycles (32 lanes per clock; packed fp32 and fp64 at half,
# transcendentals at about a quarter of that): measured with tools/ubench/valu_rate.hip -- 0.9-1.14e12 wave-instructions/s sustained
# over the chip for v_fma_f32 / v_mul_f32 / v_and_b32 against 1.23e12 at the nominal 2.4 GHz (the clock drops under a pure VALU load)
N_CU, SIMD_PER_CU, LANES_PER_SIMD = 256, 4, 32
VALU_MEASURED_CEILING_WAVE_INSTS = 1.1e12

DEFAULT_WORKLOAD = "lot-2x500-os1-64-3res"
WORKLOADS = {
    # name: (sensor, keyframes per session, 3-res?, scene, kf spacing [m], voxel [m], kNN k, kNN thr)
    "lot-2x500-os1-64-3res": ("os1-64", 500, True, "lot", 1.0, 0.05, 2, 0.01),        # BASELINE configs[1] (default)
    "lot-2x50-os1-64-1res": ("os1-64", 50, False, "lot", 1.0, 0.05, 2, 0.01),         # configs[0] shape (plumbing case)
    "lot-2x100-small-3res": ("small", 100, True, "lot", 1.0, 0.05, 2, 0.01),          # quick check
    "lot-cascade-6x500": ("os1-64", 500, True, "lot", 1.0, 0.05, 2, 0.01),            # configs[2]: sessions 01 -> 02..06 chained
    "lot-cascade-4x50-small": ("small", 50, True, "lot", 1.0, 0.05, 2, 0.01),         # quick check of the cascade workload
    "street-2x2000-hdl64e-1res": ("hdl-64e", 2000, False, "street", 1.0, 0.05, 2, 0.01),   # configs[3], KITTI-scale
    "street-2x2000-hdl64e-3res": ("hdl-64e", 2000, True, "street", 1.0, 0.05, 2, 0.01),
    "street-2x200-mls-knn": ("mls", 200, False, "street", 2.0, 0.1, 2, 0.04),         # configs[4], dense MLS kNN stress
}
CASCADE_SESSIONS = {"lot-cascade-6x500": 6, "lot-cascade-4x50-small": 4}

VALU_CLASSES = ("vote_map_cull", "vote_map_exact", "reproject_map")      # projection kernels: bound by VALU issue, not HBM
DOMINANT_KERNEL = {"vote_map_cull": "k_vote_map_cull", "vote_map_exact": "k_map_rimg_blockmin", "reproject_map": "k_map_rimg_blockmin", "knn_query": "k_knn_fast",
                   "knn_query_p2": "k_knn_slow_sorted", "voxel": "rocprim onesweep + k_voxel_centroids_packed", "merge": "k_transform_scans"}

# kernel class (ltm_profile_read) -> kernels behind it; bytes are the algorithmic bytes of DESIGN.md section 4
CLASS_KERNELS = {
    "vote_map_cull": "k_vote_map_cull", "vote_map_exact": "k_map_rimg_blockmin", "reproject_map": "k_map_rimg_blockmin",
    "vote_scan": "k_scan_rimg + k_image_max", "vote_compare": "k_compare_flag", "vote_fill": "k_fill_u64",
    "partition": "rocprim scan + k_partition_scatter", "voxel": "bbox + Morton keys + rocprim radix sort + k_voxel_centroids",
    "voxel_scanset": "per-keyframe bbox + composite keys + rocprim radix sort + k_voxel_centroids",
    "knn_build": "k_cell_keys + rocprim radix sort + k_hash_build + k_knn_bucket_build + k_knn_bitmap_build",
    "knn_query": "k_knn_fast (phase 1: 64-byte cell buckets + occupancy bitmap) / k_knn_query_cloud", "knn_query_p2": "rocprim scan + k_knn_queue_scatter + k_knn_slow (phase 2: exact search of the undecided queries)",
    "reproject_gather": "rocprim scan + k_reproject_gather", "merge": "k_transform_scans",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-stride", type=int, default=50, help="cpu_baseline, single thread: visit every s-th keyframe in per-keyframe loops (50 = 10 of 500 "
                                                               "keyframes per session: ~50 s on the GPU box's host; round 4 ran stride 10 = 165 s of a 187 s driver run)")
    ap.add_argument("--cpu-stride-allcore", type=int, default=10, help="cpu_baseline, all cores: keyframe stride")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-now", action="store_true", help="non-default workloads: time the sampled CPU leg in this run (minutes to hours on the big "
                                                                    "configurations) instead of quoting profiles/cpu_baseline_<workload>.json (tools/cpu_baseline_sampled.py)")
    ap.add_argument("--no-t-total", action="store_true", help="skip the files -> files measurement through ltm_run (default workload, one GPU)")
    ap.add_argument("--cpu-allcore", action="store_true", help="cpu_baseline: also time the oracle on all host cores now (every keyframe when the box has "
                                                               "many cores: ~3 min on 256); without it the committed measurement is quoted")
    ap.add_argument("--no-cascade-overlap", action="store_true", help="cascade workloads: finish every hand-over (scans_updated re-gridded in PCL's order: keys to host threads and back) "
                                                                      "before the next pair run starts, as in round 4, instead of beside the next run's query session (A/B)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--extra-out", default=None, help="sidecar file for everything the printed line does not carry (default profiles/bench_extra_latest.json)")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2), help="one GPU: 2 (default) = the independent chains of run() side by side on the context and its lane "
                    "(two host threads, include/ltm.h 'lanes'; removerter.Removerter.run_two_lanes); 1 = the one-lane order of rounds 1-5.  With 2 lanes the kernel-class "
                    "times / rooflines come from a separate ONE-lane profiling pass after the timed region (overlapping launches inflate each other's event times)")
    ap.add_argument("--profile-steps", type=int, default=2, help="steps of the one-lane profiling pass behind a two-lane timed region")
    ap.add_argument("--overlap-sessions", action="store_true", help="EXPERIMENT (one GPU, pair workloads): merge + grid and Step 1 of the query session on a second "
                    "context / stream / host thread beside the central session's (removerter.Removerter query_side).  Kernel times of overlapping launches "
                    "are inflated by each other, so the roofline figures of such a line describe the overlap, not the kernels")
    return ap.parse_args()


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same args>`"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback for the measured path"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from ltmapper_amd.cascade import run_cascade
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    from tools import synth

    sensor, n_kf, three_res, scene, spacing, voxel, knn_k, knn_thr = WORKLOADS[args.workload]
    n_sessions = CASCADE_SESSIONS.get(args.workload, 2)
    dev = f"cuda:{local_rank}"
    t0 = time.perf_counter()
    # synthetic sessions 01, 02, ..., generated on the GPU, already in HBM when the timed region starts
    sess_t = [synth.make_session(s, n_kf, sensor, device=dev, scene=scene, kf_spacing=spacing) for s in range(1, n_sessions + 1)]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0

    ctx = capi.Context(vfov=50.0, hfov=360.0, device=local_rank)
    P = Params(gpu_use_self_removert=three_res, remove_resolution_list=[2.5, 2.0, 1.5] if three_res else [2.5],
               num_nn_points_within=knn_k, dist_nn_points_within=knn_thr, downsample_voxel_size=voxel)

    def load(S):
        scans = ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64))
        return ctx.preclean(scans, 2.5), ctx.poses(S["poses"], S["inv"])     # precleaningKeyframes(2.5), Removerter.cpp:1660

    from ltmapper_amd.dist import CommMeter, ShardedOps, scaling_model
    meter = None
    if world > 1:
        ops = ShardedOps(HipOps(ctx), dist, rank, world)
    else:
        ops = meter = CommMeter(HipOps(ctx))      # single GPU: note what the sharded pipeline would exchange (a few dictionary updates per stage)

    loaded = [load(S) for S in sess_t]   # loading + pre-clean are Step 0 plumbing, outside the timed region
    two_lanes = args.lanes == 2 and world == 1 and not args.overlap_sessions
    plain_ops = HipOps(ctx)
    lane_ops = plain_ops.lane() if two_lanes else None
    snap = {"on": False, "prof": {}, "events": {}, "svp": 0, "step1_wall": 0.0, "step1_ms": {}, "step1_voxel_units": 0.0, "step1_events": {}, "step1_svp": 0, "swap_bytes": 0.0}
    ctx2 = q2 = None
    if args.overlap_sessions:
        assert world == 1 and n_sessions == 2, "--overlap-sessions: one GPU, one session pair"
        ctx2 = capi.Context(vfov=50.0, hfov=360.0, device=local_rank)
        S = sess_t[1]
        q2 = (ctx2.preclean(ctx2.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5), ctx2.poses(S["poses"], S["inv"]))

    def one_step(lanes=False):
        ctx.clear_caches()   # no derived data (scan range images) survives from a previous step: every step is a fresh run
        if lanes:
            lane_ops.ctx.clear_caches()
        if n_sessions > 2:
            runs = run_cascade(plain_ops if lanes else ops, P, loaded[0][0], loaded[0][1], loaded[1:], overlap=not args.no_cascade_overlap, lane_ops=lane_ops if lanes else None)
            return runs[-1]
        (cs, cp), (qs, qp) = loaded
        side = None
        if ctx2 is not None:
            ctx2.clear_caches()
            side = (HipOps(ctx2), Session("Query", q2[0], q2[1]))
        rm = Removerter(plain_ops if lanes else ops, P, Session("Central", cs, cp), Session("Query", qs, qp), query_side=side, lane_ops=lane_ops if lanes else None)
        if snap["on"]:
            # makeGlobalMap + Step 1 is the part of the step that an even number of ranks runs on two rank groups (ShardedOps.session_groups): its
            # kernel classes, wall time and would-be collectives are snapshot at its end so that the scaling model can price that split
            t_start, base_prof, base_ev, base_svp = time.perf_counter(), snap["prof"], snap["events"], snap["svp"]

            def on_stage(name):
                if name != ("remove_high_dynamic" if P.gpu_skip_hd_knn else "hd_knn"):
                    return
                snap["step1_wall"] += time.perf_counter() - t_start
                now = ctx.profile_read()
                for k, v in now.items():
                    snap["step1_ms"][k] = snap["step1_ms"].get(k, 0.0) + v["ms"] - base_prof.get(k, {}).get("ms", 0.0)
                snap["step1_voxel_units"] += now.get("voxel", {}).get("units", 0.0) - base_prof.get("voxel", {}).get("units", 0.0)
                for k, v in meter.events.items():
                    e = snap["step1_events"].setdefault(k, [0, 0])
                    e[0] += v[0] - base_ev.get(k, (0, 0))[0]
                    e[1] += v[1] - base_ev.get(k, (0, 0))[1]
                snap["step1_svp"] += meter.sharded_voxel_points - base_svp
            rm.on_stage = on_stage
        rm.run()
        if snap["on"]:
            snap["prof"], snap["events"], snap["svp"] = ctx.profile_read(), {k: tuple(v) for k, v in meter.events.items()}, meter.sharded_voxel_points
            names = ("OriginalNoisy%sMapGlobal", "%s_map_static", "%s_map_dynamic", "%s_sess_high_dyn")
            snap["swap_bytes"] += 16 * sum(len(rm.outputs[n % (sname if "Noisy" in n else sname.lower())]) for sname in ("Central", "Query") for n in names
                                           if rm.outputs.get(n % (sname if "Noisy" in n else sname.lower())) is not None) / 2.0
        return rm

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    # the outputs of the previous step are released BEFORE the next one starts (as a host that runs pair after pair would): every
    # timed step then finds its blocks in the context's pool and the timed region contains no first use of fresh device memory
    last = None
    for _ in range(args.warmup):
        last = None
        last = one_step(two_lanes)
    if two_lanes and args.warmup:      # the one-lane profiling pass behind the timed region needs its (larger) blocks in the main context's pool too
        last = None
        last = one_step(False)
    barrier()
    ctx.profile_reset()
    ctx.profile_enable(not two_lanes)
    if ctx2 is not None:
        ctx2.synchronize()
        ctx2.profile_reset()
        ctx2.profile_enable(True)
    ctx.voxel_stats(reset=True)
    if meter:
        meter.reset()
    snap["on"] = bool(meter) and n_sessions == 2 and not args.overlap_sessions and not two_lanes
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = None
        last = one_step(two_lanes)
    barrier()
    if two_lanes:
        lane_ops.ctx.synchronize()
    elapsed = time.perf_counter() - t0
    psteps = args.steps          # the steps the kernel-class profile covers
    one_lane_ms = None
    timed_stage_ms = {k: round(1e3 * v, 2) for k, v in last.timings.items()}
    if two_lanes:
        # kernel classes, rooflines, comm meter: a ONE-lane pass with the HIP-event profile on (events of overlapping launches would count the other lane's work)
        psteps = max(args.profile_steps, 1)
        last = None
        last = one_step(False)      # one untimed step in the one-lane order first: the pools re-balance (blocks that lived in the lane's pool are allocated afresh on the main one)
        barrier()
        ctx.profile_reset()
        ctx.profile_enable(True)
        ctx.voxel_stats(reset=True)
        ctx.cull_stats()
        meter.reset()
        snap["on"] = bool(meter) and n_sessions == 2
        tp = time.perf_counter()
        for _ in range(psteps):
            last = None
            last = one_step(False)
        barrier()
        one_lane_ms = 1e3 * (time.perf_counter() - tp) / psteps
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    if ctx2 is not None:       # the second context's launches belong to the same step
        ctx2.profile_enable(False)
        for k, v in ctx2.profile_read().items():
            if k in prof:
                for f in ("ms", "launches", "units", "bytes"):
                    prof[k][f] += v[f]
            else:
                prof[k] = dict(v)
    cull_surv, cull_pts = ctx.cull_stats()
    vox_grids, vox_identity = ctx.voxel_stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs_per_step = n_kf * (n_sessions - 1)   # min(N_c, N_q) keyframe pairs per session pair, one pair run per query session
    value = pairs_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    clock_hz = 1e3 * float(getattr(torch.cuda.get_device_properties(local_rank), "clock_rate", 2.4e6))
    valu_peak = N_CU * SIMD_PER_CU * LANES_PER_SIMD * clock_hz      # VALU lane-instructions per second
    pmc = load_pmc(args.workload)

    groups = traffic_groups(pmc, prof, psteps)

    steps = max(psteps, 1)
    step_kernel_ms = sum(v["ms"] for v in prof.values()) / steps

    def class_roofline(cls, v):
        """one kernel class of the step against the resource that bounds it.  HBM-bound classes: `frac` = compulsory bytes / class time / 8 TB/s (SURVEY 8d's
        `achieved`); VALU-bound classes (the projection kernels): `frac` = SQ_INSTS_VALU x 64 lanes / class time / nominal VALU lane rate.  Beside it, always:
        hbm_algorithmic_frac (SURVEY 8d's formula: one map read per keyframe -- may exceed 1 for the projection kernels, which read the map once per launch),
        hbm_compulsory_frac (bytes the launch must move as designed), hbm_measured_frac and traffic_over_compulsory from the PMC passes (null without them)"""
        if not v["launches"] or v["ms"] <= 0:
            return None
        ms = v["ms"] / steps
        sec = ms * 1e-3
        alg, comp = v["bytes"] / steps, v.get("bytes_c", v["bytes"]) / steps
        bound = "valu" if cls in VALU_CLASSES else "hbm"
        r = {"class": cls, "kernels": CLASS_KERNELS.get(cls, cls), "bound": bound, "ms_per_step": round(ms, 3), "launches_per_step": round(v["launches"] / steps, 2),
             "units_per_step": round(v["units"] / steps, 1), "algorithmic_bytes_per_step": round(alg, 1), "compulsory_bytes_per_step": round(comp, 1),
             "hbm_algorithmic_frac": round(alg / sec / (HBM_PEAK_GBS * 1e9), 4), "hbm_compulsory_frac": round(comp / sec / (HBM_PEAK_GBS * 1e9), 4),
             "valu_frac": None, "traffic": None, "traffic_scope": None, "hbm_measured_frac": None, "traffic_over_compulsory": None}
        share = 1.0
        if cls in SHARED_KERNEL_CLASSES:
            tot = sum(prof[c]["ms"] for c in SHARED_KERNEL_CLASSES if c in prof)
            share = v["ms"] / tot if tot else 1.0
        own, names = class_own_counter(pmc, cls, "traffic", share)
        if own is not None and cls not in CLASSES_WITH_LIBRARY_KERNELS:
            r.update(traffic=round(own, 1), traffic_scope="class", class_kernels_counted=names)
        else:     # rocPRIM's scans / sorts serve several classes: the group's traffic against the group's compulsory bytes and time
            g = next((g for g in groups if cls in g["classes"]), None)
            if g:
                r.update(traffic=g["traffic_bytes_per_step"], traffic_scope="group: " + g["group"], hbm_measured_frac=g["hbm_measured_frac"],
                         traffic_over_compulsory=g["traffic_over_compulsory"])
                if own is not None:
                    r.update(class_own_kernels_traffic=round(own, 1), library_kernels_not_attributable=CLASSES_WITH_LIBRARY_KERNELS.get(cls))
        if r["traffic_scope"] == "class":
            r.update(hbm_measured_frac=round(own / sec / (HBM_PEAK_GBS * 1e9), 4), traffic_over_compulsory=round(own / comp, 3) if comp else None)
        wave_insts, _ = class_own_counter(pmc, cls, "valu", share)
        if wave_insts is not None:
            r["valu_frac"] = round(wave_insts * 64.0 / sec / valu_peak, 4)
            if v["units"]:
                r["valu_insts_per_unit"] = round(wave_insts * 64.0 / (v["units"] / steps), 2)
        r["frac"] = r["valu_frac"] if bound == "valu" else r["hbm_compulsory_frac"]
        return r

    rooflines = [r for r in (class_roofline(k, v) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])) if r]
    # the dominant kernel of THIS workload's step (configs[1]: k_vote_map_cull; the street's single-res step: k_map_rimg_blockmin; ...)
    roofline = None
    if rooflines:
        d = rooflines[0]
        v = prof[d["class"]]
        avg_ms = v["ms"] / v["launches"]
        per_launch = 1.0 / (v["launches"] / steps)
        if d["bound"] == "valu":
            lane = d["valu_frac"] * valu_peak if d["valu_frac"] is not None else None
            ach, peak, unit = (round(lane / 1e12, 3) if lane else None), round(valu_peak / 1e12, 3), "T VALU lane-inst/s"
        else:
            ach, peak, unit = round(d["compulsory_bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9, 1), HBM_PEAK_GBS, "GB/s"
        roofline = {"kernel": DOMINANT_KERNEL.get(d["class"], d["kernels"]), "class": d["class"], "bound": d["bound"], "achieved": ach, "peak": peak, "unit": unit, "frac": d["frac"],
                    "traffic": round(d["traffic"] * per_launch, 1) if d["traffic"] is not None and d["traffic_scope"] == "class" else None,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": round(v["launches"] / steps, 2),
                    "share_of_step": round(d["ms_per_step"] / step_kernel_ms, 3) if step_kernel_ms else None,
                    "valu_frac": d["valu_frac"], "hbm_algorithmic_frac": d["hbm_algorithmic_frac"], "hbm_compulsory_frac": d["hbm_compulsory_frac"],
                    "hbm_measured_frac": d["hbm_measured_frac"], "traffic_over_compulsory": d["traffic_over_compulsory"],
                    "algorithmic_bytes_per_launch": round(d["algorithmic_bytes_per_step"] * per_launch, 1),
                    "compulsory_bytes_per_launch": round(d["compulsory_bytes_per_step"] * per_launch, 1),
                    "pmc": bool(pmc.get("all_kernels")), "pmc_source": pmc.get("source"),
                    "definitions": "DESIGN.md section 4: frac is the fraction of the bound named in `bound`; hbm_algorithmic_frac uses SURVEY 8d's bytes (one map read per keyframe) and may "
                                   "exceed 1 for a projection kernel, which reads the map once per launch (hbm_compulsory_frac); traffic = PMC HBM bytes per launch"}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.workload == DEFAULT_WORKLOAD or args.cpu_baseline_now:
            if n_sessions == 2:
                cpu_baseline = run_cpu_baseline(sess_t, three_res, n_kf, args, knn_k, knn_thr, voxel)
        else:
            cpu_baseline = quoted_cpu_baseline(args.workload)
    t_total = None
    if rank == 0 and world == 1 and not args.no_t_total and args.workload == DEFAULT_WORKLOAD:
        t_total = run_t_total(sess_t, n_kf)

    if rank == 0:
        M_c = len(last.outputs["OriginalNoisyCentralMapGlobal"])
        M_q = len(last.outputs["OriginalNoisyQueryMapGlobal"])
        full = {
            "metric": "keyframe-pairs/sec (removert+diff)", "value": round(value, 3), "unit": "keyframe-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            # the same region -- makeGlobalMap + Steps 1-3, loaded sessions resident, no output files, warm -- timed by the C++ host itself
            # (`ltm_run --bench`, the north-star boundary) in this very invocation; its kernel-class launch counts are in t_total.cxx_host_bench and
            # tests/test_gpu_cli.py requires them to equal the Python host's: one pipeline, two drivers.  `..._one_shot_...` is the cold files -> files run
            "cxx_host_ms_per_step": (round(t_total["cxx_host_bench"]["ms_per_step"], 3)
                                     if t_total and isinstance(t_total.get("cxx_host_bench"), dict) and t_total["cxx_host_bench"].get("ms_per_step") else None),
            "cxx_host_one_shot_steps123_ms": (round(1e3 * t_total["configs[1] 2x500 3-res"]["T_steps123_s"], 1)
                                              if t_total and isinstance(t_total.get("configs[1] 2x500 3-res"), dict) and t_total["configs[1] 2x500 3-res"].get("T_steps123_s") else None),
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "dtype_detail": "f32 (+f64 rigid transforms, u64 range|index atomics)", "data": "synthetic (tools/synth.py synth-v1, seed 20250224)",
            "config": {"workload": args.workload, "sessions": f"{scene} 01 vs 02" if n_sessions == 2 else f"{scene} cascade 01 -> 02..{n_sessions:02d} ({n_sessions - 1} chained pair runs)",
                       "keyframes_per_session": n_kf, "keyframe_pairs_per_step": pairs_per_step, "sensor": sensor,
                       "remove_resolution_list": P.remove_resolution_list if three_res else [2.5], "self_removert": three_res,
                       "knn": {"k": knn_k, "thr": knn_thr}, "voxel": voxel, "map_points_last_pair": [M_c, M_q],
                       "scan_points": [int(s["offsets"][-1]) for s in sess_t],
                       "parallelism": (f"keyframe-sharded x{world} (label all-reduce, key-range all-to-all for the merges" + (", one rank group per session in Step 1" if dist_session_groups(world) else "") + ")") if world > 1 else
                                      ("single GPU, the two sessions' merge + Step-1 chains side by side on two contexts (--overlap-sessions experiment)" if args.overlap_sessions else
                                       ("single GPU, two lanes (independent chains of run() side by side on two streams / host threads)" if two_lanes else "single GPU, one lane")),
                       "step": "makeGlobalMap + Removerter::run Steps 1-3 per pair run, inputs resident in HBM"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "rooflines": rooflines, "traffic_groups": groups or None,
            "t_total": t_total, "parity_fullsize": parity_fullsize_status(),
            "lanes": 2 if two_lanes else 1,
            "one_lane_ms_per_step": round(one_lane_ms, 3) if one_lane_ms else None,
            "class_times_from": (f"a one-lane profiling pass of {psteps} steps after the two-lane timed region (HIP events of overlapping launches would count the other lane's work)"
                                 if two_lanes else "the timed region"),
            "scaling_model": scaling_model({k: v["ms"] / psteps for k, v in prof.items()}, one_lane_ms or ms_per_step,
                                           {k: (v[0] / psteps, v[1] / psteps) for k, v in meter.events.items()},
                                           sharded_voxel_fraction=(meter.sharded_voxel_points / max(prof.get("voxel", {}).get("units", 0.0), 1.0)),
                                           step1=({"class_ms": {k: v / psteps for k, v in snap["step1_ms"].items()}, "wall_ms": 1e3 * snap["step1_wall"] / psteps,
                                                   "events": {k: (v[0] / psteps, v[1] / psteps) for k, v in snap["step1_events"].items()},
                                                   "sharded_voxel_fraction": snap["step1_svp"] / max(snap["step1_voxel_units"], 1.0),
                                                   "swap_bytes": snap["swap_bytes"] / psteps} if snap["on"] and snap["step1_wall"] > 0 else None)) if meter else None,
            "stage_ms": {k: round(1e3 * v, 2) for k, v in last.timings.items()},
            "timed_region_stage_ms": timed_stage_ms,
            "kernel_classes_ms_per_step": {k: round(v["ms"] / psteps, 3) for k, v in sorted(prof.items())},
            "voxel_grids": {"per_step": round(vox_grids / max(psteps, 1), 2), "recognised_as_identity_per_step": round(vox_identity / max(psteps, 1), 2),
                            "what": "voxel grids of clouds per step and how many of them the bounding-box pass recognised as the identity (DESIGN.md 4.2)"},
            "vote_cull": {"points_tested": cull_pts, "needed_exact_path": cull_surv, "fraction": round(cull_surv / max(cull_pts, 1), 4)},
            "synth_generation_s": round(t_gen, 2),
        }
        extra_path = write_extra(full, args)
        print(json.dumps(slim_line(full, extra_path)))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


def dist_session_groups(world):
    """does the torch.distributed host split an even world into two session groups?  (dist.ShardedOps._session_groups_on: opt-in over RCCL)"""
    e = os.environ.get("LTM_SESSION_GROUPS")
    return world >= 2 and world % 2 == 0 and e is not None and e not in ("0", "False")


EXTRA_DEFAULT = os.path.join("profiles", "bench_extra_latest.json")


def write_extra(full, args):
    """everything the printed line does not carry (per-class rooflines, traffic groups, t_total, scaling model, stage times, prose) goes to a sidecar
    file: the driver keeps the last ~8 KB of stdout and could not parse round 4's 21 KB line"""
    path = args.extra_out or os.path.join(ROOT, EXTRA_DEFAULT)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT)
    except Exception as e:
        print(f"bench.py: sidecar {path} not written: {e!r}", file=sys.stderr)
        return None


def _num(d, *keys):
    return {k: d.get(k) for k in keys if k in d} if isinstance(d, dict) else None


def slim_line(full, extra_path):
    """the ONE printed JSON line: contract fields, the dominant kernel's roofline (numbers only), the CPU baseline (numbers + one short sentence),
    a per-class table of numbers, and the path of the sidecar.  Stays well below 6000 bytes (tests/test_bench_line.py)"""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "cxx_host_ms_per_step", "lanes", "one_lane_ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full.get(k) for k in keep}
    cfg = full.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in ("workload", "keyframes_per_session", "keyframe_pairs_per_step", "sensor", "remove_resolution_list", "knn", "voxel",
                                             "map_points_last_pair", "parallelism") if k in cfg}
    if isinstance(out["config"].get("parallelism"), str):
        out["config"]["parallelism"] = out["config"]["parallelism"][:120]
    out["roofline"] = _num(full.get("roofline"), "kernel", "class", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step",
                           "share_of_step", "valu_frac", "hbm_algorithmic_frac", "hbm_compulsory_frac", "hbm_measured_frac", "traffic_over_compulsory", "pmc")
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        o = _num(cb, "value", "unit", "cores", "kind", "measured_s", "keyframe_stride", "extrapolated_step_s", "host_cores", "cgroup_cpu_quota_cpus",
                 "host", "cpu_model", "quoted_from", "same_oracle_sources")
        o["sample"] = str(cb.get("sample_short") or cb.get("sample") or "")[:200]
        ac = cb.get("all_cores")
        if isinstance(ac, dict):
            o["all_cores"] = _num(ac, "value", "cores", "cgroup_cpu_quota_cpus", "measured_s", "not_quoted")
            if ac.get("quoted_from"):      # a number carried over from a file is labelled as such in the line itself
                o["all_cores"]["quoted_from"] = str(ac["quoted_from"]).split(" ")[0]
                o["all_cores"]["measured_now"] = False
            elif ac.get("value") is not None:
                o["all_cores"]["measured_now"] = True
        rc = cb.get("reference_compiled")
        if isinstance(rc, dict) and "port_speedup_over_reference_compiled" in rc:
            o["port_speedup_over_reference_compiled"] = rc["port_speedup_over_reference_compiled"]
        out["cpu_baseline"] = o
    else:
        out["cpu_baseline"] = None
    # per kernel class: ms per step, bound, fraction of that bound, measured traffic / compulsory bytes
    out["classes"] = [{"c": r.get("class"), "ms": r.get("ms_per_step"), "b": r.get("bound"), "frac": r.get("frac"), "hbm_c": r.get("hbm_compulsory_frac"),
                       "t_over_c": r.get("traffic_over_compulsory")} for r in (full.get("rooflines") or [])[:16]]
    tt = full.get("t_total")
    if isinstance(tt, dict) and isinstance(tt.get("configs[1] 2x500 3-res"), dict):
        out["t_total_s"] = _num(tt["configs[1] 2x500 3-res"], "T_total_s", "T_step0_s", "T_steps123_s")
    pf = full.get("parity_fullsize")
    if isinstance(pf, dict):
        out["parity_fullsize_matches_sources"] = pf.get("matches_sources")
    out["extra"] = extra_path
    return out


def kernels_sha():
    from tools import provenance
    return provenance.kernels_sha()


# kernel groups for the measured HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE name kernels, not stages): substrings of kernel names ->
# the profile classes whose launches those kernels serve.  rocPRIM's sort / scan kernels cannot be told apart by caller, so the sort-based
# stages share one group and the streaming rest another.
TRAFFIC_GROUPS = [
    ("vote_map_cull", ["k_vote_map_cull"], ["vote_map_cull"]),
    ("map_rimg_blockmin", ["k_map_rimg_blockmin", "k_map_rimg_lds"], ["reproject_map", "vote_map_exact"]),
    ("knn_query", ["k_knn_query", "k_knn_fast", "k_knn_slow", "k_knn_queue"], ["knn_query", "knn_query_p2"]),
    ("sort_based (voxel grids, kNN grid build)", ["radix_sort", "merge_sort", "k_voxel", "k_morton", "k_head_flags", "k_segment_starts", "k_bbox", "k_scan_total",
                                                  "k_cell_keys", "k_hash_build", "k_gather_points", "k_gather_u64", "k_compact", "k_key_"], ["voxel", "voxel_scanset", "voxel_grid_scanset", "knn_build"]),
    ("streaming rest (scan images, compare, fills, scans + scatters, merges)", [""], ["vote_scan", "vote_compare", "vote_fill", "partition", "reproject_gather", "merge"]),
]


# the library's own kernels of each class whose traffic the counter pass can attribute to the class alone (kernel names are unique to it);
# rocPRIM's scans / sorts serve several classes and cannot be told apart by caller: classes that contain them also report the group figure
CLASS_OWN_KERNELS = {
    "vote_map_cull": ["k_vote_map_cull"],
    # one kernel, two classes (ND votes and reprojections): its counters are split between them by their share of its event time (SHARED_KERNEL_CLASSES)
    "vote_map_exact": ["k_map_rimg_blockmin", "k_map_rimg_lds", "k_pair_shell_select", "k_coarse_max"],
    "reproject_map": ["k_map_rimg_blockmin", "k_map_rimg_lds", "k_pair_shell_select", "k_coarse_max"],
    "vote_scan": ["k_scan_rimg", "k_image_max", "k_scan_qbound", "k_image_bounds"],
    "vote_compare": ["k_compare_flag"],
    "partition": ["k_partition_scatter"],
    "reproject_gather": ["k_reproject_gather"],
    "merge": ["k_transform_scans", "k_zip_concat"],
    "voxel": ["k_bbox_init", "k_bbox_reduce(", "k_bbox_reduce_check", "k_morton_keys_packed", "k_morton_keys(", "k_voxel_heads_starts", "k_head_flags", "k_segment_starts",
              "k_voxel_centroids", "k_scan_total"],
    "voxel_scanset": ["k_bbox_reduce_seg", "k_bbox_init_seg", "k_morton_keys_seg"],
    "knn_build": ["k_cell_keys", "k_hash_build", "k_gather_points", "k_knn_bucket_build", "k_knn_bitmap_build"],
    "knn_query": ["k_knn_fast", "k_knn_query_cloud", "k_knn_query_scans"],
    "knn_query_p2": ["k_knn_slow", "k_knn_queue_scatter"],
}
SHARED_KERNEL_CLASSES = ("vote_map_exact", "reproject_map")
# (vote_fill is not listed: k_fill_u64 / k_fill_u32 also initialise images, tables and flags of other stages, so their counters are not the class's own)
CLASSES_WITH_LIBRARY_KERNELS = {"partition": "rocprim scan", "reproject_gather": "rocprim scan", "voxel": "rocprim radix sort", "voxel_scanset": "rocprim radix sort + scan",
                                "voxel_grid_scanset": "rocprim radix sort + scan", "knn_build": "rocprim radix sort", "knn_query_p2": "rocprim scan"}


def class_own_counter(pmc, cls, what, share=1.0):
    """per step, over the kernels that belong to `cls` alone: what = "traffic" -> measured HBM bytes ((2 FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md),
    what = "valu" -> SQ_INSTS_VALU wave-instructions.  (None, []) without counter data collected for these sources and this workload"""
    allk = pmc.get("all_kernels")
    subs = CLASS_OWN_KERNELS.get(cls)
    if not allk or not subs:
        return None, []
    mine = [k for k in allk if any(sub in k or sub in k + "(" for sub in subs)]
    if not mine:
        return None, []
    names = sorted(k.split("::")[-1] for k in mine)
    if what == "valu":
        if not all("SQ_INSTS_VALU" in allk[k] for k in mine):
            return None, names
        return share * sum(allk[k]["SQ_INSTS_VALU"].get("sum", 0.0) for k in mine), names
    fetch = sum(allk[k].get("FETCH_SIZE", {}).get("sum", 0.0) for k in mine)
    write = sum(allk[k].get("WRITE_SIZE", {}).get("sum", 0.0) for k in mine)
    return share * (2.0 * fetch + write) * 1024.0, names


def traffic_groups(pmc, prof, steps):
    allk = pmc.get("all_kernels")
    if not allk:
        return []
    left = dict(allk)
    out = []
    for name, subs, classes in TRAFFIC_GROUPS:
        mine = [k for k in list(left) if any(sub in k for sub in subs) and "k_selfcheck" not in k]
        fetch = sum(left[k].get("FETCH_SIZE", {}).get("sum", 0.0) for k in mine)
        write = sum(left[k].get("WRITE_SIZE", {}).get("sum", 0.0) for k in mine)
        for k in mine:
            left.pop(k)
        traffic = (2.0 * fetch + write) * 1024.0       # per step: the counter passes run --steps 1 --warmup 0
        alg = sum(prof[c]["bytes"] for c in classes if c in prof) / max(steps, 1)
        comp = sum(prof[c].get("bytes_c", prof[c]["bytes"]) for c in classes if c in prof) / max(steps, 1)
        ms = sum(prof[c]["ms"] for c in classes if c in prof) / max(steps, 1)
        out.append({"group": name, "classes": classes, "kernels_matched": len(mine), "traffic_bytes_per_step": round(traffic, 1),
                    "algorithmic_bytes_per_step": round(alg, 1), "traffic_over_algorithmic": round(traffic / alg, 3) if alg else None,
                    "compulsory_bytes_per_step": round(comp, 1), "traffic_over_compulsory": round(traffic / comp, 3) if comp else None,
                    "ms_per_step": round(ms, 3), "hbm_measured_frac": round(traffic / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4) if ms else None})
    return out


def parity_fullsize_status():
    """is the committed full-size parity record (tools/parity_fullsize.py, tests/test_gpu_fullsize_parity.py) about THESE sources?"""
    from tools import provenance
    path = os.path.join(ROOT, "profiles", "parity_fullsize_2x500_3res.json")
    try:
        d = json.load(open(path))
    except Exception:
        return {"record": None, "matches_sources": False}
    return {"record": "profiles/parity_fullsize_2x500_3res.json", "outputs_compared": d.get("outputs_compared"), "outputs_differing": d.get("outputs_differing"),
            "record_product_sha": d.get("product_sha"), "this_product_sha": provenance.product_sha(),
            "matches_sources": d.get("product_sha") == provenance.product_sha() and d.get("outputs_differing") == 0}


def run_t_total(sess_t, n_kf):
    """SURVEY 8d T_total: the same two sessions written in the reference's on-disk format, then `ltm_run` files -> files for configs[1]
    (all keyframes, 3-res) and configs[0] (keyframes 0..49, single-res); best of a warm-up + 2 runs each"""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tools import synth, t_total
    root = tempfile.mkdtemp(prefix="ltm_bench_ttotal_")
    try:
        sess = [synth.to_numpy(s) for s in sess_t]
        c1, dirs = t_total.measure(sess, n_kf, three_res=True, runs=3, root=root)
        # one orchestration check: the C++ host times the same resident-input region itself (`ltm_run --bench`)
        try:
            cxx = t_total.bench_cxx_host(root, dirs, n_kf, three_res=True, steps=3, warmup=1)
        except Exception as e:
            cxx = {"error": repr(e)[:300]}
        # configs[0]: session directories that hold the first 50 scans only (links to the files just written) and their 50 pose lines --
        # the reference selects the query keyframes among ALL scans of the directory (Session::parseKeyframesInROI)
        n0 = min(50, n_kf)
        root0 = os.path.join(root, "c0")
        dirs0 = []
        for tag, S, d in zip(("01", "02"), sess, dirs):
            d0 = os.path.join(root0, tag, "Scans")
            os.makedirs(d0)
            for k in range(n0):
                os.symlink(os.path.join(d, S["names"][k]), os.path.join(d0, S["names"][k]))
            lines = open(os.path.join(root, tag, "poses.txt")).read().splitlines()[:n0]
            with open(os.path.join(root0, tag, "poses.txt"), "w") as f:
                f.write("\n".join(lines) + "\n")
            dirs0.append(d0)
        c0, _ = t_total.measure([dict(S, offsets=S["offsets"][:n0 + 1]) for S in sess], n0, three_res=False, runs=2, root=root0, dirs=dirs0)

        def brief(r):
            b = r["best"]
            return {"T_total_s": b["T_total"], "T_step0_s": b.get("T_step0"), "T_steps123_s": b.get("T_steps123"), "T_scan_writes_s": b.get("T_scan_writes"),
                    "keyframes": b["keyframes"], "input_bytes": r["input_bytes"], "output_bytes": b["output_bytes"],
                    "keyframe_pairs_per_s_incl_io": r["keyframe_pairs_per_s_incl_io"]}
        return {"what": "lt-mapper_amd/host/ltm_run, files -> files (PCD scan directories + pose files in, 16 maps + 5 x N_c scan files out), page cache warm",
                "configs[1] 2x500 3-res": brief(c1), "configs[0] 2x50 single-res": brief(c0), "cxx_host_bench": cxx}
    except Exception as e:      # the C++ host missing or failing must not cost the bench line
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def load_pmc(workload):
    """PMC counters per kernel of one step of `workload` (rocprofv3 cannot run inside this process): profiles/pmc_<workload>.json (the default
    workload's also as profiles/pmc_latest.json) are written by tools/collect_profiles.sh, stamped with the commit and a hash of the kernel sources;
    a file is used only if that hash equals the sources this run was built from and the workload matches -- otherwise traffic / VALU
    fractions are null, never stale."""
    tried = []
    for name in (f"pmc_{workload}.json", "pmc_latest.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") != workload or d.get("kernels_sha") != kernels_sha():
            tried.append(f"profiles/{name}: {d.get('workload')} / {d.get('kernels_sha')}")
            continue
        return {"all_kernels": d.get("all_kernels"),
                "source": f"profiles/{name} (commit {d.get('commit')}, kernel sources {d.get('kernels_sha')}): separate rocprofv3 --pmc passes of one step; "
                          "(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes, SQ_INSTS_VALU*64 lane-instructions"}
    return {"source": f"no counter file for workload {workload} / kernel sources {kernels_sha()} (ignored: {'; '.join(tried) or 'none found'})"}


def quoted_cpu_baseline(workload):
    """configs[2..4]: the sampled single-thread CPU leg of SURVEY 8d costs minutes to an hour of host time on these sizes, which the GPU box's budget cannot
    carry: tools/cpu_baseline_sampled.py runs it (the same oracle, the same generator and seed, a stated keyframe sample) on a host without GPU and commits
    profiles/cpu_baseline_<workload>.json; the line quotes it together with the host it was measured on"""
    from tools import provenance
    path = os.path.join(ROOT, "profiles", f"cpu_baseline_{workload}.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    d = dict(d, quoted_from=f"profiles/cpu_baseline_{workload}.json", same_oracle_sources=d.get("oracle_sha") == provenance.oracle_sha())
    return d


def run_cpu_baseline(sess_t, three_res, n_kf, args, knn_k=2, knn_thr=0.01, voxel=0.05):
    """CPU oracle (port of the reference's algorithm) on a bounded sample: the whole pipeline runs on the full-size sessions but
    every per-keyframe loop visits only each `stride`-th keyframe; per-keyframe stage times are scaled by the true visit ratio,
    whole-map stages (voxel grids, kd-tree builds) are timed in full.  Two legs: one thread (north-star denominator) and all
    cores (the reference's OpenMP sites).  The committed full unsampled runs (profiles/, tools/cpu_baseline_full.py) are quoted."""
    import numpy as np
    from oracle import oracle_py as orc
    from tools import synth
    C, Q = (synth.to_numpy(s) for s in sess_t)
    for S in (C, Q):   # pre-clean like the GPU side (Step 0)
        pts, off = [], [0]
        for k in range(len(S["offsets"]) - 1):
            p = orc.preclean(S["scans"][int(S["offsets"][k]):int(S["offsets"][k + 1])], 2.5)
            pts.append(p); off.append(off[-1] + len(p))
        S["scans"], S["offsets"] = np.concatenate(pts), np.array(off, dtype=np.uint64)
    per_kf = ("vote_large", "vote_small", "reproject_large", "reproject_small", "knn_query", "voxel_scanwise")

    def leg(threads, stride):
        stride = max(1, min(stride, n_kf))
        visited = len(range(0, n_kf, stride))
        scale = n_kf / visited
        P = orc.make_params(k=knn_k, knn_thr=knn_thr, voxel=voxel, use_self_removert=three_res, res_list=(2.5, 2.0, 1.5) if three_res else (2.5,),
                            threads=threads, kf_sample_stride=stride)
        t0 = time.perf_counter()
        res = orc.pipeline_run(P, C, Q)
        wall = time.perf_counter() - t0
        tm = res.timings()
        res.free()
        est = sum(v * (scale if k in per_kf else 1.0) for k, v in tm.items() if k != "steps_1_to_3_total")
        if args.verbose:
            print("cpu timings", threads, tm, "wall", wall, "scale", scale, file=sys.stderr)
        return {"value": round(n_kf / est, 4), "unit": "keyframe-pairs/s", "cores": threads, "keyframe_stride": stride,
                "keyframes_visited_per_session": visited, "measured_s": round(wall, 2), "extrapolated_step_s": round(est, 1)}

    ncores = os.cpu_count() or 1

    def cpu_quota():
        """CPUs' worth of time the container may use (cgroup cpu.max / cfs quota), None if unlimited or unreadable.  The gpurun boxes
        show 256 hardware threads and an affinity mask of 256 but `cpu.max = 1600000 100000`: 16 CPUs (profiles/r4_host_cpu_quota.txt),
        so a figure "on all 256 cores" there is a figure on 16 CPUs' worth of time spread over 256 threads."""
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            return None if q == "max" else round(int(q) / int(per), 2)
        except Exception:
            pass
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if q <= 0 else round(q / per, 2)
        except Exception:
            return None
    quota = cpu_quota()
    one = leg(1, args.cpu_stride)
    # the oracle parallelises over keyframes: the visited keyframes must outnumber the threads several times or the scaled stage
    # times overestimate (50 keyframes on 256 threads take one round, 500 take two, not ten) -- with many cores run every keyframe
    # That full run takes ~3 minutes on the 256-core GPU box, which the default invocation cannot afford: it is made with --cpu-allcore
    # (tools/collect_profiles.sh does); otherwise the figure of the committed default line of this round is quoted, marked as such.
    allc = None
    from tools import provenance

    def cpu_model():
        try:
            return next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:
            return None
    if ncores > 1 and args.cpu_allcore:
        allc = leg(ncores, max(1, min(args.cpu_stride_allcore, n_kf // (4 * ncores))))
        allc.update(oracle_sha=provenance.oracle_sha(), host_cores=ncores, cgroup_cpu_quota_cpus=quota, cpu_model=cpu_model(), workload=getattr(args, "workload", None))
    elif ncores > 1:
        # quoted, not measured now -- and only if it was measured with THIS oracle on a host of the same shape: a ratio of
        # a fresh GPU time and a CPU time from another host or another oracle build would mean nothing
        try:
            prev = json.load(open(os.path.join(ROOT, "profiles", "cpu_allcore_latest.json")))
            same = (prev.get("workload") == getattr(args, "workload", None) and prev.get("oracle_sha") == provenance.oracle_sha() and
                    prev.get("host_cores") == ncores and prev.get("cpu_model") == cpu_model())
            if same:
                allc = dict(prev, quoted_from="profiles/cpu_allcore_latest.json (same oracle sources, same CPU model and core count; measured with --cpu-allcore, not re-run now)")
            else:
                allc = {"value": None, "not_quoted": "profiles/cpu_allcore_latest.json was measured with another oracle build / host shape "
                                                     f"({prev.get('oracle_sha')}, {prev.get('host_cores')} x {prev.get('cpu_model')}); run with --cpu-allcore"}
        except Exception:
            allc = None
    full = {}
    for tag in ("1thread", "allcore"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f"r2_cpu_baseline_full_{tag}.json")))
            full[tag] = {k: d[k] for k in ("threads", "nproc", "cpu", "median_wall_s", "keyframe_pairs_per_s", "commit") if k in d}
        except Exception:
            pass
    # REFERENCE-COMPILED code beside the port (round 4): oracle/_ref holds the reference's own sources built against stand-in ROS / Eigen /
    # OpenCV / PCL headers (oracle/refshim).  Its dominant stage -- one vote pass, Removerter.cpp:542-593, three quarters of the CPU time --
    # is timed on the FULL-SIZE central map for a few keyframes, next to the port on the same keyframes: how much the port flatters the CPU.
    ref_cmp = None
    try:
        from oracle import ref_py
        if ref_py.available():
            I4 = np.eye(4)
            cmap = orc.voxel_centroid(orc.merge_to_global(C["scans"], C["offsets"], C["poses"], I4), voxel)
            kfs = [0, n_kf // 2, n_kf - 1][: min(3, n_kf)]
            sub = np.concatenate([C["scans"][int(C["offsets"][k]):int(C["offsets"][k + 1])] for k in kfs])
            sub_off = np.cumsum([0] + [int(C["offsets"][k + 1] - C["offsets"][k]) for k in kfs]).astype(np.uint64)
            poses = np.asarray(C["poses"]).reshape(-1, 16)[kfs]
            inv = orc.inverse_poses(poses)
            R = ref_py.Removerter(ref_py.make_params(k=knn_k, knn_thr=knn_thr, voxel=voxel))
            t0 = time.perf_counter(); lab_r = R.vote_labels(cmap, sub, sub_off, poses, 2.5, 0); t_ref = time.perf_counter() - t0
            t0 = time.perf_counter(); lab_o = orc.vote_labels(cmap, sub, sub_off, inv, I4, 50.0, 360.0, 2.5, 0.1, 0, threads=1); t_orc = time.perf_counter() - t0
            R.close()
            ref_cmp = {"kind": "reference", "what": "oracle/_ref/libltm_ref.so (the reference's unmodified sources compiled against stand-in headers, serial): one vote pass "
                                                    "(Removerter.cpp:542-593) of the full-size central map against 3 keyframes, and the port on the same input",
                       "map_points": int(len(cmap)), "keyframes": len(kfs), "reference_compiled_s_per_keyframe": round(t_ref / len(kfs), 3),
                       "port_s_per_keyframe": round(t_orc / len(kfs), 3), "port_speedup_over_reference_compiled": round(t_ref / t_orc, 2),
                       "labels_identical": bool((lab_r == lab_o).all())}
    except Exception as e:          # the checker's checker must never break the bench line
        ref_cmp = {"kind": "reference", "error": repr(e)[:200]}
    out = {"value": one["value"], "unit": "keyframe-pairs/s", "cores": 1, "kind": "port", "reference_compiled": ref_cmp, "keyframe_stride": one["keyframe_stride"],
           "sample_short": f"CPU oracle (port), 1 thread, full-size sessions, every {one['keyframe_stride']}th keyframe ({one['keyframes_visited_per_session']} of {n_kf}) in per-keyframe "
                           f"loops scaled up, whole-map stages in full",
           "sample": f"oracle/libltm_oracle.so (sort-based voxel grid + kd-tree: faster than the PCL-based reference), full-size sessions, every "
                     f"{one['keyframe_stride']}th keyframe ({one['keyframes_visited_per_session']} of {n_kf} per session) in the per-keyframe loops (votes, "
                     f"reprojections, kNN queries) scaled x{n_kf / one['keyframes_visited_per_session']:.1f}; voxel grids and kd-tree builds timed in full; "
                     f"{one['measured_s']:.1f} s measured, {one['extrapolated_step_s']:.0f} s extrapolated per step",
           "measured_s": one["measured_s"], "extrapolated_step_s": one["extrapolated_step_s"], "host_cores": ncores, "cgroup_cpu_quota_cpus": quota,
           "all_cores": allc,
           "full_unsampled_runs_committed": full or None}
    return out


if __name__ == "__main__":
    main()
