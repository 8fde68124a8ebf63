#!/usr/bin/env python3
"""bench.py -- keyframe-pairs/sec of the LT-removert / LT-map hot path (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched one rank per GPU by torch.distributed.run)

A *step* is one full pass of the hot path over one synthetic session pair that is already resident in HBM:
makeGlobalMap + Removerter::run() Steps 1-3 (Removerter.cpp:1653-1678) = remove/revert visibility votes, static
reprojection, inter-session kNN change detection, ND/PD filtering, LT-map composition and the final reprojections.
Default workload = BASELINE.json configs[1]: "ParkingLot 01 vs 02, 500 keyframes each, 3-res removert, 1xMI355X"
restated on the synthetic `lot` scene (tools/synth.py; no dataset is reachable offline).

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     -- the dominant kernel (k_map_rimg, class "vote_map"): algorithmic bytes per launch / average launch
                  duration measured with HIP events on the context's stream, against the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (a port of the reference; the reference itself cannot be built here) timed on this
                  box on a bounded keyframe sample of the same workload, single thread
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling

DEFAULT_WORKLOAD = "lot-2x500-os1-64-3res"
WORKLOADS = {
    # name: (sensor, keyframes per session, 3-res?, scene, kf spacing [m], voxel [m], kNN k, kNN thr)
    "lot-2x500-os1-64-3res": ("os1-64", 500, True, "lot", 1.0, 0.05, 2, 0.01),        # BASELINE configs[1] (default)
    "lot-2x50-os1-64-1res": ("os1-64", 50, False, "lot", 1.0, 0.05, 2, 0.01),         # configs[0] shape (plumbing case)
    "lot-2x100-small-3res": ("small", 100, True, "lot", 1.0, 0.05, 2, 0.01),          # quick check
    "street-2x2000-hdl64e-1res": ("hdl-64e", 2000, False, "street", 1.0, 0.05, 2, 0.01),   # configs[3], KITTI-scale
    "street-2x2000-hdl64e-3res": ("hdl-64e", 2000, True, "street", 1.0, 0.05, 2, 0.01),
    "street-2x200-mls-knn": ("mls", 200, False, "street", 2.0, 0.1, 2, 0.04),         # configs[4], dense MLS kNN stress
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-stride", type=int, default=100, help="cpu_baseline: visit every s-th keyframe in per-keyframe loops")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback for the measured path"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    from tools import synth

    sensor, n_kf, three_res, scene, spacing, voxel, knn_k, knn_thr = WORKLOADS[args.workload]
    dev = f"cuda:{local_rank}"
    t0 = time.perf_counter()
    # synthetic sessions 01 / 02, generated on the GPU, already in HBM when the timed region starts
    sess_t = [synth.make_session(s, n_kf, sensor, device=dev, scene=scene, kf_spacing=spacing) for s in (1, 2)]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0

    ctx = capi.Context(vfov=50.0, hfov=360.0, device=local_rank)
    P = Params(gpu_use_self_removert=three_res, remove_resolution_list=[2.5, 2.0, 1.5] if three_res else [2.5],
               num_nn_points_within=knn_k, dist_nn_points_within=knn_thr, downsample_voxel_size=voxel)

    def fresh_sessions():
        out = []
        for name, S in zip(("Central", "Query"), sess_t):
            scans = ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64))
            scans = ctx.preclean(scans, 2.5)                      # precleaningKeyframes(2.5), Removerter.cpp:1660
            out.append(Session(name, scans, ctx.poses(S["poses"], S["inv"])))
        return out

    if world > 1:
        from ltmapper_amd.dist import ShardedOps
        ops = ShardedOps(HipOps(ctx), dist, rank, world)
    else:
        ops = HipOps(ctx)

    sessions = fresh_sessions()   # loading + pre-clean are Step 0 plumbing, outside the timed region

    def one_step():
        ctx.clear_caches()   # no derived data (scan range images) survives from a previous step: every step is a fresh run
        C, Q = sessions
        rm = Removerter(ops, P, Session("Central", C.keyframe_scans_, C.keyframe_poses), Session("Query", Q.keyframe_scans_, Q.keyframe_poses))
        rm.run()
        return rm

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    last = None
    for _ in range(args.warmup):
        last = one_step()
    barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    cull_surv, cull_pts = ctx.cull_stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs_per_step = n_kf                      # min(N_c, N_q) keyframe pairs per session pair
    value = pairs_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # dominant kernel: k_vote_map_cull (profile class "vote_map_cull"; falls back to the exact kernel if culling is disabled)
    cls = "vote_map_cull" if prof.get("vote_map_cull", {}).get("launches") else "vote_map_exact"
    vm = prof.get(cls, dict(ms=0.0, launches=0, units=0.0, bytes=0.0))
    roofline = None
    if vm["launches"]:
        achieved = vm["bytes"] / (vm["ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_vote_map_cull" if cls == "vote_map_cull" else "k_map_rimg_blockmin",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": measured_traffic(cls, args.workload), "launches_per_step": vm["launches"] // max(args.steps, 1),
                    "avg_launch_ms": round(vm["ms"] / vm["launches"], 4),
                    "algorithmic_bytes_per_launch": round(vm["bytes"] / vm["launches"], 1),
                    "algorithmic_definition": "nb*(16*M + 8*R*C) per launch over nb keyframes (SURVEY 8d: map read + range|index image)",
                    "point_projections_per_s": round(vm["units"] / (vm["ms"] * 1e-3), 1)}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(sess_t, three_res, n_kf, args.cpu_stride, args.verbose, knn_k, knn_thr, voxel)

    if rank == 0:
        M_c = len(last.outputs["OriginalNoisyCentralMapGlobal"])
        M_q = len(last.outputs["OriginalNoisyQueryMapGlobal"])
        out = {
            "metric": "keyframe-pairs/sec (removert+diff)", "value": round(value, 3), "unit": "keyframe-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32 (+f64 rigid transforms, u64 range|index atomics)", "data": "synthetic (tools/synth.py synth-v1, seed 20250224)",
            "config": {"workload": args.workload, "sessions": f"{scene} 01 vs 02", "keyframes_per_session": n_kf, "sensor": sensor,
                       "remove_resolution_list": P.remove_resolution_list if three_res else [2.5], "self_removert": three_res,
                       "knn": {"k": knn_k, "thr": knn_thr}, "voxel": voxel, "map_points": [M_c, M_q],
                       "scan_points": [int(s["offsets"][-1]) for s in sess_t],
                       "parallelism": f"keyframe-sharded x{world} (label all-reduce + scan all-gather)" if world > 1 else "single GPU",
                       "step": "makeGlobalMap + Removerter::run Steps 1-3, inputs resident in HBM"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            "stage_ms": {k: round(1e3 * v, 2) for k, v in last.timings.items()},
            "kernel_classes_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items())},
            "vote_cull": {"points_tested": cull_pts, "needed_exact_path": cull_surv, "fraction": round(cull_surv / max(cull_pts, 1), 4)},
            "synth_generation_s": round(t_gen, 2),
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def measured_traffic(cls, workload):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run
    inside this process; the file records the command).  None if no measurement exists for this kernel and workload."""
    path = os.path.join(ROOT, "profiles", "r1_final_pmc_hbm_traffic.json")
    if cls != "vote_map_cull" or workload != DEFAULT_WORKLOAD or not os.path.exists(path):
        return None
    try:
        return round(json.load(open(path))["hbm_bytes_per_launch"], 1)
    except Exception:
        return None


def run_cpu_baseline(sess_t, three_res, n_kf, stride, verbose, knn_k=2, knn_thr=0.01, voxel=0.05):
    """CPU oracle (port of the reference's algorithm, single thread) on a bounded sample: the whole pipeline runs on
    the full-size sessions but every per-keyframe loop visits only each `stride`-th keyframe; per-keyframe stage times are
    scaled by the true visit ratio, whole-map stages (voxel grids, kd-tree builds) are timed in full."""
    from oracle import oracle_py as orc
    from tools import synth
    C, Q = (synth.to_numpy(s) for s in sess_t)
    for S in (C, Q):   # pre-clean like the GPU side (Step 0)
        pts, off = [], [0]
        for k in range(len(S["offsets"]) - 1):
            p = orc.preclean(S["scans"][int(S["offsets"][k]):int(S["offsets"][k + 1])], 2.5)
            pts.append(p); off.append(off[-1] + len(p))
        S["scans"], S["offsets"] = np.concatenate(pts), np.array(off, dtype=np.uint64)
    stride = max(1, min(stride, n_kf))
    visited = len(range(0, n_kf, stride))
    scale = n_kf / visited
    P = orc.make_params(k=knn_k, knn_thr=knn_thr, voxel=voxel, use_self_removert=three_res, res_list=(2.5, 2.0, 1.5) if three_res else (2.5,),
                        threads=1, kf_sample_stride=stride)
    t0 = time.perf_counter()
    res = orc.pipeline_run(P, C, Q)
    wall = time.perf_counter() - t0
    tm = res.timings()
    per_kf = ("vote_large", "vote_small", "reproject_large", "reproject_small", "knn_query", "voxel_scanwise")
    est = sum(v * (scale if k in per_kf else 1.0) for k, v in tm.items() if k != "steps_1_to_3_total")
    if verbose:
        print("cpu timings", tm, "wall", wall, "scale", scale, file=sys.stderr)
    return {"value": round(n_kf / est, 4), "unit": "keyframe-pairs/s", "cores": 1, "kind": "port",
            "sample": f"oracle/libltm_oracle.so, full-size sessions, every {stride}th keyframe ({visited} of {n_kf} per session) in the per-keyframe "
                      f"loops (votes, reprojections, kNN queries) scaled x{scale:.1f}; voxel grids and kd-tree builds timed in full; "
                      f"{wall:.1f} s measured, {est:.0f} s extrapolated per step",
            "measured_s": round(wall, 2), "extrapolated_step_s": round(est, 1)}


if __name__ == "__main__":
    main()
