#!/usr/bin/env python3
"""Full-size parity of the BASELINE configurations: the GPU pipeline (through the C ABI) against the CPU oracle on the SAME
session pair at the configuration's REAL sensor size -- every map and every per-keyframe scan set compared bitwise.

    python tools/parity_fullsize.py --config 1 [--kf 500] [--threads N] > profiles/<name>.json

  --config 1   configs[1]: lot, os1-64 (64 x 1024), 2 x 500 keyframes, 3-res selfRemovert          (~200 s of oracle on 256 cores)
  --config 2   configs[2] (cascade): lot, os1-64, sessions 01 -> 02 -> 03, 500 keyframes each, 3-res: BOTH pair runs, the second one fed with
               what the reference's loader makes of run 1's scans_updated (VoxelGrid + pre-clean) -- device hand-over vs the oracle chain
  --config 3   configs[3]: street, hdl-64e (64 x 1900), 2 x 200 keyframes, single-res, whole pipeline
  --config 4   configs[4]: street, mls-128x8192 (1 M rays), 2 x 20 keyframes at 2 m, voxel 0.1, k = 2, thr 0.04
  --config 33  configs[3], 3-res variant: street, hdl-64e, 2 x 200 keyframes, selfRemovert over [2.5, 2.0, 1.5]

The oracle needs ~49 min single-threaded for configs[1], far beyond a test; with the GPU box's 256 host cores (OpenMP over
keyframes, same serial arg-min semantics per keyframe; the label union is order-free) it takes about three minutes --
tests/test_gpu_fullsize_parity.py runs it on every box with >= 64 cores.  The record is stamped with hashes of the product and
oracle sources so that bench.py can say whether it belongs to the kernels it measures.  TEST INFRASTRUCTURE: the oracle is the
checker here, never the thing measured."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (scene, sensor, default keyframes, spacing, 3-res, voxel, k, thr)
    1: ("lot", "os1-64", 500, 1.0, True, 0.05, 2, 0.01),
    2: ("lot", "os1-64", 500, 1.0, True, 0.05, 2, 0.01),
    3: ("street", "hdl-64e", 200, 1.0, False, 0.05, 2, 0.01),
    4: ("street", "mls", 20, 2.0, False, 0.1, 2, 0.04),
    33: ("street", "hdl-64e", 200, 1.0, True, 0.05, 2, 0.01),       # configs[3] is "single-res and 3-res" (SURVEY 8d row 4): the 3-res variant
}


def _compare_run(np, rm, ref, MAPS, report, prefix=""):
    bad = 0
    for name in MAPS:
        want, got = ref.cloud(name), rm.outputs.get(name)
        if want is None or got is None:
            ok = want is None and (got is None or len(got) == 0)
            report[prefix + name] = {"points": None, "identical": ok}
        else:
            g = got.download()
            ok = g.shape == want.shape and bool((g.view(np.uint32) == want.view(np.uint32)).all())
            report[prefix + name] = {"points": int(len(want)), "identical": ok}
        bad += 0 if ok else 1
    for name, ss in rm.scan_outputs().items():
        w_pts, w_off = ref.scanset(name)
        g_pts, g_off = ss.download()
        ok = bool((g_off == w_off).all()) and g_pts.shape == w_pts.shape and bool((g_pts.view(np.uint32) == w_pts.view(np.uint32)).all())
        report[prefix + name] = {"points": int(len(w_pts)), "keyframes": int(len(w_off) - 1), "identical": ok}
        bad += 0 if ok else 1
    return bad


def run_cascade_parity(kf=None, threads=None, device="cuda:0", n_sessions=3, lanes=2):
    """configs[2]: the device cascade (lt-mapper_amd/cascade.py) against the oracle chain in which every link re-loads scans_updated like the
    reference (per-scan pcl::VoxelGrid, Session.cpp:284-289, then precleaningKeyframes(2.5), Removerter.cpp:1658-1660)"""
    import numpy as np
    import torch
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from ltmapper_amd.cascade import run_cascade
    from ltmapper_amd.removerter import HipOps, Params
    from oracle import oracle_py as orc
    from tools import provenance, synth
    from test_gpu_pipeline import MAPS
    scene, sensor, kf_default, spacing, three_res, voxel, k, thr = CONFIGS[2]
    kf = kf or kf_default
    threads = threads or (os.cpu_count() or 1)
    res = (2.5, 2.0, 1.5)
    sess_t = [synth.make_session(s, kf, sensor, device=device, scene=scene, kf_spacing=spacing) for s in range(1, n_sessions + 1)]
    torch.cuda.synchronize()
    ctx = capi.Context(vfov=50.0, hfov=360.0, device=0)
    loaded = []
    for S in sess_t:
        scans = ctx.preclean(ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5)
        loaded.append((scans, ctx.poses(S["poses"], S["inv"])))
    P = Params(gpu_use_self_removert=True, remove_resolution_list=list(res), num_nn_points_within=k, dist_nn_points_within=thr, downsample_voxel_size=voxel)
    t0 = time.perf_counter()
    ops = HipOps(ctx)
    runs = run_cascade(ops, P, loaded[0][0], loaded[0][1], loaded[1:], lane_ops=ops.lane() if lanes == 2 else None)      # the shipped schedule: two lanes
    ctx.synchronize()
    t_gpu = time.perf_counter() - t0
    cpu = []
    for (scans, _), S in zip(loaded, sess_t):
        pts, off = scans.download()
        cpu.append(dict(scans=pts, offsets=off, poses=S["poses"], inv=S["inv"]))
    op = orc.make_params(k=k, knn_thr=thr, voxel=voxel, use_self_removert=True, res_list=res, threads=threads)
    report, bad, t_cpu = {}, 0, 0.0
    central = cpu[0]
    for j, rm in enumerate(runs):
        t0 = time.perf_counter()
        ref = orc.pipeline_run(op, central, cpu[j + 1])
        t_cpu += time.perf_counter() - t0
        bad += _compare_run(np, rm, ref, MAPS, report, prefix=f"run{j + 1}/")
        upd_pts, upd_off = ref.scanset("scans_updated")
        re_pts, re_off = [], [0]
        for kk in range(len(upd_off) - 1):      # the loader's VoxelGrid + pre-clean on what the reference would read back from scans_updated/
            q = orc.preclean(orc.voxel_grid(upd_pts[int(upd_off[kk]):int(upd_off[kk + 1])], voxel), 2.5)
            re_pts.append(q); re_off.append(re_off[-1] + len(q))
        central = dict(scans=np.concatenate(re_pts), offsets=np.array(re_off, np.uint64), poses=cpu[0]["poses"], inv=cpu[0]["inv"])
        ref.free()
    ctx.close()
    try:
        commit = open(os.path.join(ROOT, ".commit_for_profiles")).read().strip()
    except OSError:
        commit = None
    return {"what": "device cascade (C ABI, lt-mapper_amd/cascade.py) vs the CPU oracle chain, every output of every pair run compared bitwise",
            "config": "BASELINE configs[2]", "lanes": lanes, "workload": f"{scene} cascade 01 -> 02..{n_sessions:02d}, {kf} keyframes per session, {sensor}, 3-res",
            "scan_points": [int(c["offsets"][-1]) for c in cpu], "gpu_run_s": round(t_gpu, 3), "oracle_run_s": round(t_cpu, 1), "oracle_threads": threads,
            "outputs_compared": len(report), "outputs_differing": bad, "product_sha": provenance.product_sha(), "kernels_sha": provenance.kernels_sha(),
            "oracle_sha": provenance.oracle_sha(), "commit": commit, "outputs": report}


def run_parity(config=1, kf=None, threads=None, device="cuda:0", sessions=3, lanes=2):
    """returns the report dict; report["outputs_differing"] == 0 means bitwise parity of all outputs.  lanes = 2: the schedule the hosts ship (independent
    chains side by side on the context and its lane, Removerter.run_two_lanes); 1: the one-lane order"""
    if config == 2:
        return run_cascade_parity(kf, threads, device, n_sessions=sessions, lanes=lanes)
    import numpy as np
    import torch
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    from oracle import oracle_py as orc
    from tools import provenance, synth
    from test_gpu_pipeline import MAPS
    scene, sensor, kf_default, spacing, three_res, voxel, k, thr = CONFIGS[config]
    kf = kf or kf_default
    threads = threads or (os.cpu_count() or 1)
    res = (2.5, 2.0, 1.5) if three_res else (2.5,)
    sess_t = [synth.make_session(s, kf, sensor, device=device, scene=scene, kf_spacing=spacing) for s in (1, 2)]
    torch.cuda.synchronize()
    ctx = capi.Context(vfov=50.0, hfov=360.0, device=0)
    loaded = []
    for S in sess_t:
        scans = ctx.preclean(ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5)
        loaded.append((scans, ctx.poses(S["poses"], S["inv"])))
    P = Params(gpu_use_self_removert=three_res, remove_resolution_list=list(res), num_nn_points_within=k, dist_nn_points_within=thr,
               downsample_voxel_size=voxel)
    t0 = time.perf_counter()
    ops = HipOps(ctx)
    rm = Removerter(ops, P, Session("Central", *loaded[0]), Session("Query", *loaded[1]), lane_ops=ops.lane() if lanes == 2 else None)
    rm.run()
    ctx.synchronize()
    t_gpu = time.perf_counter() - t0
    # the oracle sees exactly what the GPU saw after Step 0: the pre-cleaned scans, downloaded
    cpu = []
    for (scans, _), S in zip(loaded, sess_t):
        pts, off = scans.download()
        cpu.append(dict(scans=pts, offsets=off, poses=S["poses"], inv=S["inv"]))
    t0 = time.perf_counter()
    ref = orc.pipeline_run(orc.make_params(k=k, knn_thr=thr, voxel=voxel, use_self_removert=three_res, res_list=res, threads=threads), cpu[0], cpu[1])
    t_cpu = time.perf_counter() - t0
    report, bad = {}, 0
    for name in MAPS:
        want, got = ref.cloud(name), rm.outputs.get(name)
        if want is None or got is None:
            ok = want is None and got is None
            report[name] = {"points": None, "identical": ok}
        else:
            g = got.download()
            ok = g.shape == want.shape and bool((g.view(np.uint32) == want.view(np.uint32)).all())
            report[name] = {"points": int(len(want)), "identical": ok}
        bad += 0 if ok else 1
    for name, ss in rm.scan_outputs().items():
        w_pts, w_off = ref.scanset(name)
        g_pts, g_off = ss.download()
        ok = bool((g_off == w_off).all()) and g_pts.shape == w_pts.shape and bool((g_pts.view(np.uint32) == w_pts.view(np.uint32)).all())
        report[name] = {"points": int(len(w_pts)), "keyframes": int(len(w_off) - 1), "identical": ok}
        bad += 0 if ok else 1
    ref.free()
    ctx.close()
    try:
        commit = open(os.path.join(ROOT, ".commit_for_profiles")).read().strip()
    except OSError:
        commit = None
    return {"what": "GPU (C ABI) vs CPU oracle, every output of Removerter::run() compared bitwise",
            "config": f"BASELINE configs[{3 if config == 33 else config}]", "lanes": lanes, "workload": f"{scene} 2x{kf} {sensor} {'3-res' if three_res else 'single-res'} voxel {voxel} k {k} thr {thr}",
            "scan_points": [int(c["offsets"][-1]) for c in cpu], "gpu_run_s": round(t_gpu, 3), "oracle_run_s": round(t_cpu, 1),
            "oracle_threads": threads, "outputs_compared": len(report), "outputs_differing": bad,
            "product_sha": provenance.product_sha(), "kernels_sha": provenance.kernels_sha(), "oracle_sha": provenance.oracle_sha(), "commit": commit,
            "env": {k: v for k, v in os.environ.items() if k.startswith("LTM_")}, "outputs": report}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--kf", type=int, default=None)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--sessions", type=int, default=3, help="--config 2: length of the cascade 01 -> 02 -> ... (BASELINE: 6)")
    args = ap.parse_args()
    rep = run_parity(args.config, args.kf, args.threads, sessions=args.sessions)
    print(json.dumps(rep, indent=1))
    sys.exit(1 if rep["outputs_differing"] else 0)


if __name__ == "__main__":
    main()
