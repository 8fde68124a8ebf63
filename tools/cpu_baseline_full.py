#!/usr/bin/env python3
"""Full (unsampled) CPU-oracle run of a bench workload: the CPU baseline SURVEY.md 8(d) asks for.

    python tools/cpu_baseline_full.py --workload lot-2x500-os1-64-3res --threads 1 --runs 1 --out profiles/<name>.json

Runs makeGlobalMap + Removerter::run Steps 1-3 of the oracle (oracle/libltm_oracle.so, a port of the reference:
the reference itself needs ROS/PCL and cannot be built here) on the same synthetic sessions bench.py uses
(tools/synth.py, CPU-generated: the torch CPU and GPU generators draw different noise, the scene and sizes are the
same), every keyframe visited.  `threads` mirrors the reference's OpenMP sites (utility.cpp:109-110 over points,
Session.cpp:408 over keyframes): the oracle parallelises over keyframes.  Needs no GPU.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="lot-2x500-os1-64-3res")
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--runs", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    from oracle import oracle_py as orc
    from tools import synth
    sensor, n_kf, three_res, scene, spacing, voxel, knn_k, knn_thr = bench.WORKLOADS[args.workload]
    t0 = time.perf_counter()
    sess = [synth.to_numpy(synth.make_session(s, n_kf, sensor, scene=scene, kf_spacing=spacing)) for s in (1, 2)]
    t_gen = time.perf_counter() - t0
    for S in sess:                                       # precleaningKeyframes(2.5), Step 0
        pts, off = [], [0]
        for k in range(n_kf):
            p = orc.preclean(S["scans"][int(S["offsets"][k]):int(S["offsets"][k + 1])], 2.5)
            pts.append(p); off.append(off[-1] + len(p))
        S["scans"], S["offsets"] = np.concatenate(pts), np.array(off, dtype=np.uint64)
    P = orc.make_params(k=knn_k, knn_thr=knn_thr, voxel=voxel, use_self_removert=three_res,
                        res_list=(2.5, 2.0, 1.5) if three_res else (2.5,), threads=args.threads)
    runs = []
    maps = None
    for r in range(args.runs):
        t0 = time.perf_counter()
        res = orc.pipeline_run(P, sess[0], sess[1])
        wall = time.perf_counter() - t0
        tm = res.timings()
        maps = [len(res.cloud("OriginalNoisyCentralMapGlobal")), len(res.cloud("OriginalNoisyQueryMapGlobal"))]
        runs.append({"wall_s": round(wall, 2), "stage_s": {k: round(v, 2) for k, v in tm.items()}})
        res.free()
        print(f"run {r}: {wall:.1f} s", file=sys.stderr, flush=True)
    walls = sorted(x["wall_s"] for x in runs)
    med = walls[len(walls) // 2]
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {"what": "CPU oracle (port of the reference), full unsampled run: makeGlobalMap + Steps 1-3, inputs in memory",
           "workload": args.workload, "keyframes_per_session": n_kf, "threads": args.threads, "nproc": os.cpu_count(),
           "cpu": cpu_model(), "scan_points": [int(S["offsets"][-1]) for S in sess], "map_points": maps,
           "runs": runs, "median_wall_s": med, "keyframe_pairs_per_s": round(n_kf / med, 5), "synth_generation_s": round(t_gen, 1),
           "commit": head, "date": time.strftime("%Y-%m-%d")}
    s = json.dumps(out, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
