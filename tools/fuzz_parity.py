#!/usr/bin/env python3
"""Randomised end-to-end parity: N small session pairs with randomly drawn parameters -- field of view, resolutions, kNN k / threshold,
voxel size, a random LiDAR->base extrinsic, keyframe batch size, scene and sensor -- every output of Removerter::run() on the GPU (through
the C ABI) against the CPU oracle, bitwise.  Complements the fixed-parameter tests: the kernels have parameter-dependent paths (fitted /
generic elevation polynomial, fast / plain divisions after the create-time self-check, packed / pair voxel sort, kNN specialisations for
k <= 4 and the generic form, identity / non-identity extrinsic).  TEST INFRASTRUCTURE.

    python tools/fuzz_parity.py [--n 12] [--seed 1] > profiles/<name>.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 32))
    ap.add_argument("--sensors", default="tiny,small", help="comma-separated tools.synth sensor names to draw from")
    ap.add_argument("--kf", type=int, nargs=2, default=[4, 8], metavar=("MIN", "MAX"), help="keyframes per session, drawn uniformly")
    ap.add_argument("--se3", action="store_true", help="full SE(3) keyframe poses: roll / pitch N(0, 1..3 deg) and z drift per keyframe (the rays are cast from "
                                                       "that attitude), session origin 10-50 km from the coordinate origin in half of the cases")
    args = ap.parse_args()
    import numpy as np
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    from oracle import oracle_py as orc
    from tools import synth
    from tools.parity_fullsize import _compare_run
    from test_gpu_pipeline import MAPS
    rng = np.random.default_rng(args.seed)
    cases, bad_total = [], 0
    occl = {"pairs": 0, "near": 0, "projected": 0}
    for it in range(args.n):
        vfov = float(rng.choice([50.0, 40.0, 30.0, 60.0, 90.0]))
        hfov = float(rng.choice([360.0, 360.0, 180.0]))
        three = bool(rng.integers(2))
        res = sorted({float(r) for r in rng.choice([2.5, 2.0, 1.5, 1.0, 3.0], size=int(rng.integers(1, 4)), replace=False)}, reverse=True) if three else [2.5]
        k = int(rng.choice([1, 2, 2, 3, 4, 5]))
        thr = float(rng.choice([0.01, 0.003, 0.04, 0.1]))
        voxel = float(rng.choice([0.05, 0.05, 0.1, 0.2]))
        scene = str(rng.choice(["lot", "street"]))
        sensor = str(rng.choice(args.sensors.split(",")))
        n_kf = int(rng.integers(args.kf[0], args.kf[1] + 1))
        l2b = None
        if rng.integers(3) == 0:
            from scipy.spatial.transform import Rotation
            l2b = np.eye(4)
            l2b[:3, :3] = Rotation.from_euler("xyz", rng.normal(0, 0.2, 3)).as_matrix()
            l2b[:3, 3] = rng.normal(0, 0.3, 3)
        batch = int(rng.choice([0, 0, 3]))
        desc = dict(vfov=vfov, hfov=hfov, res=res if three else None, k=k, thr=thr, voxel=voxel, scene=scene, sensor=sensor, n_kf=n_kf,
                    extrinsic=None if l2b is None else "random", max_kf_batch=batch)
        pose_kw = {}
        if args.se3:
            pose_kw = dict(tilt_deg=float(rng.choice([1.0, 3.0])), z_drift=float(rng.choice([0.05, 0.3])),
                           origin=tuple(float(v) for v in (rng.uniform(1e4, 5e4, 3) * rng.choice([-1, 1], 3) * [1, 1, 0.005])) if rng.integers(2) else (0.0, 0.0, 0.0))
            desc.update(pose_kw)
        C, Q = (synth.to_numpy(synth.make_session(s, n_kf, sensor, scene=scene, seed=synth.MASTER_SEED + it, **pose_kw)) for s in (1, 2))
        if args.se3:      # the inverse poses as the reference computes them (Session.cpp:110): the oracle's restatement of Eigen's inverse
            for S in (C, Q):
                S["inv"] = orc.inverse_poses(S["poses"])
        t0 = time.perf_counter()
        ref = orc.pipeline_run(orc.make_params(vfov=vfov, hfov=hfov, k=k, knn_thr=thr, voxel=voxel, lidar2base=l2b, use_self_removert=three,
                                               res_list=tuple(res), threads=args.threads), C, Q)
        ctx = capi.Context(vfov=vfov, hfov=hfov, lidar2base=l2b, device=0, max_kf_batch=batch)
        P = Params(sequence_vfov=vfov, sequence_hfov=hfov, gpu_use_self_removert=three, remove_resolution_list=list(res), num_nn_points_within=k,
                   dist_nn_points_within=thr, downsample_voxel_size=voxel, ExtrinsicLiDARtoPoseBase=l2b)
        sessions = [Session(n, ctx.upload_scans(S["scans"], S["offsets"]), ctx.poses(S["poses"], S["inv"])) for n, S in (("Central", C), ("Query", Q))]
        rm = Removerter(HipOps(ctx), P, *sessions)
        rm.run()
        report = {}
        bad = _compare_run(np, rm, ref, MAPS, report)
        sel, fast = ctx.selfcheck() if hasattr(ctx, "selfcheck") else (None, None)
        st = ctx.occlusion_stats()       # (tile, keyframe) pairs that went through the occlusion-culled launch, and how many were projected
        for key, v in zip(("pairs", "near", "projected"), st): occl[key] += int(v)
        ref.free(); ctx.close()
        bad_total += bad
        cases.append(dict(desc, outputs_compared=len(report), outputs_differing=bad, differing=[n for n, v in report.items() if not v["identical"]],
                          points=sum(v["points"] or 0 for v in report.values()), fast_math=fast, seconds=round(time.perf_counter() - t0, 1)))
        print(f"case {it}: {desc} -> {bad} of {len(report)} outputs differ", file=sys.stderr)
    from tools import provenance
    print(json.dumps({"what": "randomised end-to-end parity, GPU (C ABI) vs CPU oracle, bitwise", "seed": args.seed, "cases": len(cases),
                      "cases_with_differences": sum(1 for c in cases if c["outputs_differing"]), "product_sha": provenance.product_sha(),
                      "oracle_sha": provenance.oracle_sha(), "sensors": args.sensors, "keyframes": args.kf,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("LTM_")}, "occlusion": occl, "detail": cases}, indent=1))
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
