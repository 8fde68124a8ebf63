"""CPU fuzz of the oracle against the REFERENCE-COMPILED build (oracle/_ref): seeded random pair runs -- sensor, scene,
keyframe count and spacing, single-res / 3-res, kNN parameters, map voxel size, SE(3) keyframe poses with a far session origin, a random
LiDAR -> base extrinsic -- through Removerter::run() of the reference's own sources and through oracle/ltm_oracle.cpp, every saved cloud
(16 maps, 5 scan directories) and the session state compared BITWISE.  No GPU involved; needs /root/reference (build container only).

  python tools/fuzz_ref_vs_oracle.py --n 60 --jobs 8 --out profiles/r4_fuzz_ref_vs_oracle.json
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAPS = ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_sess_high_dyn", "query_sess_high_dyn", "union_map_queryside",
        "union_map_centralside", "pd_map", "nd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map", "updated_map", "updated_map_strong"]
SCANS = ["scans_updated", "scans_updated_strong", "scans_pd", "scans_pd_strong", "scans_nd_strong"]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4).view(np.uint32)


def draw(seed, real_sensors=False):
    rng = np.random.default_rng(seed)
    three = bool(rng.integers(0, 2))
    if real_sensors:      # the BASELINE sensors at their full resolution, few keyframes (the reference-compiled build is serial)
        case = dict(seed=int(seed), sensor=str(rng.choice(["os1-64", "hdl-64e"])), scene=str(rng.choice(["lot", "street"])), n_kf=int(rng.integers(3, 7)),
                    spacing=float(rng.choice([1.0, 2.0, 4.0])), three=three, k=int(rng.integers(1, 4)), thr=float(rng.choice([0.01, 0.02, 0.1])),
                    voxel=float(rng.choice([0.05, 0.1])), tilt_deg=float(rng.choice([0.0, 3.0])), z_drift=float(rng.choice([0.0, 0.05])),
                    origin=[float(v) for v in (rng.choice([0.0, 1.0]) * rng.uniform(-5e4, 5e4, 3) * [1, 1, 0.002])], extrinsic=bool(rng.integers(0, 2)))
        return case
    case = dict(seed=int(seed), sensor=str(rng.choice(["tiny", "tiny", "small"])), scene=str(rng.choice(["lot", "street"])),
                n_kf=int(rng.integers(3, 9) if three else rng.integers(3, 14)), spacing=float(rng.choice([0.5, 1.0, 2.0, 4.0])), three=three,
                k=int(rng.integers(1, 5)), thr=float(rng.choice([0.005, 0.01, 0.02, 0.05, 0.1, 0.25])), voxel=float(rng.choice([0.05, 0.05, 0.1, 0.2])),
                tilt_deg=float(rng.choice([0.0, 1.0, 3.0])), z_drift=float(rng.choice([0.0, 0.05])),
                origin=[float(v) for v in (rng.choice([0.0, 1.0]) * rng.uniform(-5e4, 5e4, 3) * [1, 1, 0.002])], extrinsic=bool(rng.integers(0, 2)))
    if case["sensor"] == "small":
        case["n_kf"] = min(case["n_kf"], 6)
    return case


def run_case(case):
    from oracle import oracle_py as orc
    from oracle import ref_py
    from tools import synth
    t0 = time.time()
    rng = np.random.default_rng(case["seed"] + 1)
    mk = lambda s: synth.to_numpy(synth.make_session(s, case["n_kf"], case["sensor"], scene=case["scene"], kf_spacing=case["spacing"],   # noqa: E731
                                                     tilt_deg=case["tilt_deg"], z_drift=case["z_drift"], origin=tuple(case["origin"])))
    C, Q = mk(1), mk(2)
    for S in (C, Q):
        S["inv"] = orc.inverse_poses(S["poses"])
    ext = np.eye(4)
    if case["extrinsic"]:
        a = rng.normal(0, np.deg2rad(10.0), 3)
        cz, sz, cy, sy, cx, sx = np.cos(a[2]), np.sin(a[2]), np.cos(a[1]), np.sin(a[1]), np.cos(a[0]), np.sin(a[0])
        ext[:3, :3] = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
        ext[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    res = (2.5, 2.0, 1.5) if case["three"] else (2.5,)
    kw = dict(k=case["k"], knn_thr=case["thr"], voxel=case["voxel"], lidar2base=ext, use_self_removert=case["three"], res_list=res,
              vfov=26.9 if case["sensor"] == "hdl-64e" else 50.0)        # the yaml's sequence_vfov for that sensor (config/params_ltmapper.yaml)
    R = ref_py.Removerter(ref_py.make_params(**kw)).pipeline_run(C, Q)
    O = orc.pipeline_run(orc.make_params(**kw), C, Q)
    bad, n_pts, n_out = [], 0, 0

    def cmp(name, a, b):
        nonlocal n_pts, n_out
        if (a is None) != (b is None):
            bad.append(f"{name}: saved by one side only"); return
        if a is None:
            return
        n_out += 1
        if len(a) != len(b) or not (_bits(a) == _bits(b)).all():
            bad.append(f"{name}: {len(a)} vs {len(b)} points" if len(a) != len(b) else f"{name}: {int((_bits(a) != _bits(b)).any(1).sum())} of {len(a)} points differ")
        n_pts += len(a)
    for m in MAPS:
        cmp(m, R.cloud(m), O.cloud(m))
    for s in SCANS:
        (a, ao), (b, bo) = R.scanset(s), O.scanset(s)
        if not (np.asarray(ao) == np.asarray(bo)).all():
            bad.append(f"{s}: per-keyframe counts differ")
        cmp(s, a, b)
    for q, tag in ((0, "central"), (1, "query")):
        cmp(f"{tag} static", R.session_map(q, "static"), O.cloud(f"{tag}_map_static"))
        cmp(f"{tag} dynamic", R.session_map(q, "dynamic"), O.cloud(f"{tag}_map_dynamic"))
        for which, name in (("static_projected", f"{tag}_static_projected"), ("knn_coexist", f"{tag}_knn_coexist"), ("knn_diff", f"{tag}_knn_diff")):
            (a, ao), (b, bo) = R.session_scans(q, which, case["n_kf"]), O.scanset(name)
            if not (np.asarray(ao) == np.asarray(bo)).all():
                bad.append(f"{name}: per-keyframe counts differ")
            cmp(name, a, b)
    sizes = {m: (0 if O.cloud(m) is None else int(len(O.cloud(m)))) for m in ("central_sess_high_dyn", "nd_map", "pd_map", "strong_nd_map", "weak_nd_map", "strong_pd_map", "updated_map")}
    R.close()
    O.free()
    return dict(case=case, ok=not bad, differences=bad, outputs_compared=n_out, points_compared=n_pts, sizes=sizes, seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--seed", type=int, default=20250926)
    ap.add_argument("--jobs", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--out", default=None)
    ap.add_argument("--real-sensors", action="store_true", help="os1-64 / hdl-64e at full resolution, 3-6 keyframes per session")
    args = ap.parse_args()
    from oracle import ref_py
    if not ref_py.available():
        ref_py.build()
    cases = [draw(args.seed + i, args.real_sensors) for i in range(args.n)]
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=args.jobs) as ex:
        results = list(ex.map(run_case, cases))
    from tools import provenance
    rec = {"what": "oracle/ltm_oracle.cpp against the reference-compiled build oracle/_ref/libltm_ref.so (the reference's unmodified sources over stand-in headers): "
                   "Removerter::run() from makeGlobalMap on, every saved cloud and the kept session state, bitwise", "n_cases": len(results),
           "n_ok": sum(r["ok"] for r in results), "outputs_compared": sum(r["outputs_compared"] for r in results), "points_compared": sum(r["points_compared"] for r in results),
           "oracle_sha": provenance.oracle_sha(), "seed": args.seed, "wall_s": round(time.time() - t0, 1), "cases": results}
    print(json.dumps({k: v for k, v in rec.items() if k != "cases"}))
    for r in results:
        if not r["ok"]:
            print("FAILED", json.dumps(r["case"]), r["differences"][:4])
    if args.out:
        json.dump(rec, open(args.out, "w"), indent=1)
    return 0 if rec["n_ok"] == rec["n_cases"] else 1


if __name__ == "__main__":
    sys.exit(main())
