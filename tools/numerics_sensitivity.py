#!/usr/bin/env python3
"""What can the UNPINNED numerics move?  (VERDICT r2 "next" item 8.)

The reference cannot be built here, so two pieces of third-party arithmetic on the hot path are restated from knowledge:
Eigen 3.3.7 `Matrix4d::inverse()` (Session.cpp:109-110) and glibc `atan2f` (utility.cpp:46-47; pinned exhaustively against the
host libm, but the reference's glibc 2.31 build is not here).  This tool runs the CPU oracle (TEST INFRASTRUCTURE; the product is
not involved) on one session pair with the inverse poses computed four ways -- Eigen-ordered block form (the shipped one),
cofactor expansion, Gauss-Jordan, numpy/LAPACK -- and with a fraction of all atan2f results moved by +-1 ulp, and reports per
output how many points appear in one result and not in the other (rows compared bitwise as sets) and how far the nearest
counterpart of such a point is.  That is the error bar on "matches the reference" while oracle/_ref stays impossible.

    python tools/numerics_sensitivity.py [--sensor os1-64 --kf 50] [--threads N] > profiles/r3_numerics_sensitivity_<...>.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402

MAPS = ["OriginalNoisyCentralMapGlobal", "OriginalNoisyQueryMapGlobal", "central_map_static", "central_map_dynamic", "query_map_static",
        "query_map_dynamic", "central_sess_high_dyn", "query_sess_high_dyn", "union_map_queryside", "union_map_centralside", "pd_map", "nd_map",
        "strong_nd_map", "weak_nd_map", "strong_pd_map", "weak_pd_map", "updated_map", "updated_map_strong"]
SCANS = ["scans_updated", "scans_updated_strong", "scans_pd", "scans_pd_strong", "scans_nd_strong"]


def rows_as_keys(a):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
    return a.view([("k", "V16")]).reshape(-1)


def set_difference(a, b):
    """points (bitwise rows) of a not in b, and of b not in a"""
    ka, kb = rows_as_keys(a), rows_as_keys(b)
    only_a = a[~np.isin(ka, kb)] if len(a) else a
    only_b = b[~np.isin(kb, ka)] if len(b) else b
    return only_a, only_b


def nearest_dist(p, q):
    """distance from every point of p to its nearest point of q (xyz)"""
    if len(p) == 0 or len(q) == 0:
        return np.zeros(0)
    from scipy.spatial import cKDTree
    return cKDTree(q[:, :3].astype(np.float64)).query(p[:, :3].astype(np.float64))[0]


def compare(base, other):
    rep, tot_pts, tot_diff = {}, 0, 0
    for name in MAPS + SCANS:
        if name in MAPS:
            a, b = base.cloud(name), other.cloud(name)
        else:
            a, b = base.scanset(name)[0], other.scanset(name)[0]
        if a is None or b is None:
            rep[name] = {"points": None if a is None else int(len(a)), "other_points": None if b is None else int(len(b))}
            continue
        oa, ob = set_difference(a, b)
        d = nearest_dist(oa, b)
        e = {"points": int(len(a)), "count_delta": int(len(b)) - int(len(a)), "only_in_base": int(len(oa)), "only_in_variant": int(len(ob))}
        if len(d):
            # a row that differs only in the last bits of xyz has a counterpart within float rounding; a genuinely flipped label
            # (point present in one result only) has its nearest counterpart a voxel or more away
            e["moved_within_1e-4_m"] = int((d <= 1e-4).sum())
            e["present_in_one_only"] = int((d > 1e-4).sum())
            e["max_nearest_m"] = float(d.max())
        rep[name] = e
        tot_pts += len(a); tot_diff += len(oa)
    return rep, tot_pts, tot_diff


def run_experiment(orc, C, Q, threads, three_res=False, quick=False):
    res = (2.5, 2.0, 1.5)
    P = orc.make_params(k=2, knn_thr=0.01, use_self_removert=three_res, res_list=res if three_res else (2.5,), threads=threads)

    def run(inv_variant=0, ppm=0, numpy_inv=False):
        c, q = dict(C), dict(Q)
        for S in (c, q):
            S["inv"] = (np.array([np.linalg.inv(m.reshape(4, 4)).reshape(16) for m in S["poses"].reshape(-1, 16)]) if numpy_inv
                        else orc.inverse_poses(S["poses"], inv_variant))
        orc.set_atan2f_perturbation(ppm, 20250224)
        try:
            return orc.pipeline_run(P, c, q)
        finally:
            orc.set_atan2f_perturbation(0)

    t0 = time.perf_counter()
    base = run()
    t_base = time.perf_counter() - t0
    variants = [("inverse_cofactor", dict(inv_variant=1)), ("inverse_gauss_jordan", dict(inv_variant=2)), ("inverse_numpy_lapack", dict(numpy_inv=True)),
                ("atan2f_1ulp_every_call", dict(ppm=1000000)), ("atan2f_1ulp_1_in_1000", dict(ppm=1000))]
    if quick:
        variants = [variants[0], variants[3]]
    out = {}
    inv0 = orc.inverse_poses(C["poses"])
    for name, kw in variants:
        other = run(**kw)
        rep, tot, diff = compare(base, other)
        other.free()
        e = {"outputs": rep, "points_compared": int(tot), "points_differing": int(diff), "fraction": diff / max(tot, 1)}
        if "inv_variant" in kw or kw.get("numpy_inv"):
            iv = (np.array([np.linalg.inv(m.reshape(4, 4)).reshape(16) for m in C["poses"].reshape(-1, 16)]) if kw.get("numpy_inv")
                  else orc.inverse_poses(C["poses"], kw["inv_variant"]))
            e["inverse_max_abs_deviation_from_shipped"] = float(np.abs(iv - inv0).max())
            e["inverse_entries_differing_bitwise"] = int((iv != inv0).sum())
        out[name] = e
    base.free()
    return out, t_base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sensor", default="os1-64")
    ap.add_argument("--kf", type=int, default=50)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--three-res", action="store_true")
    args = ap.parse_args()
    import torch
    from oracle import oracle_py as orc
    from tools import synth
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    sess = [synth.to_numpy(synth.make_session(s, args.kf, args.sensor, device=dev)) for s in (1, 2)]
    for S in sess:       # Step 0: pre-clean (Removerter.cpp:1660)
        pts, off = [], [0]
        for k in range(len(S["offsets"]) - 1):
            p = orc.preclean(S["scans"][int(S["offsets"][k]):int(S["offsets"][k + 1])], 2.5)
            pts.append(p); off.append(off[-1] + len(p))
        S["scans"], S["offsets"] = np.concatenate(pts), np.array(off, np.uint64)
    out, t_base = run_experiment(orc, sess[0], sess[1], args.threads, args.three_res)
    print(json.dumps({"what": "CPU oracle with the unpinned numerics varied: points present in the baseline result and not in the variant, per output",
                      "workload": f"lot 2x{args.kf} {args.sensor} {'3-res' if args.three_res else 'single-res'}", "oracle_threads": args.threads,
                      "baseline_run_s": round(t_base, 1), "baseline": "inverse poses by the Eigen-ordered block form (orc_inverse4x4), glibc-algorithm atan2f",
                      "variants": out}, indent=1))


if __name__ == "__main__":
    main()
