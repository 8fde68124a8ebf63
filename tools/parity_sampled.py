#!/usr/bin/env python3
"""Parity at the STATED size of the BASELINE configurations the oracle cannot run in full.

configs[3] is 2 x 2000 keyframes of a 64-beam sensor over a 45 M-point map, configs[4] 2 x 200 keyframes of 1 M rays: the oracle's whole
pipeline would take hours.  What CAN be checked at that size in minutes: the per-keyframe primitives are independent per keyframe, so

  1. both session maps are built on the GPU at full size (makeGlobalMap of all keyframes: merge + 0.05 m octree grid) -- and compared with
     the oracle's merge + grid of the same scans (one pass over 2 x 220 M points; the sort-based oracle grid takes about a minute each);
  2. K keyframes are drawn at random (seeded) from the full session; for them, against the FULL-SIZE maps:
       ltm_visibility_vote   mode 0 (scan - map, k_vote_map_cull with its whole-tile range cull) at every resolution of the 3-res list
                             and mode 1 (map - scan, the exact-image kernel; the occlusion cull forced on AND off)   -> labels, exact
       ltm_reproject         (occlusion cull forced on AND off)                                                       -> points, bitwise
       ltm_knn_partition     of the drawn keyframes' reprojected scans against the OTHER session's full map          -> coexist / diff, bitwise
     each compared with the oracle primitive on the same inputs (32 keyframes x 45 M points = 1.4e9 projections per vote: seconds on the
     GPU box's 256 host threads, the keyframes being independent).

    python tools/parity_sampled.py --config 3 [--kf 2000] [--sample 32] > profiles/<name>.json

TEST INFRASTRUCTURE: the oracle is the checker here, never the thing measured."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {3: ("street", "hdl-64e", 2000, 1.0, 0.05, 2, 0.01), 4: ("street", "mls", 200, 2.0, 0.1, 2, 0.04)}


def _ctx(capi, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return capi.Context(vfov=50.0, hfov=360.0, device=0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def run(config, kf=None, sample=32, threads=None, seed=7, device="cuda:0", check_maps=True):
    import numpy as np
    import torch
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from oracle import oracle_py as orc
    from tools import provenance, synth
    scene, sensor, kf_default, spacing, voxel, k, thr = CONFIGS[config]
    kf = kf or kf_default
    threads = threads or (os.cpu_count() or 1)
    rng = np.random.default_rng(seed)
    I4 = np.eye(4)
    report, bad = {}, 0

    def note(name, ok, **kw):
        nonlocal bad
        report[name] = dict(identical=bool(ok), **kw)
        bad += 0 if ok else 1

    sess = [synth.make_session(s, kf, sensor, device=device, scene=scene, kf_spacing=spacing) for s in (1, 2)]
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    timings = {}
    host = []          # per session: full pre-cleaned scans on the host (for the oracle), poses
    maps_host = []
    for tag, S in zip(("central", "query"), sess):
        ctx = _ctx(capi)
        scans = ctx.preclean(ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5)
        poses = ctx.poses(S["poses"], S["inv"])
        t0 = time.perf_counter()
        cmap = ctx.voxel_centroid(ctx.merge_to_global(scans, poses), voxel)
        ctx.synchronize()
        timings[f"{tag}_gpu_make_global_map_s"] = round(time.perf_counter() - t0, 3)
        pts, off = scans.download()
        m = cmap.download()
        host.append(dict(scans=pts, offsets=off, poses=np.asarray(S["poses"], np.float64).reshape(-1, 16), inv=np.asarray(S["inv"], np.float64).reshape(-1, 16)))
        maps_host.append(m)
        if check_maps:
            t0 = time.perf_counter()
            want = orc.voxel_centroid(orc.merge_to_global(pts, off, host[-1]["poses"], I4), voxel)
            timings[f"{tag}_oracle_make_global_map_s"] = round(time.perf_counter() - t0, 1)
            note(f"{tag}_map_global_curr", want.shape == m.shape and (want.view(np.uint32) == m.view(np.uint32)).all(), points_in=int(off[-1]), points=int(len(m)))
            del want
        ctx.close()
    # ---- the drawn keyframes
    for si, tag in ((0, "central"), (1, "query")):
        H, M = host[si], maps_host[si]
        other = maps_host[1 - si]
        n_kf = len(H["offsets"]) - 1
        pick = np.sort(rng.choice(n_kf, size=min(sample, n_kf), replace=False))
        sub_pts = np.concatenate([H["scans"][int(H["offsets"][j]):int(H["offsets"][j + 1])] for j in pick])
        sub_off = np.cumsum([0] + [int(H["offsets"][j + 1] - H["offsets"][j]) for j in pick]).astype(np.uint64)
        sub_poses, sub_inv = H["poses"][pick].copy(), H["inv"][pick].copy()
        report[f"{tag}_keyframes_drawn"] = [int(j) for j in pick]
        for occl in (1, 0):
            ctx = _ctx(capi, LTM_OCCLUSION=occl, LTM_OCCLUSION_MIN_PAIRS=0)
            g_map = ctx.upload(M)
            g_scans = ctx.upload_scans(sub_pts, sub_off)
            g_poses = ctx.poses(sub_poses, sub_inv)
            if occl:      # mode 0 does not go through the exact-image kernel: once is enough
                for alpha in (2.5, 2.375, 2.0, 1.9, 1.5, 1.425):
                    t0 = time.perf_counter()
                    lab = ctx.visibility_partition(g_map, g_scans, g_poses, alpha, 0.1, 0, want_labels=True)[2]
                    t_g = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    want = orc.vote_labels(M, sub_pts, sub_off, sub_inv, I4, 50.0, 360.0, alpha, 0.1, 0, threads=threads)
                    note(f"{tag}_vote_mode0_res{alpha}", (lab == want).all(), flagged=int(want.sum()), map_points=int(len(M)), gpu_s=round(t_g, 3),
                         oracle_s=round(time.perf_counter() - t0, 1))
            # reprojection (exact-image kernel), then mode 1 with the reprojected scans as the source -- as filterStrongND does
            t0 = time.perf_counter()
            rp = ctx.reproject(g_map, g_poses, 3.0)
            rp_pts, rp_off = rp.download()
            t_g = time.perf_counter() - t0
            if occl:
                t0 = time.perf_counter()
                want_pts, want_off = orc.reproject(M, sub_inv, I4, 50.0, 360.0, 3.0, threads=threads)
                t_o = time.perf_counter() - t0
            note(f"{tag}_reproject_occlusion{occl}", (rp_off == want_off).all() and rp_pts.shape == want_pts.shape and (rp_pts.view(np.uint32) == want_pts.view(np.uint32)).all(),
                 points=int(len(want_pts)), gpu_s=round(t_g, 3), oracle_s=round(t_o, 1), pairs_stats=list(ctx.occlusion_stats(reset=True)) if occl else None)
            lab1 = ctx.visibility_partition(g_map, ctx.upload_scans(want_pts, want_off), g_poses, 2.5, 0.1, 1, want_labels=True)[2]
            if occl:
                want1 = orc.vote_labels(M, want_pts, want_off, sub_inv, I4, 50.0, 360.0, 2.5, 0.1, 1, threads=threads)
            note(f"{tag}_vote_mode1_occlusion{occl}", (lab1 == want1).all(), flagged=int(want1.sum()))
            if occl:
                # kNN of the drawn keyframes' reprojected scans against the OTHER session's full map (extractLowDynPointsViaKnnDiff)
                t0 = time.perf_counter()
                co, di = ctx.knn_partition(ctx.upload(other), ctx.upload_scans(want_pts, want_off), g_poses, k, thr)
                (co_p, co_o), (di_p, di_o) = co.download(), di.download()
                t_g = time.perf_counter() - t0
                t0 = time.perf_counter()
                flag, loc = orc.knn_labels(other, want_pts, want_off, sub_poses, sub_inv, I4, k, thr, threads=threads)
                t_o = time.perf_counter() - t0
                w_co, w_di = loc[flag == 1], loc[flag == 0]
                ok = co_p.shape == w_co.shape and di_p.shape == w_di.shape and (co_p.view(np.uint32) == w_co.view(np.uint32)).all() and (di_p.view(np.uint32) == w_di.view(np.uint32)).all()
                note(f"{tag}_knn_partition_vs_other_map", ok, queries=int(len(want_pts)), coexist=int(flag.sum()), target_points=int(len(other)), gpu_s=round(t_g, 3), oracle_s=round(t_o, 1))
            ctx.close()
    try:
        commit = open(os.path.join(ROOT, ".commit_for_profiles")).read().strip()
    except OSError:
        commit = None
    return {"what": "GPU (C ABI) vs CPU oracle at the configuration's STATED size: full-size session maps, and the per-keyframe primitives (vote mode 0 at six "
                    "resolutions, reprojection and mode-1 vote with the occlusion cull forced on and off, kNN partition against the other session's map) for "
                    f"{sample} randomly drawn keyframes per session, all bitwise / exact",
            "config": f"BASELINE configs[{config}]", "workload": f"{scene} 2x{kf} {sensor} voxel {voxel} k {k} thr {thr}", "keyframes_sampled_per_session": sample,
            "map_points": [int(len(m)) for m in maps_host], "scan_points": [int(h["offsets"][-1]) for h in host], "oracle_threads": threads,
            "checks": len([v for v in report.values() if isinstance(v, dict)]), "checks_failed": bad, "wall_s": round(time.perf_counter() - t_all, 1), "timings": timings,
            "product_sha": provenance.product_sha(), "kernels_sha": provenance.kernels_sha(), "oracle_sha": provenance.oracle_sha(), "commit": commit, "results": report}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--kf", type=int, default=None)
    ap.add_argument("--sample", type=int, default=32)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--no-map-check", action="store_true")
    args = ap.parse_args()
    rep = run(args.config, args.kf, args.sample, args.threads, check_maps=not args.no_map_check)
    print(json.dumps(rep, indent=1))
    sys.exit(1 if rep["checks_failed"] else 0)


if __name__ == "__main__":
    main()
