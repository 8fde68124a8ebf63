#!/usr/bin/env python3
"""A/B timing of kernel variants inside ONE process on ONE box (box-to-box differences exceed the effects being measured).

    python tools/ab_kernels.py --env LTM_TILE_CULL=1 --env LTM_TILE_CULL=0 [--rounds 3] [--kf 500]

Builds the 2x500 `lot` central session once, then for every environment setting (applied before ltm_create, which reads the
A/B switches) runs the stages that dominate the step and prints the HIP-event time per kernel class:
  vote    3 x mode-0 visibility vote of the full map against all keyframes (res 2.5 / 2.0 / 1.5)
  reproj  1 x reprojection at res 3.0
  voxel   2 x 0.05 m voxel grid of the map
Variants are interleaved round by round."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE[,NAME=VALUE...] one variant per --env")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--kf", type=int, default=500)
    ap.add_argument("--sensor", default="os1-64")
    args = ap.parse_args()
    import numpy as np
    import torch
    import ltmapper_amd  # noqa: F401
    from ltmapper_amd import capi
    from tools import synth
    S = synth.make_session(1, args.kf, args.sensor, device="cuda:0")
    torch.cuda.synchronize()
    variants = args.env or [""]
    results = {v: {} for v in variants}
    for rnd in range(args.rounds):
        for v in variants:
            for kv in filter(None, v.split(",")):
                k, val = kv.split("=")
                os.environ[k] = val
            ctx = capi.Context(vfov=50.0, hfov=360.0, device=0)
            scans = ctx.preclean(ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5)
            poses = ctx.poses(S["poses"], S["inv"])
            cmap = ctx.voxel_centroid(ctx.merge_to_global(scans, poses), 0.05)
            labels = torch.zeros(len(cmap), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            for alpha in (2.5, 2.0, 1.5):       # warm-up: pool filled
                ctx.visibility_vote(cmap, scans, poses, 0, poses.n, alpha, 0.1, 0, labels.data_ptr())
            ctx.reproject(cmap, poses, 3.0)
            ctx.synchronize()
            ctx.clear_caches()                  # the timed votes build their scan images again (class vote_scan)
            ctx.profile_reset(); ctx.profile_enable(True)
            for alpha in (2.5, 2.0, 1.5):
                ctx.visibility_vote(cmap, scans, poses, 0, poses.n, alpha, 0.1, 0, labels.data_ptr())
            ctx.reproject(cmap, poses, 3.0)
            for _ in range(2):
                ctx.voxel_centroid(cmap, 0.05)
            ctx.synchronize()
            ctx.profile_enable(False)
            prof = ctx.profile_read()
            for cls, p in prof.items():
                results[v].setdefault(cls, []).append(p["ms"] / max(p["launches"], 1))
            for kv in filter(None, v.split(",")):
                os.environ.pop(kv.split("=")[0], None)
            n_map = len(cmap)
            del cmap, scans, poses
            ctx.close()
    print(f"map {n_map} points, {args.kf} keyframes; ms per launch, per round")
    for v in variants:
        print(f"[{v or 'default'}]")
        for cls in ("vote_map_cull", "vote_map_exact", "reproject_map", "vote_scan", "voxel", "vote_compare", "reproject_gather"):
            if cls in results[v]:
                xs = results[v][cls]
                print(f"  {cls:18s} " + " ".join(f"{x:8.3f}" for x in xs) + f"   min {min(xs):8.3f}")


if __name__ == "__main__":
    main()
