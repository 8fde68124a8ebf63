"""per-pass timing of one session's selfRemovert chain: map size, vote time, survivors (round-6 experiment)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ltmapper_amd
from ltmapper_amd import capi
from tools import synth
S = synth.make_session(1, 500, "os1-64", device="cuda")
ctx = capi.Context()
scans = ctx.preclean(ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5)
poses = ctx.poses(S["poses"], S["inv"])
cur = ctx.voxel_centroid(ctx.merge_to_global(scans, poses), 0.05)
dyn = None
def vote(m, alpha, tag):
    ctx.synchronize(); ctx.cull_stats()
    t0 = time.perf_counter()
    k, f = ctx.visibility_partition(m, scans, poses, alpha, 0.1, 0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    surv, pts = ctx.cull_stats()
    print(f"{tag:28s} map {len(m):8d} pts  {1e3*dt:6.2f} ms  flagged {len(f):7d}  phase-1 survivors {100.0*surv/max(pts,1):5.1f} %", flush=True)
    return k, f
for rep in range(2):
    cur2, dyn2 = cur, None
    for res in (2.5, 2.0, 1.5):
        k, f = vote(cur2, res, f"remove {res}")
        st, dy = ctx.voxel_centroid_batch([k, f if dyn2 is None else ctx.concat([dyn2, f])], [0.05, 0.05])
        k, f = vote(dy, float(np.float32(0.95 * res)), f"revert {0.95*res:.3f} (dynamic map)")
        dy, st = ctx.voxel_centroid_batch([f, ctx.concat([st, k])], [0.05, 0.05])
        k, f = vote(st, res, f"remove {res} again")
        cur2, dyn2 = ctx.voxel_centroid_batch([k, ctx.concat([dy, f])], [0.05, 0.05])
    print("--")
