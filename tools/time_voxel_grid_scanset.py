import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import ltmapper_amd
from ltmapper_amd import capi
from tools import synth
S = synth.make_session(1, 500, "os1-64", device="cuda:0")
torch.cuda.synchronize()
ctx = capi.Context(vfov=50.0, hfov=360.0, device=0)
scans = ctx.preclean(ctx.scans_from_device(S["scans"].data_ptr(), S["offsets"].numpy().astype(np.uint64)), 2.5)
up = ctx.voxel_centroid_scanset(scans, 0.05)        # like scans_updated: per-keyframe octree grids
for mode in ("pcl", "input", "pcl"):
    os.environ["LTM_VOXELGRID_ORDER"] = mode
    for rep in range(3):
        ctx.synchronize(); t = time.perf_counter()
        out = ctx.voxel_grid_scanset(up, 0.05)
        ctx.synchronize(); dt = time.perf_counter() - t
        print(mode, rep, "%.1f ms" % (1e3 * dt), out.info(), flush=True)
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
