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
poses = ctx.poses(S["poses"], S["inv"])
cmap = ctx.voxel_centroid(ctx.merge_to_global(scans, poses), 0.05)
up = ctx.voxel_centroid_scanset(ctx.reproject(cmap, poses, 3.0), 0.05)        # like scans_updated: reprojected at 150 x 1080, per-keyframe octree grids
print("scans_updated-like set:", up.info(), flush=True)
for mode, threads in (("pcl", 64), ("pcl", 128), ("pcl", 256), ("input", 0), ("pcl", 192)):
    os.environ["LTM_VOXELGRID_ORDER"] = mode
    os.environ["LTM_VOXELGRID_THREADS"] = str(threads)
    for rep in range(3):
        ctx.synchronize(); t = time.perf_counter()
        out = ctx.voxel_grid_scanset(up, 0.05)
        ctx.synchronize(); dt = time.perf_counter() - t
        print(mode, threads, rep, "%.1f ms" % (1e3 * dt), out.info(), flush=True)
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
