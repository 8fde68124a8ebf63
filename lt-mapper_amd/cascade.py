"""Lifelong cascade driver (BASELINE.json configs[2]).

The reference leaves the lifelong loop to the user: after a run, point `central_sess_scan_dir` at the produced
`scans_updated/` (same poses) and run again against the next query session (README.md:115-118, doc/pipeline.png).  That next
run RE-LOADS the scans: `Session::loadKeyframes` puts every file through `pcl::VoxelGrid(downsample_voxel_size)`
(Session.cpp:284-289) and `Removerter::run` through `precleaningKeyframes(2.5)` (Removerter.cpp:1658-1660, Session.cpp:506-533)
before `makeGlobalMap`.  This module does the same hand-over on the device: the central session of run j+1 is
`ltm_preclean(ltm_voxel_grid_scanset(keyframe_scans_updated_ of run j, downsample_voxel_size), 2.5)` with the central poses
unchanged -- bit for bit what `ltm_run` loads from the files of run j (tests/test_gpu_cascade.py) -- and only the new query
session is uploaded between runs.
"""
from .removerter import Params, Removerter, Session

kPrecleanRadius = 2.5       # Removerter.cpp:1660


def reload_scans(ops, scans, params: Params):
    """what Session::loadKeyframes + precleaningKeyframes make of a scan set that was written to and read back from scans_updated/"""
    return ops.preclean(ops.voxel_grid_scanset(scans, params.downsample_voxel_size), kPrecleanRadius)


def run_cascade(ops, params: Params, central_scans, central_poses, queries, overlap=True, prepare_next=False, lane_ops=None):
    """central_scans / queries: (scans, poses) handles as a loader leaves them (VoxelGrid + pre-clean applied), sessions 1 and
    2..K.  Returns the list of Removerter objects (one per pair run); the live map after the last run is
    runs[-1].outputs['updated_map'], the live scans runs[-1].central_sess_.keyframe_scans_updated_.

    overlap (round 5): the re-load of scans_updated is STARTED when a run ends (ltm_voxel_grid_scanset_begin: keys to the host on the copy stream, std::sort's
    order on host threads) and FINISHED inside the next run, after that run's query session -- merge, grid, Step-1 remove / revert passes, none of which needs
    the central scans -- has been issued: the GPU works through the query session while the host sorts.  Same clouds either way (tests/test_gpu_cascade.py).

    prepare_next: also re-load the LAST run's scans_updated (-> runs[-1].next_central_scans) for a caller that will continue the cascade with further
    sessions; a cascade that ends here has no use for it (the reference re-loads scans_updated only when a next run is started on them), and round 4's
    bench step paid for that fifth, unused hand-over.

    lane_ops (round 6): every pair run takes the two-lane schedule (Removerter.run_two_lanes); the deferred hand-over ends on the main lane while the
    query session's chain is already running on the other."""
    queries = list(queries)
    runs = []
    pending = None
    can_overlap = overlap and getattr(ops, "supports_deferred_grid", False)      # one context, whole scan sets (HipOps; not the keyframe-sharded ops)
    for i_run, (q_scans, q_poses) in enumerate(queries):
        rm = Removerter(ops, params, Session("Central", central_scans, central_poses), Session("Query", q_scans, q_poses), lane_ops=lane_ops)
        if pending is not None:
            ticket, pending = pending, None
            rm.central_scans_future = lambda t=ticket: ops.preclean(ops.voxel_grid_scanset_end(t), kPrecleanRadius)
        rm.run()
        runs.append(rm)
        # "scans_updated/" re-loaded as the next central session -- if there is a next session (or the caller asked for the hand-over)
        last = i_run == len(queries) - 1      # by position: the same query handle may be replayed
        if last and not prepare_next:
            break
        if can_overlap:
            pending, central_scans = ops.voxel_grid_scanset_begin(rm.central_sess_.keyframe_scans_updated_, params.downsample_voxel_size), None
        else:
            central_scans = reload_scans(ops, rm.central_sess_.keyframe_scans_updated_, params)
    if prepare_next and runs:
        runs[-1].next_central_scans = ops.preclean(ops.voxel_grid_scanset_end(pending), kPrecleanRadius) if pending is not None else central_scans
    return runs
