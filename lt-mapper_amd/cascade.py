"""Lifelong cascade driver (BASELINE.json configs[2]).

The reference leaves the lifelong loop to the user: after a run, point `central_sess_scan_dir` at the produced
`scans_updated/` (same poses) and run again against the next query session (README.md:115-118, doc/pipeline.png).
This module does that hand-over on the device: the central session of run j+1 is the `keyframe_scans_updated_` scan set of
run j with the central poses unchanged; only the new query session is uploaded between runs.
"""
from .removerter import Params, Removerter, Session


def run_cascade(ops, params: Params, central_scans, central_poses, queries):
    """queries: list of (scans, poses) handles of sessions 2..K.  Returns the list of Removerter objects (one per pair run);
    the live map after the last run is runs[-1].outputs['updated_map'], the live scans runs[-1].central_sess_.keyframe_scans_updated_."""
    runs = []
    for q_scans, q_poses in queries:
        rm = Removerter(ops, params, Session("Central", central_scans, central_poses), Session("Query", q_scans, q_poses))
        rm.run()
        runs.append(rm)
        central_scans = rm.central_sess_.keyframe_scans_updated_      # "scans_updated/" becomes the next central session
    return runs
