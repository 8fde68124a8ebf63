"""Planned votes (include/ltm.h ltm_vote_plan_begin; csrc/ltm_k_vote_plan.inc): selfRemovert's sequence of full-map mode-0 votes (Removerter.cpp:1378-1393)
served from candidate lists built once for all resolutions must set, bit for bit, the labels of the un-planned kernels -- through the whole pipeline, for maps
that lost points, gained new ones (untracked), hold duplicate coordinates and signed zeros, and when the plan cannot be used (record-space overflow, budget)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sessions(n_kf=12, sensor="small"):
    from tools import synth
    return [synth.to_numpy(synth.make_session(s, n_kf, sensor)) for s in (1, 2)]


def _pipeline(sess, env):
    """run() with 3-res self-removert under the given environment (read at ltm_create): outputs + the plan counters"""
    from ltmapper_amd import capi
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = capi.Context()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    P = Params(gpu_use_self_removert=True, remove_resolution_list=[2.5, 2.0, 1.5])
    loaded = [(ctx.preclean(ctx.upload_scans(S["scans"], S["offsets"]), 2.5), ctx.poses(S["poses"], S["inv"])) for S in sess]
    rm = Removerter(HipOps(ctx), P, Session("Central", *loaded[0]), Session("Query", *loaded[1]))
    rm.run()
    out = {k: v.download() for k, v in rm.outputs.items() if v is not None}
    for k, v in rm.scan_outputs().items():
        out["scans:" + k], out["off:" + k] = v.download()
    stats = ctx.vote_plan_stats()
    del rm, loaded
    ctx.close()
    return out, stats


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        assert a[k].shape == b[k].shape, k
        assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all() if a[k].dtype == np.float32 else (a[k] == b[k]).all(), k


def test_pipeline_outputs_with_and_without_plans():
    sess = _sessions()
    plain, s0 = _pipeline(sess, {"LTM_VOTE_PLAN": "0"})
    planned, s1 = _pipeline(sess, {"LTM_VOTE_PLAN": "1"})
    assert s0["builds"] == 0 and s0["replays"] == 0
    assert s1["builds"] >= 2 and s1["replays"] >= 12 and s1["refused"] == 0, s1      # two sessions x six full-map votes
    _same(plain, planned)


def test_pipeline_when_the_plan_overflows_or_is_refused():
    sess = _sessions(8)
    plain, _ = _pipeline(sess, {"LTM_VOTE_PLAN": "0"})
    tiny, s = _pipeline(sess, {"LTM_VOTE_PLAN_FRACTION": "1e-9"})      # 8192 records per sub-stream: may or may not do; either way the outputs stand
    _same(plain, tiny)
    refused, s = _pipeline(sess, {"LTM_VOTE_PLAN_BUDGET_GB": "0"})
    assert s["refused"] >= 1 and s["replays"] == 0, s
    _same(plain, refused)


def _labels(ctx, cloud, scans, poses, alpha):
    kept, flagged, lab = ctx.visibility_partition(cloud, scans, poses, alpha, 0.1, 0, want_labels=True)
    del kept, flagged
    return lab


def test_votes_of_derived_maps_equal_the_unplanned_votes():
    """direct use of the entry points: one plan, then votes of (a) the map itself at every resolution, (b) a subset in another order, (c) the subset plus points the
    plan has never seen, duplicates of planned points and a signed zero; every label vector against the same vote without a plan"""
    from ltmapper_amd import capi
    S = _sessions(10)[0]
    rng = np.random.default_rng(3)
    ctx = capi.Context()
    scans = ctx.upload_scans(S["scans"], S["offsets"])
    poses = ctx.poses(S["poses"], S["inv"])
    m0 = ctx.voxel_centroid(ctx.merge_to_global(scans, poses), 0.05).download()
    assert len(m0) > 20000
    alphas = [2.5, 2.0, 1.5]
    sub = m0[np.sort(rng.choice(len(m0), len(m0) * 7 // 10, replace=False))]
    extra = sub[:4000].copy()
    extra[:, :3] += rng.normal(0, 0.03, (4000, 3)).astype(np.float32)      # new coordinates: untracked
    dup = sub[100:600].copy()                                                # coordinates the map already holds, at other indices
    zero = np.array([[-0.0, 0.0, -0.0, 1.0], [0.0, -0.0, 0.0, 2.0]], np.float32)
    mixed = np.concatenate([sub[::-1], extra, dup, zero]).astype(np.float32)
    maps = [m0, sub, mixed, rng.permutation(mixed)]
    want = [[_labels(ctx, ctx.upload(m), scans, poses, a) for a in alphas] for m in maps]
    assert ctx.vote_plan_stats()["replays"] == 0
    ctx.vote_plan_begin(scans, poses, alphas, 0.1)
    for m, w in zip(maps, want):
        for a, wl in zip(alphas, w):
            got = _labels(ctx, ctx.upload(m), scans, poses, a)
            assert (got == wl).all(), (len(m), a, int((got != wl).sum()))
    st = ctx.vote_plan_stats()
    assert st["builds"] == 1 and st["replays"] == len(maps) * len(alphas) and st["untracked_points"] > 0, st
    assert sum(int(w.sum()) for ww in want for w in ww) > 0
    # a resolution the plan does not cover, and a vote after the plan has ended, take the un-planned kernels
    ctx.vote_plan_end(scans)
    got = _labels(ctx, ctx.upload(sub), scans, poses, 2.0)
    assert (got == want[1][1]).all() and ctx.vote_plan_stats()["replays"] == 0
    ctx.close()
