"""Two lanes (include/ltm.h "lanes"; removerter.Removerter.run_two_lanes): the independent chains of run() side by side on a context and its lane give,
bit for bit, the clouds of the one-lane order; lend / give / fence / events behave as the header says."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sessions(n_kf=12, sensor="small"):
    from tools import synth
    return [synth.to_numpy(synth.make_session(s, n_kf, sensor)) for s in (1, 2)]


def _run(ctx, sess, three_res, lanes):
    from ltmapper_amd.removerter import HipOps, Params, Removerter, Session
    P = Params(gpu_use_self_removert=three_res, remove_resolution_list=[2.5, 2.0, 1.5] if three_res else [2.5])
    loaded = [(ctx.preclean(ctx.upload_scans(S["scans"], S["offsets"]), 2.5), ctx.poses(S["poses"], S["inv"])) for S in sess]
    ops = HipOps(ctx)
    rm = Removerter(ops, P, Session("Central", *loaded[0]), Session("Query", *loaded[1]), lane_ops=ops.lane() if lanes else None)
    rm.run()
    out = {k: v.download() for k, v in rm.outputs.items() if v is not None}
    for k, v in rm.scan_outputs().items():
        pts, off = v.download()
        out["scans:" + k] = pts
        out["off:" + k] = off
    return out


@pytest.mark.parametrize("three_res", [False, True])
def test_two_lanes_give_the_one_lane_outputs(three_res):
    from ltmapper_amd import capi
    sess = _sessions()
    ctx = capi.Context()
    one = _run(ctx, sess, three_res, False)
    two = _run(ctx, sess, three_res, True)
    assert set(one) == set(two)
    for k in one:
        a, b = one[k], two[k]
        assert a.shape == b.shape, k
        assert (a.view(np.uint32) == b.view(np.uint32)).all() if a.dtype == np.float32 else (a == b).all(), k
    ctx.close()


def test_cascade_on_two_lanes_gives_the_one_lane_cascade():
    """configs[2]'s driver with every pair run on two lanes (the deferred hand-over ends on the main lane while the query session's chain is already
    running on the other): the same outputs, run for run, as the one-lane cascade"""
    from ltmapper_amd import capi
    from ltmapper_amd.cascade import run_cascade
    from ltmapper_amd.removerter import HipOps, Params
    from tools import synth
    sess = [synth.to_numpy(synth.make_session(s, 10, "small")) for s in (1, 2, 3)]
    P = Params(gpu_use_self_removert=True, remove_resolution_list=[2.5, 2.0])
    outs = {}
    for lanes in (1, 2):
        ctx = capi.Context()
        loaded = [(ctx.preclean(ctx.upload_scans(S["scans"], S["offsets"]), 2.5), ctx.poses(S["poses"], S["inv"])) for S in sess]
        ops = HipOps(ctx)
        runs = run_cascade(ops, P, loaded[0][0], loaded[0][1], loaded[1:], lane_ops=ops.lane() if lanes == 2 else None)
        got = {}
        for j, rm in enumerate(runs):
            for k, v in rm.outputs.items():
                if v is not None:
                    got[(j, k)] = v.download()
            for k, v in rm.scan_outputs().items():
                got[(j, "scans:" + k)], got[(j, "off:" + k)] = v.download()
        outs[lanes] = got
        del runs, loaded
        ctx.close()
    assert set(outs[1]) == set(outs[2]) and len(outs[1]) > 30
    for k in outs[1]:
        a, b = outs[1][k], outs[2][k]
        assert a.shape == b.shape and ((a.view(np.uint32) == b.view(np.uint32)).all() if a.dtype == np.float32 else (a == b).all()), k


def test_lend_give_fence_events():
    from ltmapper_amd import capi
    rng = np.random.default_rng(5)
    ctx = capi.Context()
    lane = ctx.lane()
    pts = rng.normal(size=(5000, 4)).astype(np.float32)
    c = ctx.upload(pts)
    view = ctx.lend(c, lane)
    assert len(view) == 5000 and (view.download() == pts).all()
    g = lane.voxel_centroid(view, 0.5)          # work on the lane, on borrowed memory
    want = ctx.voxel_centroid(c, 0.5).download()
    assert (g.download().view(np.uint32) == want.view(np.uint32)).all()
    view.free()                                  # releases nothing
    assert (c.download() == pts).all()
    back = lane.give(g, ctx)                     # the lane's result moves into the main context without a copy
    assert g.h == 0 and (back.download().view(np.uint32) == want.view(np.uint32)).all()
    with pytest.raises(capi.LtmError):
        lane.give(ctx.lend(back, lane), ctx)     # a borrowed cloud cannot be given away
    # scan sets
    off = np.array([0, 1000, 1000, 5000], np.uint64)
    s = ctx.upload_scans(pts, off)
    sv = ctx.lend(s, lane)
    assert (sv.offsets() == off).all() and (sv.download()[0] == pts).all()
    sg = ctx.give(s, lane)
    assert s.h == 0 and (sg.download()[0] == pts).all() and (sg.offsets() == off).all()
    # events and fences: device-side ordering only, no host wait implied
    ev = ctx.event_record()
    lane.event_wait(ev)
    lane.event_wait(ev)
    ctx.fence(lane)
    lane.fence(ctx)
    lane.synchronize()
    ctx.synchronize()
    lane.close()
    ctx.close()
