"""The culled projection kernels validate their error bounds on the device the first time a context uses an image shape and fall back to the exact
kernels for a shape that fails (ltm_debug_cull_validation, include/ltm.h).  Forced here with a uselessly narrow distrust band: the labels must not change."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _vote(env):
    from ltmapper_amd import capi
    from tools import synth
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = capi.Context()          # the switches are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    S = synth.to_numpy(synth.make_session(1, 6, "small"))
    scans, poses = ctx.upload_scans(S["scans"], S["offsets"]), ctx.poses(S["poses"], S["inv"])
    cmap = ctx.voxel_centroid(ctx.merge_to_global(scans, poses), 0.05)
    out = {}
    for alpha in (2.5, 2.375, 1.5):
        _, _, labels = ctx.visibility_partition(cmap, scans, poses, alpha, 0.1, 0, want_labels=True)
        out[alpha] = labels
    rep = ctx.reproject(cmap, poses, 3.0).download()
    stats = ctx.cull_validation()
    ctx.close()
    return out, rep, stats


def test_a_shape_that_fails_the_validation_is_served_by_the_exact_kernels():
    good, rep_good, (checked, failed) = _vote({})
    assert checked >= 4 and failed == 0, "the shipped bounds must hold for every shape in use"
    # a band of 1e-9 pixels: points within the real error of a pixel boundary are no longer sent to the exact path -> the validator must see them
    bad, rep_bad, (checked_b, failed_b) = _vote({"LTM_CULL_EPS_SCALE": "1e-9", "LTM_CULL_EPS_FLOOR": "0"})
    assert failed_b == checked_b >= 4, "every shape must fail with a band this narrow"
    for a in good:
        assert (good[a] == bad[a]).all(), f"labels changed at resolution {a}"
    assert (rep_good[0].view(np.uint32) == rep_bad[0].view(np.uint32)).all() and (rep_good[1] == rep_bad[1]).all()
    # and with the validation switched off the same band does produce the culled kernels' answers (nothing checked): the check is what protected the labels
    _, _, (checked_off, _) = _vote({"LTM_CULL_EPS_SCALE": "1e-9", "LTM_CULL_EPS_FLOOR": "0", "LTM_CULL_SELFCHECK": "0"})
    assert checked_off == 0
