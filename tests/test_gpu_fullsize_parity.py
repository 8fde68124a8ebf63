"""BASELINE configs[1] AT FULL SIZE against the oracle, on every box that has the host cores for it: 2 x 500 keyframes of the os1-64
sensor, 3-res selfRemovert, all 18 maps and 5 scan sets of Removerter::run() compared bitwise (2 x 22 M scan points in).  The oracle
parallelises over keyframes like the reference's OpenMP sites; with 256 threads the comparison takes ~200 s, single-threaded 49
minutes -- hence the core-count gate.  The record also lands in gpurun_out/ so that it can be committed under profiles/."""
import json
import os

import pytest

from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

MIN_CORES = 64


@pytest.mark.skipif((os.cpu_count() or 1) < MIN_CORES, reason=f"the full-size oracle run needs >= {MIN_CORES} host cores (49 min single-threaded)")
def test_config1_full_size_2x500_three_res_bitwise(ltm, orc):
    from tools.parity_fullsize import run_parity
    rep = run_parity(config=1)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_fullsize_2x500_3res.json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass
    differing = [k for k, v in rep["outputs"].items() if not v["identical"]]
    assert rep["outputs_compared"] == 23 and not differing, f"outputs differing from the oracle at full size: {differing}"
    assert min(rep["scan_points"]) > 20_000_000


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


@pytest.mark.skipif((os.cpu_count() or 1) < MIN_CORES, reason=f"the oracle run needs >= {MIN_CORES} host cores")
def test_config3_real_sensor_2x120_hdl64e_three_res_occlusion_cull_forced(ltm, orc):
    """configs[3] at its real sensor size in the driver-run suite: street scene, hdl-64e (64 x 1900 rays), 2 x 120 keyframes, the
    3-res variant, the occlusion-culled launch of the exact-image kernel forced on for every reprojection / ND vote -- all 23 outputs bitwise"""
    from tools.parity_fullsize import run_parity
    rep = _with_env(dict(LTM_OCCLUSION=1, LTM_OCCLUSION_MIN_PAIRS=0), lambda: run_parity(config=33, kf=120))
    differing = [k for k, v in rep["outputs"].items() if not v["identical"]]
    assert rep["outputs_compared"] == 23 and not differing, f"outputs differing from the oracle: {differing}"
    assert min(rep["scan_points"]) > 10_000_000
    try:
        with open(os.path.join(ROOT, "gpurun_out", "parity_street_2x120_hdl64e_3res_occlusion_forced.json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass


@pytest.mark.skipif((os.cpu_count() or 1) < MIN_CORES, reason=f"the oracle run needs >= {MIN_CORES} host cores")
def test_config4_real_sensor_mls_128x8192_2x10_cull_forced(ltm, orc):
    """configs[4] at its real sensor size: 128 x 8192 = 1 M rays per scan, 2 x 10 keyframes at 2 m, voxel 0.1, k = 2, thr 0.04, occlusion cull forced"""
    from tools.parity_fullsize import run_parity
    rep = _with_env(dict(LTM_OCCLUSION=1, LTM_OCCLUSION_MIN_PAIRS=0), lambda: run_parity(config=4, kf=10))
    differing = [k for k, v in rep["outputs"].items() if not v["identical"]]
    assert rep["outputs_compared"] == 23 and not differing, f"outputs differing from the oracle: {differing}"
    assert min(rep["scan_points"]) > 5_000_000
    try:
        with open(os.path.join(ROOT, "gpurun_out", "parity_street_2x10_mls_cull_forced.json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass
