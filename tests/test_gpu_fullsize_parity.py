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
