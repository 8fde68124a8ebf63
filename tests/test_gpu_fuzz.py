"""Randomised end-to-end parity (tools/fuzz_parity.py): small session pairs with randomly drawn field of view, resolutions, kNN k / threshold,
voxel size, LiDAR->base extrinsic, keyframe batch size, scene and sensor -- every output of Removerter::run() on the GPU against the oracle,
bitwise.  A fixed seed in the suite; the tool runs more cases (profiles/r3_fuzz_parity_*.json)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_parameter_draws_keep_bitwise_parity():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--n", "10", "--seed", "20250224"], capture_output=True, text=True, timeout=900)
    assert r.stdout.strip(), r.stderr[-2000:]
    d = json.loads(r.stdout)
    bad = [c for c in d["detail"] if c["outputs_differing"]]
    assert r.returncode == 0 and not bad, f"outputs differ from the oracle for: {bad}"
    assert d["cases"] == 10 and all(c["outputs_compared"] == 23 for c in d["detail"])
