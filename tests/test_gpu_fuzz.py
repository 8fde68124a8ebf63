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


def test_random_draws_under_full_se3_poses_keep_bitwise_parity():
    """every end-to-end case so far had yaw-only poses at constant height.  Here every keyframe is rolled / pitched by N(0, 1-3 deg),
    drifts in z (the rays are cast from that attitude, so the scans stay consistent with the scene) and half of the session pairs lie 10-50 km
    from the coordinate origin, where the 6 significant digits of the pose text quantise translations to 0.1 m (tools/synth.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--n", "8", "--seed", "424242", "--se3"], capture_output=True, text=True, timeout=900)
    assert r.stdout.strip(), r.stderr[-2000:]
    d = json.loads(r.stdout)
    bad = [c for c in d["detail"] if c["outputs_differing"]]
    assert r.returncode == 0 and not bad, f"outputs differ from the oracle for: {bad}"
    assert d["cases"] == 8 and all(c["outputs_compared"] == 23 and c["tilt_deg"] >= 1.0 for c in d["detail"])
    assert any(abs(c["origin"][0]) > 1e4 for c in d["detail"]) and any(c["origin"][0] == 0.0 for c in d["detail"])
