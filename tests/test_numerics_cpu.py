"""CPU tests of the numerics that are restated from knowledge (parity unpinned): the 4x4 inverse that stands for
Eigen::Matrix4d::inverse() (Session.cpp:109-110, RosParamServer.cpp:29-30) and what its last bits -- or the last bit of atan2f --
can move in the outputs (tools/numerics_sensitivity.py; the full-size record is profiles/r3_numerics_sensitivity_*.json)."""
import numpy as np


def _random_pose(rng, six_digits=False):
    from scipy.spatial.transform import Rotation
    T = np.eye(4)
    T[:3, :3] = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix()
    T[:3, 3] = rng.normal(size=3) * 100.0
    if six_digits:      # pose text files carry 6 significant digits (ltslam/src/utility.cpp:190-200): not exactly orthonormal
        T = np.array([[float("%.6g" % v) for v in r] for r in T])
    return T


def test_library_and_oracle_inverse_agree_bitwise_and_invert(ltm, orc):
    """the product's ltm_inverse4x4 and the oracle's orc_inverse4x4 are two restatements of the same Eigen kernel: same bits;
    both are inverses to 1e-12 (rigid poses, 6-digit poses, general matrices); singular matrices are refused"""
    rng = np.random.default_rng(4)
    mats = [_random_pose(rng, six_digits=bool(i & 1)) for i in range(300)] + [rng.normal(size=(4, 4)) for _ in range(50)] + [np.eye(4)]
    for M in mats:
        a, b = ltm.inverse4x4(M), orc.inverse4x4(M)
        assert (a.view(np.uint64) == b.view(np.uint64)).all(), "library and oracle inverse differ"
        assert np.abs(a @ M - np.eye(4)).max() <= 1e-9 * max(1.0, np.abs(M).max() ** 2)
        assert np.abs(a - np.linalg.inv(M)).max() <= 1e-10 * max(1.0, np.abs(np.linalg.inv(M)).max())
    assert (ltm.inverse4x4(np.eye(4))[:3] == np.eye(4)[:3]).all()
    sing = np.ones((4, 4))
    try:
        ltm.inverse4x4(sing)
        raise AssertionError("singular matrix accepted")
    except ValueError:
        pass


def test_inverse_variants_differ_only_in_last_bits(orc):
    rng = np.random.default_rng(5)
    differing = 0
    for i in range(200):
        M = _random_pose(rng, six_digits=True)
        base = orc.inverse4x4(M)
        for v in (1, 2):
            other = orc.inverse4x4(M, v)
            assert np.abs(other - base).max() <= 1e-12 * max(1.0, np.abs(base).max())
            differing += int((other != base).any())
    assert differing > 0, "the three evaluation orders are expected to disagree in some last bits (otherwise the sensitivity test tests nothing)"


def test_outputs_are_insensitive_to_the_unpinned_last_bits(orc):
    """a small session pair through the whole pipeline: inverse by cofactor expansion instead of the Eigen order, and EVERY atan2f moved
    by one ulp.  The fraction of output points that change bounds what the unpinned third-party arithmetic can do to parity."""
    from tools import synth
    from tools.numerics_sensitivity import run_experiment
    C, Q = (synth.to_numpy(synth.make_session(s, 8, "small")) for s in (1, 2))
    out, _ = run_experiment(orc, C, Q, threads=4, quick=True)
    inv = out["inverse_cofactor"]
    assert inv["inverse_entries_differing_bitwise"] >= 0
    assert inv["fraction"] <= 1e-4, f"last bits of the inverse moved {inv['points_differing']} of {inv['points_compared']} output points"
    at = out["atan2f_1ulp_every_call"]
    assert at["points_compared"] > 100_000
    assert at["fraction"] <= 2e-3, f"+-1 ulp on every atan2f moved {at['points_differing']} of {at['points_compared']} output points"
