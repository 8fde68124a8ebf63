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


def test_voxel_sort_key_compression_preserves_order_and_equality(ltm):
    """the voxel grid sorts on a Morton code with the bits left out that are functions of more significant bits inside the cloud's bounding
    box (ltm_debug_voxel_key_bits): for random boxes and random points in them the compressed code must order and group the points exactly
    like the full code, and the lot-shaped box must save at least one 8-bit radix pass"""
    rng = np.random.default_rng(8)

    def spread3(v):
        out = np.zeros_like(v, dtype=np.uint64)
        for b in range(21):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out

    def pext(code, mask):
        out = np.zeros_like(code); o = 0
        for b in range(63):
            if (mask >> b) & 1:
                out |= ((code >> np.uint64(b)) & np.uint64(1)) << np.uint64(o); o += 1
        return out

    boxes = [((-60.0, -40.0, -0.3), (60.0, 40.0, 10.2), 0.05), ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 0.4), ((-3.1, 2.0, -7.0), (-3.1, 2.0, -7.0), 0.05)]
    for _ in range(40):
        c = rng.uniform(-300, 300, 3); e = rng.uniform(0.01, 1.0, 3) * rng.choice([1.0, 30.0, 250.0], 3)
        boxes.append((tuple(c - e), tuple(c + e), float(rng.choice([0.05, 0.1, 0.4, 1.0]))))
    saved_lot = None
    for mn, mx, leaf in boxes:
        mn, mx = np.float32(mn), np.float32(mx)
        nbits, mask, depth, fmin = ltm.voxel_key_bits(mn, mx, leaf)
        assert nbits == bin(mask).count("1") or (nbits == 1 and mask == 0)
        assert mask < (1 << (3 * depth))
        pts = rng.uniform(mn.astype(np.float64), mx.astype(np.float64), (4000, 3)).astype(np.float32)
        pts = np.clip(pts, mn, mx)
        pts[:8] = [[mn[0], mn[1], mn[2]], [mx[0], mx[1], mx[2]], [mn[0], mx[1], mn[2]], [mx[0], mn[1], mx[2]], [mn[0], mn[1], mx[2]], [mx[0], mx[1], mn[2]],
                   [mn[0], mx[1], mx[2]], [mx[0], mn[1], mn[2]]]
        k = ((pts.astype(np.float64) - fmin) / np.float64(np.float32(leaf))).astype(np.uint64)
        assert (k < (1 << depth)).all()
        full = (spread3(k[:, 0]) << np.uint64(2)) | (spread3(k[:, 1]) << np.uint64(1)) | spread3(k[:, 2])
        comp = pext(full, mask)
        o = np.argsort(full, kind="stable")
        assert (np.argsort(comp, kind="stable") == o).all(), f"order changed for box {mn} {mx} leaf {leaf}"
        assert ((np.diff(full[o]) == 0) == (np.diff(comp[o]) == 0)).all(), "voxel boundaries changed"
        if saved_lot is None:
            saved_lot = 3 * depth - nbits
    assert saved_lot >= 4, f"a 120 x 80 x 10.5 m map at 0.05 m: {saved_lot} of 36 bits dropped (expected the z bits tied to the top bit)"
