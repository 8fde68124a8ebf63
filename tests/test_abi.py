"""The C-ABI library loads on a machine without a GPU and exports every symbol include/ltm.h declares
(no compute calls here).  Also: the product package never touches the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ltm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ltm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(ltm):
    names = _declared()
    assert len(names) >= 40
    lib = ltm.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ltm.h but not exported by libltm_hip.so"
        assert n in ltm.SIGNATURES, f"{n} has no ctypes signature in capi.py"
    assert sorted(ltm.SIGNATURES) == names
    assert lib.ltm_abi_version() == 1


def test_create_fails_loudly_without_gpu(ltm):
    import torch
    if torch.cuda.is_available():
        return
    cfg = ltm.LtmConfig()
    cfg.vfov, cfg.hfov, cfg.device = 50.0, 360.0, 0
    for i in range(16):
        cfg.lidar2base[i] = 1.0 if i % 5 == 0 else 0.0
    h = ctypes.c_void_p()
    rc = ltm.load_library().ltm_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -2 and not h.value, "ltm_create must fail with LTM_E_DEVICE when there is no GPU (no CPU fallback)"


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "lt-mapper_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in text and "ltm_oracle" not in text and "libltm_oracle" not in text, f"{f} references the oracle"
                # ... nor the reference-compiled build (oracle/_ref, oracle/refshim): test infrastructure as well
                assert "ref_py" not in text and "libltm_ref" not in text and "refshim" not in text and "removert_removert" not in text.replace("removert_removert node", ""), \
                    f"{f} references the reference-compiled build"


def test_product_binaries_link_neither_the_oracle_nor_the_reference_build():
    import subprocess
    for so in (os.path.join(ROOT, "lt-mapper_amd", "libltm_hip.so"), os.path.join(ROOT, "lt-mapper_amd", "host", "ltm_run")):
        if not os.path.exists(so):
            continue
        deps = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
        assert "ltm_oracle" not in deps and "ltm_ref" not in deps, deps


def test_elevation_polynomial_fit_of_the_bounded_error_projection(ltm):
    """The degree-3 elevation polynomial the vote / exact-image kernels use is fitted on the host when a context is created
    (ltm_api_core.cpp fit_elevation_poly, Lawson minimax).  Host arithmetic only, so it is checked here without a device: the
    reported error is what an independent binary32 evaluation of the returned coefficients gives, it is small enough where
    the fit is declared usable, and the decision flips where the distrust band's error budget (geom_for) says it must."""
    import numpy as np
    lib = ltm.load_library()
    c = (ctypes.c_float * 4)()
    err = ctypes.c_double()
    used = {}
    for vfov in (10.0, 26.9, 30.0, 40.0, 45.0, 50.0, 52.0, 60.0, 70.0, 86.0, 90.0, 120.0):
        rc = lib.ltm_debug_elevation_fit(ctypes.c_float(vfov), c, ctypes.byref(err))
        assert rc in (0, 1), (vfov, rc)
        used[vfov] = rc
        if 0.5 * vfov + 2.0 > 45.0:
            assert rc == 0 and list(c) == [1.0, 0.0, 0.0, 0.0], "outside the clamp argument the generic polynomial must stay"
            continue
        tmax = np.tan(np.deg2rad(0.5 * vfov + 2.0))
        t = np.linspace(0.0, tmax, 400001).astype(np.float32)
        u = (t * t).astype(np.float32)
        # float32 Horner with fused multiply-adds emulated in float64 and rounded once per step (what v_fma_f32 does)
        p = np.float32(c[3])
        for k in (2, 1, 0):
            p = (p.astype(np.float64) * u.astype(np.float64) + np.float64(c[k])).astype(np.float32)
        approx = (t.astype(np.float64) * p.astype(np.float64)).astype(np.float32)
        measured = np.abs(approx.astype(np.float64) - np.arctan(t.astype(np.float64))).max()
        assert abs(measured - err.value) <= 0.1 * err.value + 3e-8, (vfov, measured, err.value)
        assert rc == (1 if err.value <= 1.0e-6 else 0)
        # near-Taylor coefficients: 1, -1/3, 1/5, -1/7 bent by the minimax fit
        assert abs(c[0] - 1.0) < 1e-3 and abs(c[1] + 1 / 3) < 2e-2 and 0.1 < c[2] < 0.21 and -0.15 < c[3] < 0.0
    assert used[50.0] == 1 and used[26.9] == 1 and used[30.0] == 1, "the shipped sensors (os1-64 50 deg, hdl-64e 26.9 deg) use the fitted form"
    assert used[60.0] == 0 and used[70.0] == 0 and used[86.0] == 0, "3e-6 rad and worse: not good enough for the band"
    assert lib.ltm_debug_elevation_fit(ctypes.c_float(-1.0), c, ctypes.byref(err)) < 0


def _pcl_order(ltm, keys, use_std_sort, fallbacks=None):
    import numpy as np
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    out = np.empty(len(keys), dtype=np.uint32)
    fb = ctypes.c_uint32(0)
    rc = ltm.load_library().ltm_debug_pcl_sort_order(keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(keys),
                                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(use_std_sort), ctypes.byref(fb))
    assert rc == 0
    if fallbacks is not None:
        fallbacks.append(fb.value)
    return out


def test_pcl_sort_order_is_exactly_what_std_sort_leaves(ltm):
    """lt-mapper_amd/csrc/ltm_pclsort.h restates libstdc++'s introsort for (leaf index, point index) pairs compared by leaf index only -- the
    permutation pcl::VoxelGrid sums its voxels in -- without std::sort's branch mispredictions.  It must give the SAME permutation as std::sort
    (both reached through the C ABI, host arithmetic only): random sizes and key ranges with many duplicates, presorted / reversed / organ-pipe /
    constant inputs, sizes around the block and threshold boundaries, and realistic keyframe sizes."""
    import numpy as np
    rng = np.random.default_rng(20250926)
    n_cases = 0
    sizes = list(range(0, 40)) + [63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 258, 300, 511, 512, 513, 1000, 4095, 4096, 4097] + [int(v) for v in rng.integers(300, 30000, 120)]
    for n in sizes:
        for rng_span in (3, max(n // 8, 1), max(2 * n // 3, 1), 1 << 31):
            keys = rng.integers(0, rng_span, n, dtype=np.uint64).astype(np.uint32)
            variants = [keys]
            if n_cases % 5 == 0:
                s = np.sort(keys)
                noisy = s.copy()
                if n > 4:
                    i, j = rng.integers(0, n, (2, max(n // 50, 1)))
                    noisy[i], noisy[j] = s[j], s[i]
                organ = np.minimum(np.arange(n), n - np.arange(n)).astype(np.uint32)
                variants += [s, s[::-1].copy(), noisy, organ, np.full(n, 7, np.uint32)]
            for k in variants:
                a, b = _pcl_order(ltm, k, True), _pcl_order(ltm, k, False)
                assert (a == b).all(), f"n={n} span={rng_span}: the permutations differ at {int(np.flatnonzero(a != b)[0])}"
                assert (np.diff(k[a].astype(np.int64)) >= 0).all() and len(set(a.tolist())) == n
                n_cases += 1
    for n in (107_000, 131_072, 260_000):      # a scan under a 5 cm grid: ~1.5 points per leaf
        keys = rng.integers(0, 2 * n // 3, n, dtype=np.uint64).astype(np.uint32)
        assert (_pcl_order(ltm, keys, True) == _pcl_order(ltm, keys, False)).all()
        n_cases += 1
    assert n_cases > 1000


def test_pcl_sort_order_follows_std_sort_into_its_heap_sort_fallback(ltm):
    """tests/golden/stdsort_adversary_keys.npz: keys frozen by McIlroy's adversary ("A Killer Adversary for Quicksort", 1999) played against
    libstdc++'s std::sort itself (the comparator decides the keys while std::sort runs) -- inputs on which THAT std::sort runs out of its depth
    limit 2 floor(log2 n) and finishes a segment with heap sort.  The restatement must take the same turn (it reports that it did) and leave
    the same permutation; Musser's median-of-3 killer and a saw-tooth ride along."""
    import numpy as np
    adv = np.load(os.path.join(ROOT, "tests", "golden", "stdsort_adversary_keys.npz"))
    for name in adv.files:
        keys = adv[name]
        fb = []
        a, b = _pcl_order(ltm, keys, True), _pcl_order(ltm, keys, False, fb)
        assert (a == b).all(), name
        assert fb[0] >= 1, f"{name}: the adversarial input no longer reaches the heap-sort fallback (another C++ library?)"
    for n in (300, 2000, 20000, 120000):
        k = n // 2
        killer = np.zeros(n, np.uint32)
        i = np.arange(1, k + 1)
        odd = i[i % 2 == 1]
        killer[odd - 1] = odd
        killer[odd[odd < n]] = (k + odd)[odd < n]
        killer[k + i - 1] = 2 * i
        saw = (np.arange(n) % 17 * 1000 + np.arange(n) // 17).astype(np.uint32)
        for keys in (killer, killer[::-1].copy(), saw):
            assert (_pcl_order(ltm, keys, True) == _pcl_order(ltm, keys, False)).all(), n
