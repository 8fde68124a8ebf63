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
    (ltm_api.cpp fit_elevation_poly, Lawson minimax).  Host arithmetic only, so it is checked here without a device: the
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
