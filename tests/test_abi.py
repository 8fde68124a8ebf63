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
