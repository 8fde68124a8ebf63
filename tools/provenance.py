"""Hashes that tie a committed measurement / parity record to the sources it was made with (bench.py and the parity tools stamp
their records with them and refuse to quote a record whose hash differs from the working tree)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lt-mapper_amd", "csrc")


def _sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def kernels_sha():
    """the device code: what the PMC / rocprof figures of a kernel depend on"""
    return _sha(("ltm_kernels.hip", "ltm_device_math.h", "ltm_kernels.h"))


def product_sha():
    """device code + the C ABI orchestration: what the RESULTS of the library depend on"""
    return _sha(("ltm_kernels.hip", "ltm_device_math.h", "ltm_kernels.h", "ltm_api.cpp", "ltm_pclsort.h"))


def oracle_sha():
    h = hashlib.sha256()
    for f in ("ltm_oracle.cpp", "oracle_math.h", "ltm_oracle.h"):
        h.update(open(os.path.join(ROOT, "oracle", f), "rb").read())
    return h.hexdigest()[:16]
