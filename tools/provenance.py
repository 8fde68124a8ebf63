"""Hashes that tie a committed measurement / parity record to the sources it was made with (bench.py and the parity tools stamp
their records with them and refuse to quote a record whose hash differs from the working tree)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lt-mapper_amd", "csrc")
# round 6: the two monolithic sources were split by stage
KERNEL_SOURCES = ("ltm_kernels_common.h", "ltm_k_projection.hip", "ltm_k_stream.hip", "ltm_k_voxel.hip", "ltm_k_knn.hip", "ltm_device_math.h", "ltm_kernels.h")
API_SOURCES = ("ltm_internal.h", "ltm_api_core.cpp", "ltm_api_vote.cpp", "ltm_api_voxel.cpp", "ltm_api_knn.cpp", "ltm_pclsort.h")


def _sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def kernels_sha():
    """the device code: what the PMC / rocprof figures of a kernel depend on"""
    return _sha(KERNEL_SOURCES)


def product_sha():
    """device code + the C ABI orchestration: what the RESULTS of the library depend on"""
    return _sha(KERNEL_SOURCES + API_SOURCES)


def oracle_sha():
    h = hashlib.sha256()
    for f in ("ltm_oracle.cpp", "oracle_math.h", "ltm_oracle.h"):
        h.update(open(os.path.join(ROOT, "oracle", f), "rb").read())
    return h.hexdigest()[:16]
