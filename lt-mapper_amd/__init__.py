"""ltmapper_amd -- MI355X-native LT-removert / LT-map hot path (gisbi-kim/lt-mapper, package `removert`).

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI of include/ltm.h  -> libltm_hip.so
  host/      C++ mirror of the reference's Removerter / Session class surface + the `ltm_run` CLI
  capi.py    ctypes binding of the C ABI (tests, bench.py)
  removerter.py  Python mirror of Removerter::run() over the C ABI (bench.py, parity tests)
  dist.py    keyframe sharding + label/all-gather exchange over torch.distributed (RCCL / gloo)
  cascade.py lifelong loop: scans_updated of run j become the central session of run j+1 (configs[2])

There is no CPU implementation in this package: every stage runs in libltm_hip.so on a gfx950 device.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
