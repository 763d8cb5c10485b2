"""pnec_amd -- MI355X-native PNEC relative-pose solver (hot path of tum-vision/pnec).

Layout:
  csrc/            HIP kernels + the C ABI (include/pnec_hip.h) + host C++ facade + pybind module
  capi.py          ctypes binding of libpnec_hip.so (no fallback: raises if the library is missing)
  batch.py         batches of frame pairs in HBM; numpy (host space) or torch.cuda (device space)
  simulation.py    synthetic inputs following the reference simulator's distributions (harness)
  distributed.py   one-process-per-GPU sharding of independent pair batches + one RCCL gather
"""
from . import capi  # noqa: F401
from .batch import Batch, SolveResult, select_best  # noqa: F401

__all__ = ["capi", "Batch", "SolveResult", "select_best"]
