"""traceweaver_amd -- MI355X-native span->parent assignment engine behind TraceWeaver's predictor API.

Only the hot path of the reference's `trace_reconstructor` lives here:
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/traceweaver_amd.h)
  engine.py    numpy-level wrapper around the C-ABI (ctypes)
  predictor.py `TraceWeaverGPU`, a drop-in for TraceWeaverV3.FindAssignments (traceweaver_v3.py:1087)
  gmm.py       the per-edge mixture refit between the two passes (traceweaver_v3.py:764-786)
  sharding.py  one-process-per-GPU partitioning of independent service units
  synth.py     synthetic Jaeger-shape workloads (bench / scale tests)
"""
from .engine import Engine, EngineError, UnitArrays  # noqa: F401

__all__ = ["Engine", "EngineError", "UnitArrays"]
