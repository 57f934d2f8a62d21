"""MI355X-native low-level search engine for CBS / ECBS (the one hot path of Sartor02/libMultiRobotPlanning).

Layout:
  csrc/ll_kernel.hip      hand-written gfx950 kernels: A* / focal A*-epsilon over (time, cell) states
  csrc/mrp_ll_host.cpp    C-ABI (include/mrp_ll.h): context, map upload, batch packing, launch, results
  csrc/hl/                host-side C++ conflict-tree drivers (CBS, ECBS) that call the C-ABI (include/mrp_hl.h)
  ll.py / hl.py           ctypes plumbing used by tests and bench.py

The product path never imports ``oracle`` and has no CPU fallback.
"""
import os as _os

# one hardware queue per host worker stream (ROCm default: 4); only effective if set before HIP initialises
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")

from . import _build  # noqa: F401,E402

__all__ = ["ll", "hl", "_build"]
