"""rocprofv3 range markers around the phases of a training step (tracing row).

``with profiling.range("D phase"):`` pushes / pops a roctx range (libroctx64 from the ROCm image, bound with ctypes); collected
with ``rocprofv3 --marker-trace --kernel-trace -- python bench.py ...`` the kernel trace then groups under
D phase / G forward / Adam(D) / G rest / Adam(G).  Without the library (or with SP_MARKERS=0) the context manager does nothing.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

_lib = None
if os.environ.get("SP_MARKERS", "1") == "1":
    for _name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
        try:
            _lib = ctypes.CDLL(_name)
            _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            _lib.roctxRangePushA.restype = ctypes.c_int
            _lib.roctxRangePop.restype = ctypes.c_int
            break
        except (OSError, AttributeError):
            _lib = None


def enabled() -> bool:
    return _lib is not None


@contextlib.contextmanager
def range(name: str):
    if _lib is None:
        yield
        return
    _lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        _lib.roctxRangePop()
