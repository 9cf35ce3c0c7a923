"""rocTX ranges around the kernel groups of a run ("Tracing"): ``rocprofv3 --marker-trace --kernel-trace`` then
shows which stage -- stain pre-normalisation, CNN forward, result gather, post-processing, canvas stitching -- a kernel belongs to.

``with tracing.range("cnn_forward"): ...`` pushes / pops a range through ``libroctx64.so`` when the library can be loaded (it ships
with ROCm); without it, or with ``TIA_ROCTX=0``, the context manager does nothing.  Ranges nest; they cost two C calls each and no
synchronisation, so they stay on in production runs.
"""

from __future__ import annotations

import contextlib
import ctypes
import os

_LIB = None
_TRIED = False


def _lib():
    global _LIB, _TRIED  # noqa: PLW0603
    if _TRIED:
        return _LIB
    _TRIED = True
    if os.environ.get("TIA_ROCTX", "1") == "0":
        return None
    for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            _LIB = lib
            break
        except (OSError, AttributeError):
            continue
    return _LIB


def enabled() -> bool:
    return _lib() is not None


@contextlib.contextmanager
def range(name: str):  # noqa: A001  (mirrors roctx / nvtx naming)
    lib = _lib()
    if lib is None:
        yield
        return
    lib.roctxRangePushA(("tia/" + name).encode())
    try:
        yield
    finally:
        lib.roctxRangePop()
