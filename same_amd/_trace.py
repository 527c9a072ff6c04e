"""Stage markers for the host side of the path (the reference has none).

`with stage("prune"):` does nothing unless tracing is on.  SAME_TRACE=1 accumulates wall time per stage
(`report()` / `reset()`), and additionally emits rocTX ranges when librocprofiler-sdk-roctx / libroctx64 can be loaded, so
`rocprofv3 --marker-trace --kernel-trace` shows which kernels belong to which stage of prepare_same_inputs / run_same.
No GPU call is made here; a missing library only disables the ranges."""
import ctypes
import os
import threading
import time
from contextlib import contextmanager

_on = os.environ.get("SAME_TRACE", "0") not in ("", "0")
_totals = {}
_lock = threading.Lock()
_roctx = None


def _load_roctx():
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        for prefix in ("", "/opt/rocm/lib/"):
            try:
                lib = ctypes.CDLL(prefix + name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                return lib
            except (OSError, AttributeError):
                continue
    return False


def enable(on=True):
    global _on
    _on = bool(on)


def enabled():
    return _on


@contextmanager
def stage(name):
    if not _on:
        yield
        return
    global _roctx
    if _roctx is None:
        _roctx = _load_roctx()
    if _roctx:
        _roctx.roctxRangePushA(f"same:{name}".encode())
    t0 = time.perf_counter()
    try:
        yield
    finally:
        dt = time.perf_counter() - t0
        if _roctx:
            _roctx.roctxRangePop()
        with _lock:
            n, tot = _totals.get(name, (0, 0.0))
            _totals[name] = (n + 1, tot + dt)


def add(name, seconds):
    """Account `seconds` to `name` (used by _lib.instrument() for time spent inside libsame_hip calls)."""
    with _lock:
        n, tot = _totals.get(name, (0, 0.0))
        _totals[name] = (n + 1, tot + seconds)


def report():
    """{stage: (calls, seconds)} accumulated since the last reset()."""
    with _lock:
        return dict(_totals)


def reset():
    with _lock:
        _totals.clear()
