"""The window path's own triangulator (opt-in): `same_delaunay2d` of libsame_hip instead of scipy.spatial.Delaunay, where that is
provably the same thing.

The reference triangulates every window's kept aligned cells with Qhull through scipy (src/same.py:1023); at ~1.6 us per point
that is three quarters of a cfg 5 pass, on the host, with the GPU waiting (`qhull_wait_share` 0.8).  libsame_hip's triangulator is
6-9 times faster, holds no GIL (plain threads instead of helper processes and pipes) and answers ONLY when its answer is beyond
doubt the SET of triangles Qhull gives (include/same_hip.h, csrc/delaunay.cpp; otherwise SAME_EUNSURE -> scipy is asked, as ever).
What it cannot give is Qhull's ORDER of the triangles and of their corners, and the reference's numbers touch that order in three
places (an XY-order edge whose ends share a coordinate; a signed area within rounding of zero; equal smallest perimeters among a
node's same-type triangles).  The device counts those places per window (`order ties`, same_window_filter_finish), and a window
with a count other than zero -- or with a cosine at the angle threshold, which the host re-decides -- is finished again with
scipy's simplices (windows.iter_device_windows).  A window's match rows, flags and sweep counters are therefore the reference's
either way; what does differ is the order of a window's kept TRIANGLES on the device (`fetch_triangles`, signs, weights), which is
why `run_same` / `sliding_window_matching` -- they hand the triangle list to the solver, index = constraint id -- keep scipy.

Use: optim_params["hip_delaunay"] = "native" (or $SAME_DELAUNAY=native) with `sliding_window_incumbent` on resident frames; the
default is "qhull".  tests/test_delaunay_cpu.py (sets of triangles against scipy, fallbacks), tests/test_gpu_delaunay.py (tables of
both ways bit-identical; forced ties), tools/delaunay_margin.py (where Qhull itself stops being exact).
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib

GUARD = 64.0   # x Qhull's round-off allowance; over 6 000 calibration sets Qhull's triangles never differed from the exact ones above 0.27


def mode(optim_params=None):
    """'native' | 'qhull' from optim_params['hip_delaunay'], else $SAME_DELAUNAY, else 'qhull'."""
    m = (optim_params or {}).get("hip_delaunay") or os.environ.get("SAME_DELAUNAY") or "qhull"
    m = str(m).lower()
    if m not in ("native", "qhull"):
        raise ValueError(f"hip_delaunay / SAME_DELAUNAY must be 'native' or 'qhull', not {m!r}")
    return m


def native_simplices(points, guard=GUARD, with_margin=False):
    """(Tr, 3) int32 counter-clockwise triangles of the Delaunay triangulation of `points` ((n, 2) float64), or None when the library
    would not answer for Qhull (SAME_EUNSURE).  Raises when libsame_hip is missing: there is no second implementation."""
    import ctypes

    lib = _lib.load()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
    n = len(pts)
    out = np.empty((max(2 * n - 5, 1), 3), np.int32)
    n_tris, margin = ctypes.c_int64(0), ctypes.c_double(0.0)
    rc = lib.same_delaunay2d(pts.ctypes.data, n, out.ctypes.data, len(out), ctypes.byref(n_tris), float(guard), ctypes.byref(margin))
    if rc == _lib.SAME_EUNSURE:
        return (None, margin.value) if with_margin else None
    if rc != 0:
        raise _lib.SameHipError(rc, "same_delaunay2d")
    tris = out[:n_tris.value]
    return (tris, margin.value) if with_margin else tris


class _Ticket:
    """`.result()` = the simplices; `.native` (after result) = they are this library's, not Qhull's; `.qhull()` = scipy's."""

    def __init__(self, owner, points, future):
        self.owner, self.points, self.future, self.native, self._qhull = owner, points, future, False, None

    def result(self):
        tris = self.future.result()
        if tris is None:
            return self.qhull()
        self.native = True
        return tris

    def qhull(self):
        if self._qhull is None:
            from . import qhull_pool

            self.owner.asked_qhull += 1
            self.native = False
            self._qhull = qhull_pool.pool().submit(self.points).result()
        return self._qhull


class NativeTriangulator:
    """`submit(points, key=None) -> ticket` like the Qhull helper pool's, answered by same_delaunay2d on a thread of this process
    (ctypes drops the GIL for the call).  `threads`: default one and a half per CPU of this process's share ($SAME_DELAUNAY_THREADS)."""

    def __init__(self, threads=None, guard=GUARD):
        from . import qhull_pool

        if threads is None:
            # one and a half threads per CPU of this process's share, as the Qhull helpers have it: a thread whose answer is ready sits
            # idle until a worker picks it up (16 CPUs: 3 500 windows/s with 16 threads, 4 000-4 200 with 24; profiles/r06_native_delaunay.log)
            share = qhull_pool.cpu_budget() / qhull_pool.cpu_sharers()[0]
            threads = int(os.environ.get("SAME_DELAUNAY_THREADS", "0")) or min(32, max(1, int(1.5 * share)))
        self.threads, self.guard = max(1, int(threads)), float(guard)
        self.pool = ThreadPoolExecutor(self.threads, thread_name_prefix="same-delaunay")
        self.submitted = self.asked_qhull = 0
        _lib.load()

    def submit(self, points, key=None):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        self.submitted += 1
        return _Ticket(self, pts, self.pool.submit(native_simplices, pts, self.guard))

    def close(self):
        self.pool.shutdown(wait=True)


_shared, _shared_lock = None, threading.Lock()


def shared():
    """The process's triangulator (threads are started once)."""
    global _shared
    with _shared_lock:
        if _shared is None:
            _shared = NativeTriangulator()
        return _shared
