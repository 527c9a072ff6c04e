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

# x Qhull's round-off allowance: 60 x the largest ratio (0.27) at which Qhull's triangles differed from the exact ones in 6 000
GUARD = 16.0
               # calibration sets (an edge's property, not a set's: larger sets have smaller margins only because they have more edges)


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
        tris, asked = self.future.result()
        if tris is None:                        # left to Qhull by the library: a helper has been at it since
            if self._qhull is None:
                self._qhull = asked.result()
                self.owner._note(True)
            return self._qhull
        self.native = True
        self.owner._note(False)
        return tris

    def qhull(self):
        """scipy's simplices for a window whose numbers hang on Qhull's order (asked for now: nobody could know before)"""
        if self._qhull is None:
            from . import qhull_pool

            self._qhull = qhull_pool.pool().submit(self.points).result()
            self.owner._note(True)
        self.native = False
        return self._qhull


class _QhullTicket:
    """a ticket of the Qhull helper pool behind the same face (`native` stays false)"""
    native = False

    def __init__(self, ticket):
        self.ticket = ticket

    def result(self):
        return self.ticket.result()

    qhull = result


class NativeTriangulator:
    """`submit(points, key=None) -> ticket` like the Qhull helper pool's, answered by same_delaunay2d on a thread of this process
    (ctypes drops the GIL for the call).  `threads`: default one and a half per CPU of this process's share ($SAME_DELAUNAY_THREADS).
    A set the library leaves to Qhull goes to a Qhull helper from the triangulator's thread, windows ahead of its use.  Where most
    windows end up with scipy anyway (sections on a lattice; whole-number coordinates: order ties in every window) the triangulator
    steps aside: after `WINDOW` tickets of which more than half went back to scipy, the next `BYPASS` go to the helpers directly."""

    WINDOW, BYPASS = 16, 128

    def __init__(self, threads=None, guard=GUARD):
        from collections import deque

        from . import qhull_pool

        if threads is None:
            # one and a half threads per CPU of this process's share (as the Qhull helpers have it): the worker threads and the runtime's
            # own threads take CPU time too, and a triangulator thread that is descheduled holds a window back.  Same lease, back to back
            # (profiles/r06_native_threads_ab.log): two ranks on 16 CPUs 4 400-4 850 windows/s with 8 threads each, 5 260-5 350 with 12;
            # one rank 3 930-4 280 with 16 and 3 630-4 330 with 24 (no difference beyond the lease's noise)
            share = qhull_pool.cpu_budget() / qhull_pool.cpu_sharers()[0]
            threads = int(os.environ.get("SAME_DELAUNAY_THREADS", "0")) or min(32, max(1, int(1.5 * share)))
        self.threads, self.guard = max(1, int(threads)), float(guard)
        self.pool = ThreadPoolExecutor(self.threads, thread_name_prefix="same-delaunay")
        self.submitted = self.asked_qhull = self.bypassed = 0
        self._recent, self._bypass, self._lock = deque(maxlen=self.WINDOW), 0, threading.Lock()
        _lib.load()

    def _note(self, sent_back):
        with self._lock:
            self.asked_qhull += bool(sent_back)
            self._recent.append(bool(sent_back))
            if len(self._recent) == self.WINDOW and 2 * sum(self._recent) > self.WINDOW:
                self._bypass = self.BYPASS
                self._recent.clear()

    def _work(self, pts):
        tris = native_simplices(pts, self.guard)
        if tris is not None:
            return tris, None
        from . import qhull_pool

        return None, qhull_pool.pool().submit(pts)

    def submit(self, points, key=None):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        with self._lock:
            self.submitted += 1
            bypass = self._bypass > 0
            if bypass:
                self._bypass -= 1
                self.bypassed += 1
        if bypass:
            from . import qhull_pool

            return _QhullTicket(qhull_pool.pool().submit(pts))
        return _Ticket(self, pts, self.pool.submit(self._work, pts))

    def reset(self):
        """forget what the recent windows did (a new job may be nothing like the last)"""
        with self._lock:
            self._recent.clear()
            self._bypass = 0

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
