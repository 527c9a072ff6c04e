"""Delaunay triangulations computed ahead of time in helper processes.

The triangulation of a window's aligned cells is an INPUT of the device path (scipy/Qhull on the host, exactly as the
reference calls it, src/same.py:1023) and, at ~45 ms per 12.7k-cell window, three quarters of a window's pre-MIP wall
time; scipy holds the GIL for the whole call, so a thread cannot hide it.  The window loop therefore hands the points of
windows n+1..n+k to a few helper processes while window n runs its prune / filter / cost / sweep kernels (and, in a real
run, its MIP solve), and picks the simplices up when it gets there.  Same library, same input bytes, same options: the
simplices are identical to an in-process call.

Helpers are plain `python -c` children speaking a length-prefixed binary protocol over their pipes (no multiprocessing:
nothing re-imports the caller's `__main__`, nothing is forked from a process that has initialised the GPU, and the
helpers import numpy + scipy.spatial only -- they never touch the GPU).  `SAME_QHULL_WORKERS` sets their number
(default: up to 4, at most half the cores; 0 = compute in-process, no helpers).
"""
import atexit
import os
import struct
import subprocess
import sys
import threading

import numpy as np

_WORKER = r"""
import struct, sys
import numpy as np
from scipy.spatial import Delaunay
inp, out = sys.stdin.buffer, sys.stdout.buffer
while True:
    head = inp.read(8)
    if len(head) < 8:
        break
    (n,) = struct.unpack("<q", head)
    if n < 0:
        break
    buf = inp.read(16 * n)
    try:
        s = np.ascontiguousarray(Delaunay(np.frombuffer(buf, np.float64).reshape(n, 2)).simplices, dtype=np.int32)
        out.write(struct.pack("<q", len(s)) + s.tobytes())
    except Exception:
        out.write(struct.pack("<q", -1))     # the caller repeats the call in-process to raise the same error
    out.flush()
"""


def _delaunay_here(points):
    from scipy.spatial import Delaunay

    return Delaunay(points).simplices


class _Ticket:
    def __init__(self, pool, worker, points):
        self.pool, self.worker, self.points, self._value = pool, worker, points, None

    def result(self):
        """The (Tr, 3) int32 simplices -- blocks until the helper has answered."""
        if self._value is None:
            self._value = self.pool._collect(self)
        return self._value


class QhullPool:
    def __init__(self, workers):
        self.n = int(workers)
        self.procs = []
        self.pending = {}            # worker index -> ticket whose answer has not been read yet
        self.next = 0
        self.lock = threading.Lock()

    def _spawn(self):
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        return subprocess.Popen([sys.executable, "-c", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)

    def submit(self, points):
        """Start Delaunay(points) in a helper; -> ticket with .result().  points: (n, 2) float64."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        with self.lock:
            if self.n <= 0:
                return _Ticket(self, None, pts)
            # a started helper with nothing in flight is preferred (a fresh one needs ~0.3 s to import scipy); a new helper
            # is started only while every running one is busy; when all n are busy the next in turn is drained and reused
            for q in range(len(self.procs)):            # a helper that died while idle is replaced where it stood
                if q not in self.pending and self.procs[q].poll() is not None:
                    self.procs[q] = self._spawn()
            idle = [q for q in range(len(self.procs)) if q not in self.pending]
            if idle:
                w = idle[0]
            elif len(self.procs) < self.n:
                w = len(self.procs)
                self.procs.append(self._spawn())
            else:
                w = self.next % self.n
                self.next += 1
            if w in self.pending:                       # one request in flight per helper: read the old answer first
                old = self.pending.pop(w)
                old._value = self._read(w, old)
            t = _Ticket(self, w, pts)
            if self.procs[w].poll() is not None:        # the helper has died since its last request: start another
                self.procs[w] = self._spawn()
            try:
                p = self.procs[w]
                p.stdin.write(struct.pack("<q", len(pts)) + pts.tobytes())
                p.stdin.flush()
                self.pending[w] = t
            except (OSError, ValueError):
                t.worker = None                          # helper is gone: this one is computed in-process on result()
            return t

    def _read(self, w, ticket):
        p = self.procs[w]
        try:
            head = p.stdout.read(8)
            if len(head) == 8:
                (n,) = struct.unpack("<q", head)
                if n >= 0:
                    raw = p.stdout.read(12 * n)
                    if len(raw) == 12 * n:
                        return np.frombuffer(raw, np.int32).reshape(n, 3).copy()
        except OSError:
            pass
        if p.poll() is not None:                         # died: replace it for later requests
            self.procs[w] = self._spawn()
        return _delaunay_here(ticket.points)             # error in the helper (or a dead helper): same call, here

    def _collect(self, ticket):
        if ticket.worker is None:
            return _delaunay_here(ticket.points)
        with self.lock:
            if self.pending.get(ticket.worker) is ticket:
                del self.pending[ticket.worker]
                return self._read(ticket.worker, ticket)
        return ticket._value if ticket._value is not None else _delaunay_here(ticket.points)

    def close(self):
        with self.lock:
            for p in self.procs:
                try:
                    p.stdin.write(struct.pack("<q", -1))
                    p.stdin.flush()
                    p.stdin.close()
                except (OSError, ValueError):
                    pass
            for p in self.procs:
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    p.kill()
            self.procs, self.pending = [], {}


_pool = None
_pool_lock = threading.Lock()


def default_workers():
    v = os.environ.get("SAME_QHULL_WORKERS")
    if v is not None:
        return max(0, int(v))
    return max(0, min(4, (os.cpu_count() or 1) // 2))


def warm(count=None):
    """Start up to `count` helpers now (default: all), so that their interpreter start-up is not paid by the first requests."""
    p = pool()
    with p.lock:
        while len(p.procs) < min(p.n, p.n if count is None else int(count)):
            p.procs.append(p._spawn())


def pool():
    """Process-wide pool, created on first use and closed at interpreter exit."""
    global _pool
    with _pool_lock:
        if _pool is None:
            _pool = QhullPool(default_workers())
            atexit.register(_pool.close)
        return _pool


def lookahead():
    """How many windows ahead of the running one are triangulated (one per helper)."""
    return pool().n
