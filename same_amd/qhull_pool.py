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
(default: one and a half times this process's SHARE of the CPUs it may use, at most 24 -- a helper is idle while its points and
simplices travel and until the next request reaches it, so a modest over-subscription keeps the cores busy: 576 -> 694 windows/s
from 16 to 24 helpers under a 16-CPU quota, profiles/r03_cfg5_device_pipeline.log; 0 = compute in-process, no helpers).  The share:
ranks launched plainly see the same affinity mask and cgroup quota, so the budget is divided by the ranks that share it
(`cpu_sharers()`: `local_world()` = LOCAL_WORLD_SIZE, else WORLD_SIZE of a single-host rendezvous -- unless the mask is already a
1 / L slice of the host, i.e. the launcher bound every rank to CPUs of its own) -- eight ranks on a 16-CPU quota start 24
helpers between them, not 192 (two ranks on one box ran 455 windows/s against 714 for one before the division).

Placement matters more than the count: Qhull lives in the last-level cache, and eight helpers that the scheduler stacks on one
CCD of an EPYC host triangulate a 13 000-point set in 45 ms each against 21 ms alone.  Helper i is therefore confined to the
CPUs of ONE L3 domain, consecutive helpers to different ones, and the local ranks of a job divide the host's domains between
them (rank r of L takes the r-th of L equal runs of domains): 7.3 -> 2.7 ms per set with eight helpers on the MI355X box's host
(profiles/r03_qhull_scaling.log).  `SAME_QHULL_PIN=0` leaves placement to the
scheduler.
"""
import atexit
import os
import struct
import subprocess
import sys
import threading

import numpy as np

_MAGIC = 0x51484C4C   # "QHLL": first word of every reply, so stray bytes on the pipe are noticed instead of read as a length

_WORKER = r"""
import os, struct, sys
# the protocol owns the ORIGINAL stdout; anything else that prints (a Qhull or BLAS banner, a warnings hook) lands on stderr
proto = os.fdopen(os.dup(1), "wb")
os.dup2(2, 1)
sys.stdout = sys.stderr
import numpy as np
from scipy.spatial import Delaunay
inp = sys.stdin.buffer
while True:
    head = inp.read(16)
    if len(head) < 16:
        break
    seq, n = struct.unpack("<qq", head)
    if n < 0:
        break
    buf = inp.read(16 * n)
    if len(buf) < 16 * n:
        break
    try:
        s = np.ascontiguousarray(Delaunay(np.frombuffer(buf, np.float64).reshape(n, 2)).simplices, dtype=np.int32)
        proto.write(struct.pack("<iqq", 0x51484C4C, seq, len(s)) + s.tobytes())
    except Exception:
        proto.write(struct.pack("<iqq", 0x51484C4C, seq, -1))     # the caller repeats the call in-process to raise the same error
    proto.flush()
"""


def _delaunay_here(points):
    from scipy.spatial import Delaunay

    return Delaunay(points).simplices


class _Ticket:
    def __init__(self, pool, worker, points, seq=0):
        self.pool, self.worker, self.points, self._value, self.seq = pool, worker, points, None, seq

    def result(self):
        """The (Tr, 3) int32 simplices -- blocks until the helper has answered."""
        if self._value is None:
            self._value = self.pool._collect(self)
        return self._value


def _l3_domains():
    """The CPUs this process may run on, one list per last-level-cache domain (a CCD on EPYC); [] when the host does not say."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return []
    domains = {}
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list") as f:
                key = f.read().strip()
        except OSError:
            return []
        domains.setdefault(key, []).append(c)
    return list(domains.values())


def cpu_budget():
    """CPUs this process can actually keep busy: its affinity mask, cut to the cgroup's CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:                                                           # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def local_world():
    """(ranks of this job that share this host's CPUs, this rank's index among them).  LOCAL_WORLD_SIZE / LOCAL_RANK when the launcher
    sets them (torch.distributed.run, bench.launch); else WORLD_SIZE / RANK when the rendezvous is visibly on this host alone (a loopback
    MASTER_ADDR, or bench.py's own rendezvous directory); else (1, 0).  SAME_LOCAL_WORLD overrides the count."""
    def num(name):
        try:
            return int(os.environ[name])
        except (KeyError, ValueError):
            return None

    n, r = num("SAME_LOCAL_WORLD"), num("LOCAL_RANK")
    if n is None:
        n = num("LOCAL_WORLD_SIZE")
    if n is None and (os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1") or os.environ.get("SAME_RDV_DIR")):
        n = num("WORLD_SIZE")
        if r is None:
            r = num("RANK")
    n = max(1, n or 1)
    return n, min(max(0, r or 0), n - 1)


# (ranks whose CPUs overlap this rank's, this rank's index among them), from the ranks' own masks (learn_cpu_sharing)
_LEARNED_SHARERS = None


def learn_cpu_sharing(group):
    """Decide from EVIDENCE whether this rank shares its CPUs with other ranks of the job: every rank tells the others (over the host
    group: dist.py / bench.py call this once the group is up) its host and its affinity mask.  Ranks on this host whose masks overlap
    this rank's share its CPUs -- a plain launch, or one container cpuset for all ranks (docker --cpuset-cpus=0-15 for eight ranks on a
    128-CPU host: every rank's mask is 16 CPUs and THE SAME 16) -- and the budget is divided among them; a bound launch (slurm
    --cpu-bind, numactl, one cpuset per rank) gives pairwise disjoint masks and every rank keeps its whole mask.  -> cpu_sharers()."""
    global _LEARNED_SHARERS
    import socket

    try:
        mask = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        mask = None
    every = group.allgather_object({"host": socket.gethostname(), "mask": mask, "rank": int(group.rank)})
    me = every[group.rank]
    if mask is None:
        _LEARNED_SHARERS = None
        return cpu_sharers()
    mine = set(mask)
    sharing = sorted(e["rank"] for e in every if e["host"] == me["host"] and (e["mask"] is None or mine & set(e["mask"])))
    _LEARNED_SHARERS = (len(sharing), sharing.index(int(group.rank)))
    return _LEARNED_SHARERS


def cpu_sharers():
    """(ranks of this job that share the CPUs of `cpu_budget()`, this rank's index among them).  Ranks launched plainly all see the same
    affinity mask and cgroup quota, and the budget is divided between them.  A BOUND launch (slurm --cpu-bind, numactl, one cpuset per
    rank) hands every rank a mask of its own: `cpu_budget()` is then already this rank's share and dividing it again would leave 1 / L
    of the helpers (3 instead of 24 at eight ranks).  The two cannot be told apart by the size of a mask (one container cpuset shared
    by all ranks is small too), only by comparing the ranks' masks: `learn_cpu_sharing(group)` does, where a host group is up.
    Without that evidence the budget is divided by `local_world()` -- the safe side: oversubscribing the CPUs with helpers costs far
    more (455 against 714 windows/s measured) than leaving some idle; a bound launch without a host group says SAME_CPU_SHARERS=1."""
    n, r = local_world()
    v = os.environ.get("SAME_CPU_SHARERS")
    if v is not None:
        try:
            n = max(1, int(v))
            return n, min(r, n - 1)
        except ValueError:
            pass
    if _LEARNED_SHARERS is not None:
        return _LEARNED_SHARERS
    return (n, r) if n > 1 else (1, 0)


def _domain_share(domains, n_local, local_rank):
    """The L3 domains local rank `local_rank` of `n_local` places its helpers in: an equal run of the list when there are at least as
    many domains as ranks, otherwise the one domain the rank's position falls into (ranks then share it)."""
    d = len(domains)
    if d == 0 or n_local <= 1:
        return list(domains)
    if d >= n_local:
        lo, hi = local_rank * d // n_local, (local_rank + 1) * d // n_local
        return domains[lo:hi]
    return [domains[local_rank * d // n_local]]


class QhullPool:
    def __init__(self, workers, pin=None):
        self.n = int(workers)
        if pin is None:
            pin = os.environ.get("SAME_QHULL_PIN", "1") not in ("", "0", "off")
        self.domains = _l3_domains() if pin else []
        if len(self.domains) < 2:
            self.domains = []                                          # one cache domain (or an unknown layout): nothing to choose
        self.domains = _domain_share(self.domains, *cpu_sharers())     # this rank's part of the CPUs it shares with other local ranks
        self.first_domain = 0
        self.procs = []
        self.pending = {}            # worker index -> ticket whose answer has not been read yet
        self.writing = set()         # workers whose request is being written (outside the lock): busy, and not to be drained
        self.next = 0
        self.seq = 0                 # request number, echoed by the helper: an answer is only taken for the request it names
        self.lock = threading.Lock()

    def _spawn(self, w):
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        p = subprocess.Popen([sys.executable, "-c", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
        if self.domains:                                               # helper w lives in one L3 domain (see the module text)
            try:
                os.sched_setaffinity(p.pid, self.domains[(self.first_domain + w) % len(self.domains)])
            except (AttributeError, OSError):
                pass
        # a window's points are ~200 KB and its simplices ~250 KB; with the default 64 KiB pipes the submitting side blocks in
        # write() until a busy helper gets round to reading, i.e. it waits for Qhull after all.  1 MiB is the unprivileged limit.
        try:
            import fcntl

            for f in (p.stdin, p.stdout):
                fcntl.fcntl(f.fileno(), getattr(fcntl, "F_SETPIPE_SZ", 1031), 1 << 20)
        except (ImportError, OSError, ValueError):
            pass
        return p

    def _readable(self, workers):
        """Those of `workers` whose answer (or end of file) has arrived.  Call with the lock held."""
        import select

        by_fd = {}
        for w in workers:
            try:
                by_fd[self.procs[w].stdout.fileno()] = w
            except (OSError, ValueError):
                return [w]                                # its pipe is already closed: reading it will say so
        if not by_fd:
            return []
        try:
            ready = select.select(list(by_fd), [], [], 0)[0]
        except (OSError, ValueError):
            return list(workers)
        return sorted(by_fd[fd] for fd in ready)

    @staticmethod
    def _wait(files, timeout):
        """Sleep until one of the helpers' pipes has something to read (lock NOT held: other threads keep using the pool)."""
        import select

        try:
            select.select([f.fileno() for f in files], [], [], timeout)
        except (OSError, ValueError):                      # a pipe was closed under us (its helper was replaced): look again
            pass

    def submit(self, points):
        """Start Delaunay(points) in a helper; -> ticket with .result().  points: (n, 2) float64."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        while True:
            with self.lock:
                if self.n <= 0:
                    return _Ticket(self, None, pts)
                # a started helper with nothing in flight is preferred (a fresh one needs ~0.3 s to import scipy); a new helper is
                # started only while every running one is busy; when all n are busy the first whose answer has ARRIVED is drained
                # and reused -- nobody waits for Qhull while holding the lock, so the other threads' hand-overs and pick-ups go on
                for q in range(len(self.procs)):            # a helper that died while idle is replaced where it stood
                    if q not in self.pending and self.procs[q].poll() is not None:
                        self.procs[q] = self._spawn(q)
                idle = [q for q in range(len(self.procs)) if q not in self.pending and q not in self.writing]
                w = None
                if idle:
                    w = idle[0]
                elif len(self.procs) < self.n:
                    w = len(self.procs)
                    self.procs.append(self._spawn(w))
                else:
                    ready = self._readable([q for q in self.pending if q not in self.writing])
                    if ready:
                        w = min(ready, key=lambda q: (q - self.next) % self.n)       # in turn among the ready ones
                        self.next = w + 1
                if w is not None:
                    t, p = self._claim(w, pts)
                    break
                busy = [self.procs[q].stdout for q in self.pending if q not in self.writing]
            self._wait(busy, 0.25)
        # the ~200 KB of points go down the pipe WITHOUT the lock: a helper that is slow to read (or a pipe the kernel would not
        # enlarge) then stalls this thread only, not every other thread's hand-overs and pick-ups
        try:
            p.stdin.write(struct.pack("<qq", t.seq, len(pts)))
            if len(pts):
                p.stdin.write(memoryview(pts).cast("B"))  # the array's own bytes: no tobytes() copy, no concatenation
            p.stdin.flush()
            ok = True
        except (OSError, ValueError):
            ok = False
        with self.lock:
            self.writing.discard(w)
            if not ok and self.pending.get(w) is t:
                del self.pending[w]
                t.worker = None                          # helper is gone: this one is computed in-process on result()
        return t

    def _claim(self, w, pts):
        """Reserve helper w for these points (lock held); its previous answer, if still unread, has arrived and is read first.
        -> (ticket, the helper process to write the request to once the lock is released)."""
        if w in self.pending:                           # one request in flight per helper
            old = self.pending.pop(w)
            try:
                old._value = self._read(w, old)
            except Exception:                            # Qhull refused the OLD request's points: that ticket raises on its own result()
                old.worker = None
        self.seq += 1
        t = _Ticket(self, w, pts, self.seq)
        if self.procs[w].poll() is not None:            # the helper has died since its last request: start another
            self.procs[w] = self._spawn(w)
        self.pending[w] = t
        self.writing.add(w)
        return t, self.procs[w]

    def _retire(self, w):
        """Stop helper w (its stream can no longer be trusted, or it is gone) and put a fresh one in its place."""
        p = self.procs[w]
        try:
            p.kill()
            p.wait(timeout=5)
        except (OSError, subprocess.TimeoutExpired):
            pass
        for f in (p.stdin, p.stdout):
            try:
                f.close()
            except (OSError, ValueError):
                pass
        self.procs[w] = self._spawn(w)

    def _read(self, w, ticket):
        """The answer to `ticket` from helper w.  Anything but a well-formed reply to exactly this request -- wrong magic, another
        request's number, an implausible count, a short read -- means the stream is out of step: the helper is replaced before it
        is used again (a later request must never read leftover bytes as simplices) and the call is made here instead."""
        p = self.procs[w]
        n = None
        try:
            head = p.stdout.read(20)
            if len(head) == 20:
                magic, seq, n = struct.unpack("<iqq", head)
                # a planar triangulation has < 2n triangles
                if magic == _MAGIC and seq == ticket.seq and 0 <= n <= 4 * len(ticket.points) + 16:
                    raw = p.stdout.read(12 * n)
                    if len(raw) == 12 * n:
                        return np.frombuffer(raw, np.int32).reshape(n, 3).copy()
                elif magic == _MAGIC and seq == ticket.seq and n == -1:
                    return _delaunay_here(ticket.points)     # Qhull refused these points: the same call here raises the same error
        except OSError:
            pass
        self._retire(w)
        return _delaunay_here(ticket.points)

    def _collect(self, ticket):
        if ticket.worker is None:
            return _delaunay_here(ticket.points)
        while True:
            with self.lock:
                if self.pending.get(ticket.worker) is not ticket:
                    break                                  # a later hand-over to the same helper has read this answer already
                if self._readable([ticket.worker]):
                    del self.pending[ticket.worker]
                    return self._read(ticket.worker, ticket)
                f = self.procs[ticket.worker].stdout
            self._wait([f], 0.25)                          # Qhull is still at it: wait without the lock
        return ticket._value if ticket._value is not None else _delaunay_here(ticket.points)

    def close(self):
        with self.lock:
            for p in self.procs:
                try:
                    p.stdin.write(struct.pack("<qq", 0, -1))
                    p.stdin.flush()
                    p.stdin.close()
                except (OSError, ValueError):
                    pass
            for p in self.procs:
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    p.kill()
            self.procs, self.pending, self.writing = [], {}, set()


_pool = None
_pool_lock = threading.Lock()


def default_workers():
    v = os.environ.get("SAME_QHULL_WORKERS")
    if v is not None:
        return max(0, int(v))
    b = cpu_budget()
    return max(1, min(24, (3 * b) // (2 * cpu_sharers()[0])))


def warm(count=None):
    """Start up to `count` helpers now (default: all), so that their interpreter start-up is not paid by the first requests."""
    p = pool()
    with p.lock:
        while len(p.procs) < min(p.n, p.n if count is None else int(count)):
            p.procs.append(p._spawn(len(p.procs)))


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
