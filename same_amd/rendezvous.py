"""Host control plane for the ranks of one node, in plain Python (no torch, no MPI).

One process per GPU needs very little from the host side: hand the 128-byte RCCL unique id from
rank 0 to the others, a barrier and a max-reduce around timed regions, and now and then an
exchange of small host objects (per-window match tables).  `HostGroup` does that over loopback
TCP as a star: rank 0 listens on an ephemeral port of 127.0.0.1 and publishes `{port, token}` in
a rendezvous directory; the other ranks poll for that file and connect.  Every collective is one
round trip through rank 0 (gather in rank order, answer to all), so results are ordered and
identical on every rank.

How ranks find the directory (`default_rdv_dir`):
  * `SAME_RDV_DIR` if the launcher set it (bench.py's own launcher makes a fresh temp dir);
  * otherwise a path under the temp dir keyed by MASTER_PORT, the parent's PID and the parent's
    start time -- under `python -m torch.distributed.run` every worker is a child of the same
    agent process, whose TCP store already occupies MASTER_PORT itself, so the ranks meet on a
    port of their own.

The data path never goes through here on GPUs: pruned candidate lists and sweep flags travel by
RCCL (csrc/comm.hip).  `allgather_array` exists for CPU tests and as a transport (never compute)
fallback when the RCCL communicator cannot be created.
"""
import io
import json
import os
import secrets
import stat
import socket
import struct
import tempfile
import threading
import time

import numpy as np

_HDR = struct.Struct("<IQ")  # sequence number, payload length


def _proc_start_time(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]  # field 22: starttime in clock ticks
    except (OSError, IndexError):
        return "0"


def default_rdv_dir():
    d = os.environ.get("SAME_RDV_DIR")
    if d:
        return d
    ppid = os.getppid()
    key = f"{os.environ.get('MASTER_PORT', '0')}_{ppid}_{_proc_start_time(ppid)}"
    return os.path.join(tempfile.gettempdir(), f"same_rdv_{os.getuid()}_{key}")


def _check_private(path, what, want_dir):
    """The rendezvous directory and the file in it are this job's only credentials: refuse anything this user does not own
    outright.  A directory someone else made first (the default path is predictable), a symlink, or group / world access bits
    would let another local user read the token or point the ranks at a port of their own."""
    st = os.lstat(path)
    if stat.S_ISLNK(st.st_mode) or (want_dir and not stat.S_ISDIR(st.st_mode)) or (not want_dir and not stat.S_ISREG(st.st_mode)):
        raise PermissionError(f"{what} {path} is not a plain {'directory' if want_dir else 'file'}: refusing to use it")
    if st.st_uid != os.getuid():
        raise PermissionError(f"{what} {path} belongs to uid {st.st_uid}, not to this user ({os.getuid()}): refusing to use it")
    if st.st_mode & 0o077:
        raise PermissionError(f"{what} {path} is accessible to other users (mode {stat.S_IMODE(st.st_mode):o}): refusing to use it")


# ---- small host objects, without pickle -------------------------------------------------------------------------------
# What the ranks exchange (dicts of numbers and strings, lists, numpy arrays, the per-window match tables as pandas frames)
# travels as a JSON skeleton plus a binary section: numeric arrays are appended as .npy bytes (read back with
# allow_pickle=False) and the skeleton refers to them by position -- no base64, no megabyte strings inside JSON, and nothing
# a peer sends can name a callable, so a channel that was somehow reached by someone else still cannot execute code in a rank.
def _enc(o, blobs):
    import numpy as np

    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    if isinstance(o, np.generic):
        return _enc(o.item(), blobs)
    if isinstance(o, bytes):
        blobs.append(o)
        return {"__b__": len(blobs) - 1}
    if isinstance(o, np.ndarray):
        if o.dtype.kind in "biufc":
            buf = io.BytesIO()
            np.save(buf, np.ascontiguousarray(o), allow_pickle=False)
            blobs.append(buf.getvalue())
            return {"__nd__": len(blobs) - 1}
        return {"__ndo__": [_enc(x, blobs) for x in o.reshape(-1).tolist()], "shape": list(o.shape),
                "dtype": o.dtype.str if o.dtype.kind in "US" else "O"}
    if isinstance(o, tuple):
        return {"__t__": [_enc(x, blobs) for x in o]}
    if isinstance(o, (list, set, frozenset)):
        return [_enc(x, blobs) for x in o] if isinstance(o, list) else {"__s__": [_enc(x, blobs) for x in sorted(o, key=repr)]}
    if isinstance(o, dict):
        return {"__d__": [[_enc(k, blobs), _enc(v, blobs)] for k, v in o.items()]}
    try:
        import pandas as pd
    except ImportError:   # pragma: no cover
        pd = None
    if pd is not None and isinstance(o, pd.DataFrame):
        return {"__df__": {"columns": [_enc(c, blobs) for c in o.columns], "data": [_enc(o[c].to_numpy(), blobs) for c in o.columns],
                           "index": _enc(o.index.to_numpy(), blobs)}}
    raise TypeError(f"allgather_object cannot carry a {type(o).__name__} (numbers, strings, lists, dicts, numpy arrays and pandas frames "
                    f"only)")


def _dec(o, blobs):
    import numpy as np

    if isinstance(o, list):
        return [_dec(x, blobs) for x in o]
    if not isinstance(o, dict):
        return o
    if "__nd__" in o:
        return np.load(io.BytesIO(blobs[int(o["__nd__"])]), allow_pickle=False)
    if "__ndo__" in o:
        vals = [_dec(x, blobs) for x in o["__ndo__"]]
        if o["dtype"] != "O":
            return np.array(vals, dtype=o["dtype"]).reshape(o["shape"])
        a = np.empty(len(vals), dtype=object)
        for i, v in enumerate(vals):
            a[i] = v
        return a.reshape(o["shape"])
    if "__b__" in o:
        return bytes(blobs[int(o["__b__"])])
    if "__t__" in o:
        return tuple(_dec(x, blobs) for x in o["__t__"])
    if "__s__" in o:
        return set(_dec(x, blobs) for x in o["__s__"])
    if "__d__" in o:
        return {_dec(k, blobs): _dec(v, blobs) for k, v in o["__d__"]}
    if "__df__" in o:
        import pandas as pd

        d = o["__df__"]
        cols = [_dec(c, blobs) for c in d["columns"]]
        idx = _dec(d["index"], blobs)
        # column by column, so every column keeps the dtype its array travelled with
        return pd.DataFrame({i: pd.Series(_dec(a, blobs), index=idx) for i, a in enumerate(d["data"])}, index=idx).set_axis(cols, axis=1) \
            if cols else pd.DataFrame(index=idx)
    raise ValueError("malformed object frame from a peer")


def pack_object(obj):
    """obj -> bytes: u64 skeleton length, JSON skeleton, then for every binary part u64 length + bytes."""
    blobs = []
    head = json.dumps(_enc(obj, blobs)).encode()
    return b"".join([struct.pack("<Q", len(head)), head] + [part for blob in blobs for part in (struct.pack("<Q", len(blob)), blob)])


def unpack_object(payload):
    view = memoryview(payload)
    (n,) = struct.unpack_from("<Q", view, 0)
    head, off, blobs = json.loads(bytes(view[8: 8 + n]).decode()), 8 + n, []
    while off < len(view):
        (m,) = struct.unpack_from("<Q", view, off)
        if off + 8 + m > len(view):
            raise ValueError("truncated object frame from a peer")
        blobs.append(view[off + 8: off + 8 + m])
        off += 8 + m
    return _dec(head, blobs)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        r = sock.recv_into(view[got:], n - got)
        if r == 0:
            raise ConnectionError("peer closed the rendezvous connection")
        got += r
    return bytes(buf)


def _send_frame(sock, seq, payload):
    sock.sendall(_HDR.pack(seq, len(payload)))
    if payload:
        sock.sendall(payload)


def _recv_frame(sock, seq):
    got_seq, n = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if got_seq != seq:
        raise RuntimeError(f"rendezvous out of step: expected collective #{seq}, peer sent #{got_seq}")
    return _recv_exact(sock, n) if n else b""


class HostGroup:
    """rank/world as given (or from RANK / WORLD_SIZE); world 1 needs no sockets."""

    _made = {}       # (rendezvous directory, rank) -> groups this process has made there so far
    _made_lock = threading.Lock()

    def __init__(self, rank=None, world=None, rdv_dir=None, timeout=600.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        if not (0 <= self.rank < self.world):
            raise ValueError(f"rank {self.rank} outside world {self.world}")
        self.timeout = float(timeout)
        self._seq = 0
        self._peers = {}    # rank 0: rank -> socket
        self._hub = None    # other ranks: socket to rank 0
        self._listener = None
        self._dir = None
        if self.world == 1:
            return
        self._dir = rdv_dir or default_rdv_dir()
        # One rendezvous file per group a process makes in this directory, numbered in the order they are made (the ranks of a job make
        # their groups in the same order): a rank that is already starting its next group while rank 0 has not yet closed the last one
        # must not read the LAST group's file -- it would say hello to a listener that is about to go away and wait for an answer.
        with HostGroup._made_lock:
            nth = HostGroup._made.get((self._dir, self.rank), 0)        # (per rank: tests run several ranks of a group as threads of one process)
            HostGroup._made[(self._dir, self.rank)] = nth + 1
        self._hub_name = "hub.json" if nth == 0 else f"hub.{nth}.json"
        hub_file = os.path.join(self._dir, self._hub_name)
        if self.rank == 0:
            os.makedirs(self._dir, mode=0o700, exist_ok=True)
            if os.lstat(self._dir).st_uid == os.getuid():
                os.chmod(self._dir, 0o700)       # the token below is this job's only credential: keep it to this user (a failure raises)
            _check_private(self._dir, "rendezvous directory", want_dir=True)
            token = secrets.token_hex(16)
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.bind(("127.0.0.1", 0))
            ls.listen(self.world)
            ls.settimeout(self.timeout)
            self._listener = ls
            tmp = hub_file + f".{os.getpid()}.tmp"
            with os.fdopen(os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600), "w") as f:
                json.dump({"port": ls.getsockname()[1], "token": token, "world": self.world}, f)
            os.replace(tmp, hub_file)  # atomic: readers see nothing or the whole file
            while len(self._peers) < self.world - 1:
                conn, _ = ls.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(5.0)      # a rank says hello at once; something that connects and stalls must not hold up the others
                try:
                    hello = _recv_exact(conn, 36)
                    peer = struct.unpack("<I", hello[32:])[0]
                except (OSError, ConnectionError):
                    conn.close()
                    continue
                if hello[:32] != token.encode() or not (0 < peer < self.world) or peer in self._peers:
                    conn.close()  # not one of this job's ranks
                    continue
                conn.settimeout(self.timeout)
                self._peers[peer] = conn
        else:
            deadline = time.monotonic() + self.timeout
            info = None
            while info is None:
                try:
                    if not os.path.lexists(hub_file):     # rank 0 writes it after it has made the directory private
                        raise FileNotFoundError(hub_file)
                    _check_private(self._dir, "rendezvous directory", want_dir=True)   # PermissionError is not retried: it propagates
                    _check_private(hub_file, "rendezvous file", want_dir=False)
                    with open(hub_file) as f:
                        info = json.load(f)
                except PermissionError:
                    raise
                except (OSError, ValueError):
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rank {self.rank}: no rendezvous file {hub_file} after {self.timeout:.0f} s")
                    time.sleep(0.02)
            if info.get("world") != self.world:
                raise RuntimeError(f"rendezvous file is for world {info.get('world')}, this rank expects {self.world}")
            s = socket.create_connection(("127.0.0.1", int(info["port"])), timeout=self.timeout)
            s.settimeout(self.timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(info["token"].encode() + struct.pack("<I", self.rank))
            self._hub = s
        self.barrier()

    # -- the one primitive: ordered all-gather of byte strings ---------------------------------
    def allgather_bytes(self, payload):
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        seq = self._seq = (self._seq + 1) & 0xFFFFFFFF
        if self.rank == 0:
            parts = [payload] + [_recv_frame(self._peers[r], seq) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for r in range(1, self.world):
                _send_frame(self._peers[r], seq, blob)
            return parts
        _send_frame(self._hub, seq, payload)
        blob, parts, off = _recv_frame(self._hub, seq), [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8: off + 8 + n])
            off += 8 + n
        return parts

    def barrier(self):
        self.allgather_bytes(b"")

    def bcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b"")[src]

    def max(self, v):
        return max(struct.unpack("<d", p)[0] for p in self.allgather_bytes(struct.pack("<d", float(v))))

    def min(self, v):
        return min(struct.unpack("<d", p)[0] for p in self.allgather_bytes(struct.pack("<d", float(v))))

    def sum_int(self, values):
        """Element-wise sum of a short int64 vector over ranks (sweep counters)."""
        v = np.ascontiguousarray(values, dtype=np.int64)
        return np.sum([np.frombuffer(p, np.int64) for p in self.allgather_bytes(v.tobytes())], axis=0).reshape(v.shape)

    def allgather_array(self, arr):
        """Concatenate equal-shaped arrays along axis 0 in rank order."""
        a = np.ascontiguousarray(arr)
        parts = [np.frombuffer(p, a.dtype).reshape((-1,) + a.shape[1:]) for p in self.allgather_bytes(a.tobytes())]
        return np.concatenate(parts, axis=0)

    def allgather_object(self, obj):
        """Small host objects between this job's own ranks: numbers, strings, lists / tuples / sets / dicts of them, numpy
        arrays and pandas frames (the per-window match tables), as a JSON skeleton + .npy bytes -- no pickle, so nothing a
        peer sends can run code here.  The connections are loopback-only and admitted with the job's token, which sits in a 0600 file
        of a 0700 directory whose ownership every rank checks."""
        if self.world == 1:
            return [obj]
        return [unpack_object(p) for p in self.allgather_bytes(pack_object(obj))]

    def close(self):
        for s in list(self._peers.values()) + [self._hub, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._hub, self._listener = {}, None, None
        if self.rank == 0 and self._dir:
            try:
                os.remove(os.path.join(self._dir, self._hub_name))
                os.rmdir(self._dir)
            except OSError:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
