"""Multi-GPU forms of the path: one process per GPU, aligned-row blocks for the cost build + prune,
triangle blocks for the violation sweeps, whole windows for the sliding-window plan.

SURVEY 8e: every kernel on the path is a map over aligned rows / triangles / windows with read-only
shared inputs, so ranks own contiguous blocks (refs, match vector and ref coordinates replicated, no
merge step) and the exchanges are
  * an all-gather of the fixed-width pruned candidate lists -- int32 idx[rows][k] (-1 padded) and
    float64 cost[rows][k] (+inf padded);
  * for the sweeps (src/same.py:645-669, src/violationhelper.py:53-117, src/same.py:1362-1402): an
    all-gather of the per-triangle flag blocks plus an all-reduce of the counters (integer sums) and
    of the per-point flags (OR).
On GPUs both run over RCCL on the context's stream (`RcclGroup`, csrc/comm.hip); `HostGroup`
(rendezvous.py: loopback TCP in plain Python) carries the same exchanges for host arrays -- that is
the CPU-test transport and a transport-only fallback, never a compute fallback.  Results are
identical for any world size: blocks concatenate in rank order, and the ascending flipped-triangle
list (src/same.py:687-703 depends on that order) is rebuilt from the complete flag array.
"""
import ctypes

import numpy as np

from . import _lib
from .knn import compact_pairs, pairs_from_padded
from .rendezvous import HostGroup  # noqa: F401  (re-exported: the host transport of this module)

TRI_BLOCK_ALIGN = 256  # triangle blocks start on whole-workgroup boundaries (and whole 64-bit flag-mask words)


def row_block(n_rows, world, rank):
    """Equal-width blocks (the last ones may be short or empty): -> (begin, end, block_rows)."""
    block = -(-n_rows // world) if world > 0 else n_rows
    begin = min(rank * block, n_rows)
    return begin, min(begin + block, n_rows), block


def tri_block(n_tri, world, rank):
    """Triangle blocks for the sharded sweeps: equal width, a multiple of TRI_BLOCK_ALIGN, so block r starts at r*block in
    the gathered arrays and a triangle's position there is its index.  -> (begin, end, block)."""
    block = -(-max(n_tri, 1) // world)
    block = -(-block // TRI_BLOCK_ALIGN) * TRI_BLOCK_ALIGN
    begin = min(rank * block, n_tri)
    return begin, min(begin + block, n_tri), block


class RcclGroup:
    """RCCL communicator bound to a Context.  `exchange_id(bytes_or_None) -> bytes` is any host broadcast from rank 0
    (HostGroup.bcast_bytes; a launcher with MPI or a file can pass its own)."""

    synchronous = False   # collectives are enqueued on the context's streams

    def __init__(self, ctx, world, rank, exchange_id):
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        uid = None
        if self.rank == 0:
            buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
            # a failure here still goes through the exchange (as an empty id), so that every rank sees it and takes the same
            # way out instead of waiting for a broadcast that never comes
            uid = buf.raw if ctx.lib.same_comm_unique_id(buf) == 0 else b""
        uid = exchange_id(uid)
        if uid is None or len(uid) != _lib.UNIQUE_ID_BYTES:
            raise _lib.SameHipError(_lib.SAME_EIO, "no RCCL unique id (rank 0 could not create one, or the exchange lost it)")
        ctx.check(ctx.lib.same_comm_init(ctx.handle, self.world, self.rank, uid), "same_comm_init")

    def rccl_version(self):
        return self.info()["version"]

    def info(self):
        """What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) and the RCCL version."""
        n, r, v, d = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(-1)
        c = self.ctx
        c.check(c.lib.same_comm_info(c.handle, ctypes.byref(n), ctypes.byref(r), ctypes.byref(v)), "same_comm_info")
        c.check(c.lib.same_comm_device(c.handle, ctypes.byref(d)), "same_comm_device")
        return {"nranks": n.value, "rank": r.value, "device": d.value, "version": v.value}

    def gather_time(self):
        """(ms, bytes this rank sent) of the all-gathers since the last wait(): HIP events on the stream they ran on."""
        ms, nb = ctypes.c_float(0), ctypes.c_int64(0)
        c = self.ctx
        c.check(c.lib.same_comm_gather_time(c.handle, ctypes.byref(ms), ctypes.byref(nb)), "same_comm_gather_time")
        return ms.value, nb.value

    def allgather_dev(self, send_buf, recv_buf, send_bytes, send_offset=0):
        c = self.ctx
        c.check(c.lib.same_allgather_dev(c.handle, send_buf.ptr + send_offset, recv_buf.ptr, send_bytes), "same_allgather_dev")

    def allgather_dev_async(self, send_buf, recv_buf, send_bytes):
        """Gather on the context's communication stream, overlapping whatever compute is queued next."""
        c = self.ctx
        c.check(c.lib.same_allgather_dev_async(c.handle, send_buf.ptr, recv_buf.ptr, send_bytes), "same_allgather_dev_async")

    def allreduce_dev(self, buf, count, dtype, op):
        c = self.ctx
        c.check(c.lib.same_allreduce_dev(c.handle, buf.ptr, int(count), int(dtype), int(op)), "same_allreduce_dev")

    def wait(self):
        """Order the compute stream after every gather issued so far (stream-side; the host does not block)."""
        self.ctx.check(self.ctx.lib.same_comm_wait(self.ctx.handle), "same_comm_wait")

    def fused(self):
        """`with comm.fused():` -- the collectives issued inside go to RCCL as one group (one fused launch)."""
        return _RcclGroupScope(self.ctx)

    def close(self):
        self.ctx.lib.same_comm_destroy(self.ctx.handle)


class _RcclGroupScope:
    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.ctx.check(self.ctx.lib.same_comm_group_start(self.ctx.handle), "same_comm_group_start")

    def __exit__(self, *exc):
        self.ctx.check(self.ctx.lib.same_comm_group_end(self.ctx.handle), "same_comm_group_end")
        return False


class _NoGroup:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


RcclGather = RcclGroup  # round-1 name


class HostTransport:
    """RcclGroup's interface for device buffers, carried by a HostGroup instead (D2H, loopback TCP all-gather, H2D).
    A TRANSPORT fallback for when the RCCL communicator cannot be created (e.g. several ranks sharing one GPU: RCCL refuses
    duplicate devices) -- compute stays on the GPU; synchronous, so nothing overlaps.  bench.py reports it when used."""

    synchronous = True    # every exchange has completed when its call returns

    def __init__(self, ctx, group):
        self.ctx, self.group, self.world, self.rank = ctx, group, group.world, group.rank
        self._ms, self._bytes = 0.0, 0

    def allgather_dev(self, send_buf, recv_buf, send_bytes, send_offset=0):
        import time

        t0 = time.perf_counter()
        host = send_buf.download((send_bytes,), np.uint8, offset_bytes=send_offset)
        recv_buf.upload(self.group.allgather_array(host))
        self._ms += (time.perf_counter() - t0) * 1e3
        self._bytes += send_bytes

    def info(self):
        return {"nranks": self.world, "rank": self.rank, "device": self.ctx.device, "version": 0}

    def gather_time(self):
        """(ms, bytes) of the gathers since the last wait(): host wall time (D2H + loopback TCP + H2D, synchronous)."""
        return self._ms, self._bytes

    def allgather_dev_async(self, send_buf, recv_buf, send_bytes):
        self.allgather_dev(send_buf, recv_buf, send_bytes)

    def allreduce_dev(self, buf, count, dtype, op):
        dt = {_lib.DT_U8: np.uint8, _lib.DT_I32: np.int32, _lib.DT_U64: np.uint64, _lib.DT_F64: np.float64}[dtype]
        parts = self.group.allgather_array(buf.download((1, count), dt))
        red = {_lib.OP_SUM: parts.sum(axis=0, dtype=dt), _lib.OP_MAX: parts.max(axis=0), _lib.OP_MIN: parts.min(axis=0)}[op]
        buf.upload(np.ascontiguousarray(red, dtype=dt))

    def wait(self):
        self._ms, self._bytes = 0.0, 0

    def close(self):
        pass




# ---- cost build + prune over aligned-row blocks ----------------------------------------------------
def hip_block_compute(ctx, A, R, axy, rxy, radius, knn, w):
    """Per-rank compute on the GPU: resident operands, prune + padded costs for one row block."""
    A, R = _lib.as_c(A, np.float64), _lib.as_c(R, np.float64)
    axy, rxy = _lib.as_c(axy, np.float64), _lib.as_c(rxy, np.float64)
    dA, dR, dax, drx = ctx.to_device(A), ctx.to_device(R), ctx.to_device(axy), ctx.to_device(rxy)
    T = A.shape[1]

    def compute(row_begin, row_end, block_rows):
        rows = row_end - row_begin
        didx, dcost, dcnt = ctx.alloc(block_rows * knn * 4), ctx.alloc(block_rows * knn * 8), ctx.alloc(max(block_rows, 1) * 4)
        ctx.check(ctx.lib.same_dev_memset(ctx.handle, didx.ptr, 0xFF, didx.nbytes), "memset")  # -1 padding rows
        ctx.check(ctx.lib.same_knn_prune_dev(ctx.handle, dax.ptr, drx.ptr, len(rxy), row_begin, row_end, float(radius),
                                             int(knn), didx.ptr, None, dcnt.ptr), "same_knn_prune_dev")
        ctx.check(ctx.lib.same_padded_cost_f64_dev(ctx.handle, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, row_begin,
                                                   row_begin + block_rows, int(knn), didx.ptr, float(w), dcost.ptr),
                  "same_padded_cost_f64_dev")
        return didx, dcost, rows

    return compute


def sharded_knn_cost_host(compute_block, n_aligned, knn, group):
    """Host-array form: compute_block(begin, end) -> (idx (rows,k) int32, cost (rows,k) f64);
    returns the gathered, unpadded (idx, cost) for all n_aligned rows on every rank."""
    begin, end, block = row_block(n_aligned, group.world, group.rank)
    idx_p = np.full((block, knn), -1, np.int32)
    cost_p = np.full((block, knn), np.inf, np.float64)
    if end > begin:
        idx, cost = compute_block(begin, end)
        idx_p[: end - begin], cost_p[: end - begin] = idx, cost
    return group.allgather_array(idx_p)[:n_aligned], group.allgather_array(cost_p)[:n_aligned]


def sharded_knn_cost_device(ctx, compute, n_aligned, knn, comm):
    """Device form: prune + cost for this rank's block, RCCL all-gather, one D2H of the result."""
    begin, end, block = row_block(n_aligned, comm.world, comm.rank)
    didx, dcost, _ = compute(begin, end, block)
    gidx, gcost = ctx.alloc(block * knn * 4 * comm.world), ctx.alloc(block * knn * 8 * comm.world)
    comm.allgather_dev(didx, gidx, block * knn * 4)
    comm.allgather_dev(dcost, gcost, block * knn * 8)
    idx = gidx.download((block * comm.world, knn), np.int32)[:n_aligned]
    cost = gcost.download((block * comm.world, knn), np.float64)[:n_aligned]
    return idx, cost


def pairs_and_costs(aligned_df, ref_df, idx, cost):
    """Gathered padded lists -> compacted frames, valid_pairs and c, exactly as the single-GPU
    path produces them (src/utils.py:731-742 order)."""
    knn_pairs = pairs_from_padded(idx)
    rows, cols = np.nonzero(idx >= 0)
    c = list(cost[rows, cols])
    new_a, new_r, new_pairs = compact_pairs(aligned_df, ref_df, knn_pairs)
    return new_a, new_r, new_pairs, c


# ---- violation sweeps over triangle blocks -----------------------------------------------------------
def _gather_blocks(group, part, block, n, width=None):
    """Pad this rank's rows to `block`, all-gather, cut to n."""
    shape = (block,) if width is None else (block, width)
    padded = np.zeros(shape, part.dtype)
    padded[: len(part)] = part
    return group.allgather_array(padded)[:n]


def sharded_orient_sweep_host(compute_flags, n_tri, group):
    """Lazy-constraint orientation sweep (src/same.py:645-669) over triangle blocks.
    compute_flags(t0, t1) -> uint8 flag per triangle of [t0, t1) (0 skipped, 1 checked, 2 flipped).
    -> (checked, violating triangle indices ascending, flags of all triangles), identical on every rank."""
    t0, t1, block = tri_block(n_tri, group.world, group.rank)
    mine = np.asarray(compute_flags(t0, t1), np.uint8) if t1 > t0 else np.zeros(0, np.uint8)
    flags = _gather_blocks(group, mine, block, n_tri)
    checked = int(group.sum_int([np.count_nonzero(mine)])[0])
    return checked, np.flatnonzero(flags == 2).astype(np.int32), flags


def sharded_xyorder_sweep_host(compute_block, n_tri, n_aligned, group):
    """XY-order sweep (src/violationhelper.py:53-117) over triangle blocks.
    compute_block(t0, t1) -> (edge_flags (n,3) u8, tri_flag (n,) u8, point_flag (n_aligned,) u8, counts (3,) int64).
    -> the same four for the whole triangle list on every rank."""
    t0, t1, block = tri_block(n_tri, group.world, group.rank)
    if t1 > t0:
        edge, tflag, pflag, counts = compute_block(t0, t1)
    else:
        edge, tflag = np.zeros((0, 3), np.uint8), np.zeros(0, np.uint8)
        pflag, counts = np.zeros(n_aligned, np.uint8), np.zeros(3, np.int64)
    edge_all = _gather_blocks(group, np.asarray(edge, np.uint8).reshape(-1, 3), block, n_tri, 3)
    tflag_all = _gather_blocks(group, np.asarray(tflag, np.uint8), block, n_tri)
    pflag_all = group.allgather_array(np.asarray(pflag, np.uint8).reshape(1, -1)).max(axis=0)  # OR over ranks
    return edge_all, tflag_all, pflag_all, group.sum_int(counts)


def sharded_area_flip_host(compute_block, n_tri, group):
    """Signed areas before / after and flips (src/same.py:1362-1402) over triangle blocks.
    compute_block(t0, t1) -> (before f64, after f64, matched3 (n,3) u8, flipped u8)."""
    t0, t1, block = tri_block(n_tri, group.world, group.rank)
    if t1 > t0:
        before, after, m3, fl = compute_block(t0, t1)
    else:
        before, after, m3, fl = np.zeros(0), np.zeros(0), np.zeros((0, 3), np.uint8), np.zeros(0, np.uint8)
    return (_gather_blocks(group, np.asarray(before, np.float64), block, n_tri),
            _gather_blocks(group, np.asarray(after, np.float64), block, n_tri),
            _gather_blocks(group, np.asarray(m3, np.uint8).reshape(-1, 3), block, n_tri, 3),
            _gather_blocks(group, np.asarray(fl, np.uint8), block, n_tri))


def sharded_verify_spatial_preservation(aligned_df, ref_df, matches_df, triangle_info, group, block_sweep=None, ctx=None):
    """verify_spatial_preservation (src/violationhelper.py:1-134) with the triangle loop split over the ranks of `group`;
    every rank returns the single-process report.  block_sweep(axy, rxy, tris_block, match) defaults to the HIP sweep."""
    from . import ops, sweeps

    fn = block_sweep or (lambda a, r, t, m: ops.xyorder_sweep(a, r, t, m, ctx=ctx))

    def sweep(axy, rxy, tris, match):
        return sharded_xyorder_sweep_host(lambda t0, t1: fn(axy, rxy, tris[t0:t1], match), len(tris), len(axy), group)

    return sweeps.verify_spatial_preservation(aligned_df, ref_df, matches_df, triangle_info, _sweep=sweep)


def sharded_triangle_area_flips(aligned_df, ref_df, aligned_delaunay, aligned_to_ref, group, block_sweep=None, ctx=None):
    """triangle_area_flips (src/same.py:1355-1402) over triangle blocks; every rank returns the single-process result."""
    from . import ops, sweeps

    fn = block_sweep or (lambda a, r, t, m: ops.area_flip(a, r, t, m, ctx=ctx))

    def sweep(axy, rxy, tris, match):
        return sharded_area_flip_host(lambda t0, t1: fn(axy, rxy, tris[t0:t1], match), len(tris), group)

    return sweeps.triangle_area_flips(aligned_df, ref_df, aligned_delaunay, aligned_to_ref, _sweep=sweep)


class ShardedSweeps:
    """Device form: the three sweeps of ONE problem over this rank's triangle block, exchanged over RCCL.

    Operands are resident and replicated on every rank (aligned / ref XY, triangles, match vector -- a few MB); outputs
    are complete on every rank after `run()`.  All work is enqueued on the context's stream; only the orientation
    sweep's two counters and its flipped list come back to the host (as in the single-GPU sweep)."""

    def __init__(self, ctx, comm, sweep_handle, daxy, drxy, dtris, n_tri, n_aligned):
        self.ctx, self.comm, self.sweep = ctx, comm, sweep_handle
        self.daxy, self.drxy, self.dtris, self.Tr, self.n_m = daxy, drxy, dtris, int(n_tri), int(n_aligned)
        world = comm.world
        self.t0, self.t1, self.block = tri_block(self.Tr, world, comm.rank)
        full = self.block * world
        a = ctx.alloc
        # this rank's blocks live at their absolute positions inside full-size arrays; gathers go into second arrays
        self.flag_l, self.flag_g = a(full), a(full)
        self.edge_l, self.edge_g = a(full * 3), a(full * 3)
        self.tflag_l, self.tflag_g = a(full), a(full)
        self.m3_l, self.m3_g = a(full * 3), a(full * 3)
        self.flip_l, self.flip_g = a(full), a(full)
        self.before_l, self.before_g = a(full * 8), a(full * 8)
        self.after_l, self.after_g = a(full * 8), a(full * 8)
        self.pflag, self.counts = a(max(self.n_m, 1)), a(32)
        for b in (self.flag_l, self.edge_l, self.tflag_l, self.m3_l, self.flip_l, self.before_l, self.after_l):
            ctx.check(ctx.lib.same_dev_memset(ctx.handle, b.ptr, 0, b.nbytes), "memset")  # padding past the last triangle
        self.viol = np.empty(max(self.Tr, 1), np.int32)

    def run(self, dmatch):
        """Enqueue all three sweeps for this rank's block + the exchanges; -> (checked, violating idx ascending)."""
        c, L, H = self.ctx, self.ctx.lib, self.ctx.handle
        t0, n, B = self.t0, self.t1 - self.t0, self.block
        off = self.comm.rank * B   # == t0 unless this rank's block is empty
        chk = c.check
        chk(L.same_orient_flags_dev(self.sweep, dmatch.ptr, t0, self.t1, self.flag_l.ptr), "same_orient_flags_dev")
        chk(L.same_xyorder_sweep_dev(H, self.daxy.ptr, self.n_m, self.drxy.ptr, self.dtris.ptr + 12 * t0, n, dmatch.ptr,
                                     self.edge_l.ptr + 3 * off, self.tflag_l.ptr + off, self.pflag.ptr, self.counts.ptr), "xyorder")
        chk(L.same_area_flip_dev(H, self.daxy.ptr, self.drxy.ptr, self.dtris.ptr + 12 * t0, n, dmatch.ptr,
                                 self.before_l.ptr + 8 * off, self.after_l.ptr + 8 * off, self.m3_l.ptr + 3 * off,
                                 self.flip_l.ptr + off), "area_flip")
        g = self.comm
        grouped = getattr(g, "fused", _NoGroup)                      # a host transport has nothing to fuse
        with grouped():    # seven arrays, one fused RCCL launch
            for loc, glob, width in ((self.flag_l, self.flag_g, 1), (self.edge_l, self.edge_g, 3), (self.tflag_l, self.tflag_g, 1),
                                     (self.m3_l, self.m3_g, 3), (self.flip_l, self.flip_g, 1), (self.before_l, self.before_g, 8),
                                     (self.after_l, self.after_g, 8)):
                g.allgather_dev(loc, glob, B * width, send_offset=off * width)
        with grouped():
            g.allreduce_dev(self.counts, 3, _lib.DT_U64, _lib.OP_SUM)
            g.allreduce_dev(self.pflag, self.n_m, _lib.DT_U8, _lib.OP_MAX)
        checked, nviol = ctypes.c_int64(0), ctypes.c_int64(0)
        chk(L.same_orient_from_flags_dev(self.sweep, self.flag_g.ptr, ctypes.byref(checked), self.viol.ctypes.data,
                                         ctypes.byref(nviol)), "same_orient_from_flags_dev")
        return checked.value, self.viol[: nviol.value].copy()

    def close(self):
        """Free the device blocks this object allocated (the bound sweep, the operands and the communicator are the caller's)."""
        for name in ("flag_l", "flag_g", "edge_l", "edge_g", "tflag_l", "tflag_g", "m3_l", "m3_g", "flip_l", "flip_g", "before_l",
                     "before_g",
                     "after_l", "after_g", "pflag", "counts"):
            buf = getattr(self, name, None)
            if buf is not None:
                buf.free()
                setattr(self, name, None)

    def download(self):
        """Complete outputs (any rank): dict of host arrays shaped like the single-GPU sweeps' outputs."""
        Tr = self.Tr
        return {"flag": self.flag_g.download((Tr,), np.uint8), "edge": self.edge_g.download((Tr, 3), np.uint8),
                "tri_flag": self.tflag_g.download((Tr,), np.uint8), "point_flag": self.pflag.download((self.n_m,), np.uint8),
                "counts": self.counts.download((3,), np.uint64).astype(np.int64),
                "before": self.before_g.download((Tr,), np.float64), "after": self.after_g.download((Tr,), np.float64),
                "matched3": self.m3_g.download((Tr, 3), np.uint8), "flipped": self.flip_g.download((Tr,), np.uint8)}


def _pack_table(cols, arrs, n):
    import struct

    return struct.pack("<Q", n) + b"".join(a.tobytes() for a in arrs)


def _unpack_table(raw, cols, arrs, rank):
    import struct

    (m,) = struct.unpack_from("<Q", raw, 0)
    off, part = 8, {}
    for c, a in zip(cols, arrs):
        part[c] = np.frombuffer(raw, a.dtype, m, off).copy()
        off += m * a.dtype.itemsize
    if off != len(raw):
        raise ValueError(f"table of rank {rank} does not have the agreed columns")
    return part


_last_table_gather = [0.0, 0]    # (ms on the gather's stream or host wall ms, bytes this rank sent) of the latest allgather_table


def last_table_gather():
    """(ms, bytes sent by this rank) of the latest allgather_table's collective: HIP events on its stream (RCCL), host wall time
    (host transport)."""
    return tuple(_last_table_gather)


def allgather_table(ctx, comm, group, table):
    """One exchange of per-rank tables (dict: column -> 1-D numeric array, all of one length, same columns / dtypes on every
    rank) as a DEVICE collective: the columns are packed into one byte block per rank, padded to the longest, all-gathered with
    `comm.allgather_dev` (RCCL over xGMI; `HostTransport` carries the same call where RCCL is not available) and unpacked.
    -> list of tables in rank order, identical on every rank.  This is the exchange step of the window configuration
    (BASELINE cfg 5): every rank's central-trimmed match table, once per pass.  comm None: one rank, nothing to exchange;
    ctx None (CPU tests): the same blocks through the host group's byte all-gather."""
    import struct

    cols = list(table)
    arrs = [np.ascontiguousarray(table[c]) for c in cols]
    n = len(arrs[0]) if arrs else 0
    if any(len(a) != n or a.ndim != 1 or a.dtype.kind not in "biuf" for a in arrs):
        raise ValueError("allgather_table carries equal-length 1-D numeric columns only")
    if comm is None and ctx is not None:
        return [dict(zip(cols, arrs))]
    blob = _pack_table(cols, arrs, n)
    if ctx is None:
        return [_unpack_table(raw, cols, arrs, r) for r, raw in enumerate(group.allgather_bytes(blob))]
    # host transport (no RCCL communicator): the block is host bytes already -- no device round trip
    if getattr(comm, "synchronous", False):
        import time

        t0 = time.perf_counter()
        parts = group.allgather_bytes(blob)
        _last_table_gather[:] = ((time.perf_counter() - t0) * 1e3, len(blob))
        return [_unpack_table(raw, cols, arrs, r) for r, raw in enumerate(parts)]
    sizes = [struct.unpack("<Q", p)[0] for p in group.allgather_bytes(struct.pack("<Q", len(blob)))]
    width = (max(sizes) + 255) & ~255
    send, recv = ctx.alloc(width), ctx.alloc(width * group.world)
    try:
        padded = np.zeros(width, np.uint8)
        padded[: len(blob)] = np.frombuffer(blob, np.uint8)
        send.upload(padded)
        comm.wait()                                # a fresh timing batch: gather_time() below is this exchange alone
        comm.allgather_dev(send, recv, width)
        ctx.sync()
        got = recv.download((group.world, width), np.uint8)
        _last_table_gather[:] = comm.gather_time()
    finally:
        send.free()
        recv.free()
    return [_unpack_table(got[r, : sizes[r]].tobytes(), cols, arrs, r) for r in range(group.world)]


# ---- sliding windows over ranks -------------------------------------------------------------------------
class MergeChannel:
    """What the per-rank window merge (merge.merged_part_rows) needs from the job: this rank's place, `tables(table)` = every rank's
    small table of seam rows (numeric columns: ONE device all-gather, `allgather_table`; anything else -- string cell ids -- through the
    host group's object exchange) and a maximum over ranks.  `ctx` / `comm` None: the host group carries the bytes (CPU tests)."""

    def __init__(self, group, ctx=None, comm=None):
        self.group, self.ctx, self.comm = group, ctx, comm
        self.rank, self.world = int(group.rank), int(group.world)
        self.sent_rows, self.gather_ms = 0, 0.0

    def tables(self, table):
        self.sent_rows += len(next(iter(table.values()))) if table else 0
        if all(np.asarray(v).dtype.kind in "biuf" for v in table.values()):
            out = allgather_table(self.ctx if self.comm is not None else None, self.comm, self.group, table)
            if self.comm is not None:
                self.gather_ms += last_table_gather()[0]
            return out
        return self.group.allgather_object(table)

    def max(self, v):
        return self.group.max(float(v))


def sharded_sliding_window_incumbent(ref, moving, commonCT=None, group=None, exchange=None, rank=None, world=None, deal="block",
                                     gather="all", **kwargs):
    """`sliding_window_incumbent` (same_amd/incumbent.py: the window loop with the greedy MIP start as every window's solution) on N
    GPUs, the way `sharded_sliding_window_matching` shards the solver loop."""
    from .incumbent import sliding_window_incumbent

    return _sharded_windows(sliding_window_incumbent, ref, moving, commonCT, group, exchange, rank, world, deal, gather, kwargs)


def sharded_sliding_window_matching(ref, moving, commonCT=None, group=None, exchange=None, rank=None, world=None, deal="block",
                                    gather="all", **kwargs):
    """BASELINE cfg 5 on N GPUs: windows are independent, so every rank (one process per GPU) runs its share of the window plan -- a
    contiguous run of the plan (`deal='block'`, windows.assign_window_blocks) or every N-th window, heaviest first (`'round_robin'`) --
    and the per-window match tables are exchanged once over a host channel.  gather='all': every rank returns the frame a single
    process would return (windows in plan order, rows in window order); 'root': rank 0 does, the others their own part; 'none':
    every rank its own part (a run of the plan under the block deal) -- what `sharded_merged_window_matches` takes, which never moves
    the whole table anywhere.

    The host channel is `group.allgather_object` of a `HostGroup` (built from RANK / WORLD_SIZE when none is given); any
    launcher can supply its own `exchange(obj) -> [obj of rank 0, ..., obj of rank world-1]` together with `rank` and
    `world` (MPI, files, a queue).  `outprefix`, if given, gets a per-rank subdirectory (`rank{r}`) so ranks never write
    the same CSV."""
    from .window_api import sliding_window_matching

    return _sharded_windows(sliding_window_matching, ref, moving, commonCT, group, exchange, rank, world, deal, gather, kwargs)


def _sharded_windows(run, ref, moving, commonCT, group, exchange, rank, world, deal, gather, kwargs):
    import os

    import pandas as pd

    if gather not in ("all", "root", "none"):
        raise ValueError(f"gather must be 'all', 'root' or 'none', got {gather!r}")
    own_group = None
    if exchange is None:
        if group is None:
            group = own_group = HostGroup()
        world, rank, exchange = group.world, group.rank, group.allgather_object
        _learn_cpu_sharing(group)
    elif rank is None or world is None:
        raise ValueError("a custom exchange needs rank and world")
    try:
        if kwargs.get("outprefix"):
            kwargs["outprefix"] = os.path.join(kwargs["outprefix"], f"rank{rank}")
        part = run(ref, moving, commonCT=commonCT, _shard=(int(rank), int(world), deal), **kwargs)
        if gather == "none":
            return part
        parts = [p for p in exchange(part) if p is not None and len(p)]
    finally:
        if own_group is not None:
            own_group.close()
    if gather == "root" and rank != 0:
        return part
    if not parts:
        return pd.DataFrame()
    return _plan_order(parts)


def _learn_cpu_sharing(group):
    """the Qhull helper budget of a rank depends on whether the ranks of this host share CPUs: settled from their masks, once per group"""
    if getattr(group, "_cpu_sharing_known", False) or not hasattr(group, "allgather_object"):
        return
    from . import qhull_pool

    qhull_pool.learn_cpu_sharing(group)
    try:
        group._cpu_sharing_known = True
    except AttributeError:
        pass


def _plan_order(parts):
    """The ranks' tables -> one table in plan order.  Each part is a sequence of whole windows with ascending `__plan_pos`: the windows'
    blocks are put in order (a handful of slices under the block deal, where the parts are runs of the plan already) -- no sort of rows."""
    import pandas as pd

    whole = pd.concat(parts, ignore_index=True)
    pos = whole["__plan_pos"].to_numpy()
    whole = whole.drop(columns=["__plan_pos"])
    if len(pos) and np.all(pos[1:] >= pos[:-1]):
        return whole
    cut = np.flatnonzero(np.diff(pos)) + 1
    begins, ends = np.concatenate(([0], cut)), np.concatenate((cut, [len(pos)]))
    order = np.argsort(pos[begins], kind="stable")
    rows = np.concatenate([np.arange(begins[b], ends[b]) for b in order.tolist()])
    from .merge import _take_rows

    return _take_rows(whole, rows)


def sharded_merged_window_incumbent(ref, moving, commonCT=None, group=None, ctx=None, comm=None, deal="block", gather=None, **kwargs):
    """The window loop AND the window merge (src/same.py:297-595 with the greedy incumbent per window, then src/helpers.py:692-815) on N
    GPUs without ever bringing the ranks' tables together: every rank runs its share of the plan, de-duplicates and matches what only it
    can see, and one small all-gather of the seam rows (over `comm` -- RCCL -- on `ctx`; over the host group without them) settles the
    rest.  -> this rank's part of the merged table (aligned ids ascending); gather='all' / 'root' additionally assembles the single
    process's table on every rank / on rank 0 (`merge.join_merged_parts`)."""
    from .incumbent import sliding_window_incumbent
    from .merge import join_merged_parts

    own_group = None
    if group is None:
        group = own_group = HostGroup()
    try:
        _learn_cpu_sharing(group)
        channel = MergeChannel(group, ctx, comm)
        shard = (group.rank, group.world, deal) if group.world > 1 else None
        part = sliding_window_incumbent(ref, moving, commonCT=commonCT, ctx=ctx, merge=True, _shard=shard, _merge_channel=channel, **kwargs)
        if gather in ("all", "root") and group.world > 1:
            tab, extra = (part[0], part[1:]) if isinstance(part, tuple) else (part, ())
            parts = group.allgather_object(tab)
            if gather == "all" or group.rank == 0:
                op = kwargs.get("optim_params") or {}
                tab = join_merged_parts(parts, op.get("cell_id_col") or "Cell_Num_Old")
            part = (tab,) + tuple(extra) if extra else tab
        return part
    finally:
        if own_group is not None:
            own_group.close()


def sharded_merged_window_matches(ref, moving, commonCT=None, group=None, ctx=None, comm=None, deal="block", gather=None, outprefix=None,
                                  moving_delaunay=None, moving_delaunay_vertex_col=None, optim_params=None, gurobi_params=None,
                                  ignore_precomputed_triangulation=False, **kwargs):
    """The SOLVER loop on N GPUs followed by the window merge: `sliding_window_matching` (src/same.py:297-595) on every rank's run of the
    plan, then `merge_window_matches_unique_ref` (src/helpers.py:692-815) dealt the same way -- every rank merges its own table, one
    all-gather of the seam rows (as id codes: numeric whatever the ids are) settles the rest.  -> this rank's part of the merged table
    (aligned ids ascending); gather='all' / 'root' joins the parts.  kwargs: sliding_window_matching's private switches (`_pipeline`,
    `_solve`, `_run_window`)."""
    import os

    from .merge import join_merged_parts, merge_table_part
    from .window_api import _WindowJob, codes_of_ids, frame_id_codes, sliding_window_matching

    own_group = None
    if group is None:
        group = own_group = HostGroup()
    try:
        _learn_cpu_sharing(group)
        if outprefix:
            outprefix = os.path.join(outprefix, f"rank{group.rank}")
        job = _WindowJob(ref, moving, commonCT, outprefix, moving_delaunay, moving_delaunay_vertex_col, optim_params, gurobi_params,
                         ignore_precomputed_triangulation, (group.rank, group.world, deal))
        part = sliding_window_matching(ref, moving, _job=job, **kwargs)
        cid = job.optim_params["cell_id_col"]
        _codes, unique, (mov_ids, ref_ids) = frame_id_codes(job.moving, job.ref, cid)
        merged = merge_table_part(part, job.plan, job.owner, MergeChannel(group, ctx, comm), cid,
                                  reach=abs(float(job.optim_params["radius"])),
                                  ids_unique=unique, id_codes=lambda a, r: (codes_of_ids(mov_ids, a), codes_of_ids(ref_ids, r)))
        if gather in ("all", "root") and group.world > 1:
            parts = group.allgather_object(merged)
            if gather == "all" or group.rank == 0:
                merged = join_merged_parts(parts, cid)
        return merged
    finally:
        if own_group is not None:
            own_group.close()
