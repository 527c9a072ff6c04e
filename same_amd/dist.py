"""Multi-GPU form of the cost build + prune: aligned-row blocks per rank, one all-gather.

SURVEY 8e: every kernel on the path is a map over aligned rows with read-only shared inputs,
so ranks own contiguous row blocks (refs replicated, no merge step) and the only exchange is
an all-gather of the fixed-width pruned candidate lists -- int32 idx[rows][k] (-1 padded) and
float64 cost[rows][k] (+inf padded).  One process per GPU; on GPUs the gather is RCCL over xGMI
on the context's stream (csrc/comm.hip); `HostGather` does the same exchange through a
torch.distributed process group on host arrays (used by the CPU tests and as a fallback
transport -- never as a compute fallback).  Results are identical for any world size:
blocks are concatenated in rank order, then compacted exactly like the single-GPU path.
"""
import numpy as np

from . import _lib
from .knn import compact_pairs, pairs_from_padded


def row_block(n_rows, world, rank):
    """Equal-width blocks (the last ones may be short or empty): -> (begin, end, block_rows)."""
    block = -(-n_rows // world) if world > 0 else n_rows
    begin = min(rank * block, n_rows)
    return begin, min(begin + block, n_rows), block


class HostGather:
    """all-gather of equal-sized host arrays through a torch.distributed group (gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def allgather(self, arr):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        return np.concatenate([o.numpy() for o in outs], axis=0)


class RcclGather:
    """RCCL communicator bound to a Context.  `exchange_id(bytes_or_None) -> bytes` is any host
    broadcast from rank 0 (bench.py uses torch.distributed's store)."""

    def __init__(self, ctx, world, rank, exchange_id):
        self.ctx, self.world, self.rank = ctx, world, rank
        uid = None
        if rank == 0:
            import ctypes

            buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
            ctx.check(ctx.lib.same_comm_unique_id(buf), "same_comm_unique_id")
            uid = buf.raw
        uid = exchange_id(uid)
        assert len(uid) == _lib.UNIQUE_ID_BYTES
        ctx.check(ctx.lib.same_comm_init(ctx.handle, world, rank, uid), "same_comm_init")

    def allgather_dev(self, send_buf, recv_buf, send_bytes):
        c = self.ctx
        c.check(c.lib.same_allgather_dev(c.handle, send_buf.ptr, recv_buf.ptr, send_bytes), "same_allgather_dev")

    def allgather_dev_async(self, send_buf, recv_buf, send_bytes):
        """Gather on the context's communication stream, overlapping whatever compute is queued next."""
        c = self.ctx
        c.check(c.lib.same_allgather_dev_async(c.handle, send_buf.ptr, recv_buf.ptr, send_bytes), "same_allgather_dev_async")

    def wait(self):
        """Order the compute stream after every gather issued so far (stream-side; the host does not block)."""
        self.ctx.check(self.ctx.lib.same_comm_wait(self.ctx.handle), "same_comm_wait")

    def close(self):
        self.ctx.lib.same_comm_destroy(self.ctx.handle)


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


def sharded_knn_cost_host(compute_block, n_aligned, knn, gather):
    """Host-array form: compute_block(begin, end) -> (idx (rows,k) int32, cost (rows,k) f64);
    returns the gathered, unpadded (idx, cost) for all n_aligned rows on every rank."""
    begin, end, block = row_block(n_aligned, gather.world, gather.rank)
    idx_p = np.full((block, knn), -1, np.int32)
    cost_p = np.full((block, knn), np.inf, np.float64)
    if end > begin:
        idx, cost = compute_block(begin, end)
        idx_p[: end - begin], cost_p[: end - begin] = idx, cost
    return gather.allgather(idx_p)[:n_aligned], gather.allgather(cost_p)[:n_aligned]


def sharded_knn_cost_device(ctx, compute, n_aligned, knn, gather):
    """Device form: prune + cost for this rank's block, RCCL all-gather, one D2H of the result."""
    begin, end, block = row_block(n_aligned, gather.world, gather.rank)
    didx, dcost, _ = compute(begin, end, block)
    gidx, gcost = ctx.alloc(block * knn * 4 * gather.world), ctx.alloc(block * knn * 8 * gather.world)
    gather.allgather_dev(didx, gidx, block * knn * 4)
    gather.allgather_dev(dcost, gcost, block * knn * 8)
    idx = gidx.download((block * gather.world, knn), np.int32)[:n_aligned]
    cost = gcost.download((block * gather.world, knn), np.float64)[:n_aligned]
    return idx, cost


def pairs_and_costs(aligned_df, ref_df, idx, cost):
    """Gathered padded lists -> compacted frames, valid_pairs and c, exactly as the single-GPU
    path produces them (src/utils.py:731-742 order)."""
    knn_pairs = pairs_from_padded(idx)
    rows, cols = np.nonzero(idx >= 0)
    c = list(cost[rows, cols])
    new_a, new_r, new_pairs = compact_pairs(aligned_df, ref_df, knn_pairs)
    return new_a, new_r, new_pairs, c


def sharded_sliding_window_matching(ref, moving, commonCT=None, group=None, exchange=None, rank=None, world=None, **kwargs):
    """BASELINE cfg 5 on N GPUs: windows are independent, so every rank (one process per GPU) runs its round-robin share of
    the window plan -- heaviest windows first, `windows.assign_windows` -- and the per-window match tables are exchanged once
    over a host channel (they are small frames, not a device collective).  Every rank returns the frame a single process
    would return: windows in plan order, rows in window order.

    The host channel is `torch.distributed.all_gather_object` on `group` by default; any launcher can supply its own
    `exchange(obj) -> [obj of rank 0, ..., obj of rank world-1]` together with `rank` and `world` (MPI, files, a queue).
    `outprefix`, if given, gets a per-rank subdirectory (`rank{r}`) so ranks never write the same CSV."""
    import os

    import pandas as pd

    from .api import sliding_window_matching

    if exchange is None:
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def exchange(obj):
            parts = [None] * world
            dist.all_gather_object(parts, obj, group=group)
            return parts
    elif rank is None or world is None:
        raise ValueError("a custom exchange needs rank and world")
    if kwargs.get("outprefix"):
        kwargs["outprefix"] = os.path.join(kwargs["outprefix"], f"rank{rank}")
    part = sliding_window_matching(ref, moving, commonCT=commonCT, _shard=(int(rank), int(world)), **kwargs)
    parts = [p for p in exchange(part) if p is not None and len(p)]
    if not parts:
        return pd.DataFrame()
    merged = pd.concat(parts, ignore_index=True)
    return merged.sort_values("__plan_pos", kind="stable").drop(columns=["__plan_pos"]).reset_index(drop=True)
