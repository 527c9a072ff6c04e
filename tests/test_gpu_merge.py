"""The window merge on the device (csrc/window_merge.hip: accumulator, collect / resolve / finish) against the host statement of the
same merge (same_amd/merge.py with the oracle's de-duplication), which tests/test_merge_parts_cpu.py and the golden fixtures hold to
the reference's merge_window_matches_unique_ref (src/helpers.py:692-815)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _device_merge(acc, a, r, viol, wid, order_pos, order_seq, n_codes_a, n_codes_r):
    """rows from the host -> the merged rows' (aligned code, ref code, window id, violation flag), through the device path"""
    from same_amd import merge as M
    from same_amd.windows import resolve_accumulators

    acc.load(a, r, viol.astype(np.uint8), wid, order_pos, order_seq, n_codes_a, n_codes_r)
    counts, rest = resolve_accumulators([acc], None, None)
    assert counts[0] == len(a) and counts[1] <= counts[0] and counts[2] == len(rest) and counts[3] == counts[1] - counts[2]
    won = M._resolve_rows(rest["ac"].astype(np.int64), rest["rc"].astype(np.int64), (rest["flags"] & 1) != 0, rest["wid"].astype(np.int64),
                          M.already_deduplicated) if len(rest) else np.zeros(0, np.int64)
    final = acc.finish(rest["row"][won])
    return final, counts


def test_accumulator_merge_of_host_rows_equals_the_host_merge(oracle):
    """Random tables from conflict-free to all-conflict, with duplicate pairs whose order decides (equal flag and window id): load ->
    resolve -> the host's matching of the contested cells -> finish gives, row for row, what merge.py's own statement gives with the
    oracle's de-duplication; the counts add up; rows that stand alone never reach the host."""
    from same_amd import merge as M
    from same_amd.windows import MergeAccumulator

    rng = np.random.default_rng(5)
    acc = MergeAccumulator()
    seen_rest, seen_alone, rows_total = 0, 0, 0
    for case in range(60):
        n = int(rng.choice([0, 1, 2, 7, 300, 5000, 40000, 250000]))
        ids = int(rng.choice([1, 3, max(2, n // 3), max(1, n), 4 * max(1, n)]))
        a, r = rng.integers(0, ids, n), rng.integers(0, ids, n)
        if case % 3 == 1 and n > 1:                       # a tiled run: one-to-one but for a few cells of the overlaps
            a, r = rng.permutation(max(n, ids))[:n], rng.permutation(max(n, ids))[:n]
            q = rng.integers(0, n, max(1, n // 20))
            a[q] = a[rng.integers(0, n, len(q))]
            q = rng.integers(0, n, max(1, n // 20))
            r[q] = r[rng.integers(0, n, len(q))]
        if case % 4 == 2 and n > 4:                       # whole rows repeated: the de-duplication's third key (the earlier row) decides
            q = rng.integers(0, n, n // 4)
            a[q], r[q] = a[(q + 1) % n], r[(q + 1) % n]
        viol, wid = rng.random(n) < 0.3, rng.integers(0, 4, n)
        pos, seq = np.zeros(n, np.int32), np.arange(n, dtype=np.int32)
        n_a, n_r = int(a.max()) + 1 if n else 0, int(r.max()) + 1 if n else 0
        final, counts = _device_merge(acc, a, r, viol, wid, pos, seq, n_a, n_r)
        want = M._resolve_rows(a, r, viol, wid, oracle.merge_dedup) if n else np.zeros(0, np.int64)
        assert np.array_equal(final["cidx"], want), (case, n, ids)                    # cidx carries the loaded row's number
        assert np.array_equal(final["a_row"], a[want]) and np.array_equal(final["r_row"], r[want]) and np.array_equal(final["wid"], wid[want])
        assert np.array_equal((final["flags"] & 1) != 0, viol[want])
        assert len(np.unique(final["a_row"])) == len(final) == len(np.unique(final["r_row"]))
        if n:
            assert np.all(np.diff(final["a_row"]) > 0)                                 # aligned codes ascending (src/helpers.py:799-808)
        seen_rest += counts[2]
        seen_alone += counts[3]
        rows_total += n
    assert rows_total > 500_000 and seen_rest > 100_000 and seen_alone > 100_000
    acc.close()


def test_device_seam_flags_equal_the_host_rule():
    """The rows a rank's accumulator leaves to the common step are exactly the rows merge.seam_rows marks (the same two box tests per
    foreign window, on the sections' coordinates), for both deals -- read off the REST list of a pass over one rank's share."""
    import same_amd
    from same_amd import merge as M
    from same_amd import synth
    from same_amd.incumbent import _begin_accumulators
    from same_amd.window_api import _WindowJob
    from same_amd.windows import resolve_accumulators

    cells = synth.make_cells(40_000, 4, seed=3)
    r_df = synth.to_frame(cells)
    m_df = synth.to_frame(synth.make_jittered(cells, seed=4))
    cols = synth.type_columns(4)
    op = dict(radius=25, knn=6, window_size=500, overlap=100, min_cells_per_window=20, hip_cost_dtype="float32")

    class Channel:
        def __init__(self, rank, world):
            self.rank, self.world = rank, world

    for world, deal in ((3, "block"), (4, "round_robin")):
        for rank in range(world):
            job = _WindowJob(r_df, m_df, cols, None, None, None, op, None, False, (rank, world, deal))
            frames, own = job.device_frames("device")
            try:
                contexts = frames.worker_contexts(1)
                accs = _begin_accumulators(job, frames, contexts, [0, len(job.todo)], Channel(rank, world))
                pos_of = {id(w): pos for pos, w in job.todo}
                collector = lambda states, windows: accs[0].collect(states, [w["trim"] for w in windows], [w["window_id"] for w in windows],
                                                                    [pos_of[id(w)] for w in windows])
                for dw in frames.windows([w for _p, w in job.todo], collector=collector):
                    assert dw.error is None
                counts, rest = resolve_accumulators(accs, frames.dmov, frames.dref)
                device_seam = rest[(rest["flags"] & 4) != 0]
                # the same pass's rows on the host: the pre-merge table of this rank, seam rule applied to every row
                table = same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op), _shard=(rank, world, deal))
                x, y, u, v = (table[c].to_numpy() for c in ("X", "Y", "ref_X", "ref_Y"))
                host = M.seam_rows(table["__plan_pos"].to_numpy(), lambda b, e: (x[b:e], y[b:e], u[b:e], v[b:e]), job.plan, job.owner, rank, 25.0)
                assert counts[0] == len(table) == counts[1] and 0 < host.sum() < len(table)          # (no pair twice in this job)
                want = set(zip(table["Aligned_Cell_Num_Old"].to_numpy()[host].tolist(), table["Ref_Cell_Num_Old"].to_numpy()[host].tolist(),
                               table["window_id"].to_numpy()[host].tolist()))
                got = set(zip(device_seam["ac"].tolist(), device_seam["rc"].tolist(), device_seam["wid"].tolist()))
                assert got == want, (world, deal, rank, len(got), len(want))
                accs[0].finish(np.zeros(0, np.int32))
            finally:
                if own:
                    frames.close()


def test_accumulator_grows_and_is_reused(oracle):
    """An accumulator begun for fewer rows than a pass brings grows (rows so far move along); begun again it starts empty."""
    from same_amd.windows import MergeAccumulator, resolve_accumulators

    acc = MergeAccumulator()
    rng = np.random.default_rng(2)
    for n in (50_000, 10, 120_000, 0, 3):
        a, r = rng.permutation(max(n, 1) * 2)[:n], rng.permutation(max(n, 1) * 2)[:n]
        final, counts = _device_merge(acc, a, r, np.zeros(n, bool), np.zeros(n, np.int64), np.zeros(n, np.int32), np.arange(n, dtype=np.int32),
                                      2 * max(n, 1), 2 * max(n, 1))
        assert counts == (n, n, 0, n) and np.array_equal(np.sort(a), final["a_row"])
    with pytest.raises(Exception, match="aligned codes"):
        acc.load(np.array([5]), np.array([0]), np.zeros(1, np.uint8), np.zeros(1), np.zeros(1), np.zeros(1), 3, 3)
    acc.close()


def test_table_columns_from_the_device_equal_the_host_gather(monkeypatch):
    """merge=True on the device route: the merged table's float64 columns (types, X, Y, ref_X, ref_Y) are written by the device into
    page-locked host memory while the host gathers the rest -- the table is the one the host's own gather makes (SAME_TABLE_COLUMNS=host),
    bit for bit, also when the pool's blocks are re-used and when callers keep more tables alive than the pool hands out."""
    import gc

    import same_amd
    from same_amd import synth
    from same_amd.windows import PINNED_BLOCKS

    cells = synth.make_cells(60_000, 5, seed=8)
    r_df = synth.to_frame(cells)
    m_df = synth.to_frame(synth.make_jittered(cells, seed=9))
    cols = synth.type_columns(5)
    op = dict(radius=25, knn=6, window_size=600, overlap=150, min_cells_per_window=20, hip_cost_dtype="float32")
    with same_amd.resident_frames(r_df, m_df) as res:
        call = lambda: same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), merge=True, workers=2)
        monkeypatch.setenv("SAME_TABLE_COLUMNS", "host")
        want = call()
        monkeypatch.delenv("SAME_TABLE_COLUMNS")
        assert len(want) > 40_000 and PINNED_BLOCKS.out == 0
        kept = []
        for rep in range(PINNED_BLOCKS.LIMIT + 2):          # the first LIMIT tables hold a block each; then the host gathers
            got = call()
            assert list(got.columns) == list(want.columns) and got.equals(want), rep
            assert all(got[c].dtype == want[c].dtype for c in want.columns)
            kept.append(got)
        assert PINNED_BLOCKS.out == PINNED_BLOCKS.LIMIT
        first = kept[0]["X"].to_numpy().copy()
        del kept[1:], got
        gc.collect()
        assert PINNED_BLOCKS.out == 1 and len(PINNED_BLOCKS.free) == PINNED_BLOCKS.KEEP
        again = call()                                        # a re-used block: the table kept alive is not written over
        assert again.equals(want) and np.array_equal(kept[0]["X"].to_numpy(), first)
        assert same_amd.merge_window_matches_unique_ref([same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op))]).equals(want)
    del kept, again
    gc.collect()
    assert PINNED_BLOCKS.out == 0


def test_merge_channel_over_a_real_rccl_communicator(tmp_path):
    """dist.MergeChannel with an RCCL communicator (size 1 on this GPU: what a rank of N does, with itself as the only peer): the seam
    rows' table goes through allgather_table's device path -- size exchange over the host group, upload, ncclAllGather on the context's
    stream, download -- and comes back as it was; an empty table too.  Then a whole merged pass with that channel in place."""
    import same_amd
    from same_amd import _lib, synth
    from same_amd.dist import MergeChannel, RcclGroup
    from same_amd.rendezvous import HostGroup

    ctx = _lib.Context(0)
    with HostGroup(0, 1, rdv_dir=str(tmp_path / "rdv"), timeout=60) as group:
        comm = RcclGroup(ctx, 1, 0, lambda b: b)
        try:
            channel = MergeChannel(group, ctx, comm)
            rng = np.random.default_rng(1)
            for n in (0, 1, 3000):
                sent = {"a": rng.integers(0, 10 ** 6, n), "r": rng.integers(0, 10 ** 6, n), "viol": (rng.random(n) < 0.5).astype(np.uint8),
                        "window": rng.integers(0, 99, n), "order": rng.integers(0, 2 ** 40, n), "row": np.arange(n), "rank": np.zeros(n, np.int32)}
                (back,) = channel.tables(sent)
                assert list(back) == list(sent) and all(np.array_equal(back[k], sent[k]) and back[k].dtype == sent[k].dtype for k in sent)
            assert channel.sent_rows == 3001 and channel.gather_ms > 0.0
            cells = synth.make_cells(30_000, 4, seed=5)
            r_df, m_df = synth.to_frame(cells), synth.to_frame(synth.make_jittered(cells, seed=6))
            cols = synth.type_columns(4)
            op = dict(radius=25, knn=6, window_size=600, overlap=150, min_cells_per_window=20, hip_cost_dtype="float32")
            want = same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op), merge=True, ctx=ctx)
            # world 1 behind a channel: no seams, nothing to exchange, the same table
            got = same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op), merge=True, ctx=ctx, _merge_channel=channel)
            assert len(want) > 20_000 and got.equals(want)
        finally:
            comm.close()
    ctx.close()


def test_ranks_without_windows_still_take_part():
    """More ranks than windows: the ranks the deal leaves empty-handed make the same exchange as the others (with nothing to send) and
    return an empty part; the parts together are the single process's merged table."""
    import threading

    import same_amd
    from same_amd import _lib, synth
    from same_amd.merge import join_merged_parts
    from same_amd.windows import window_plan

    cells = synth.make_cells(4000, 3, seed=21)
    r_df = synth.to_frame(cells)
    m_df = synth.to_frame(synth.make_jittered(cells, seed=22))
    cols = synth.type_columns(3)
    op = dict(radius=25, knn=5, window_size=400, overlap=20, min_cells_per_window=20)
    plan = window_plan(cells["xy"], m_df[["X", "Y"]].to_numpy(), 400, 20, 20)
    world = len(plan) + 2
    assert 2 <= len(plan) <= 5
    want = same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op), merge=True)

    class Hub:
        slots, barrier = [None] * world, threading.Barrier(world)

    class Channel:
        def __init__(self, rank):
            self.rank, self.world, self.sent_rows, self.gather_ms = rank, world, 0, 0.0

        def tables(self, table):
            Hub.slots[self.rank] = table
            Hub.barrier.wait(120)
            out = list(Hub.slots)
            Hub.barrier.wait(120)
            return out

    for deal in ("block", "round_robin"):
        parts, errors = [None] * world, []

        def body(rank):
            ctx = _lib.Context(_lib.default_context().device)
            try:
                parts[rank] = same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op), merge=True, ctx=ctx, workers=1,
                                                                _shard=(rank, world, deal), _merge_channel=Channel(rank))
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                Hub.barrier.abort()
            finally:
                ctx.close()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        [t.start() for t in threads]
        [t.join(300) for t in threads]
        assert not errors, (deal, errors[:1])
        assert sum(len(p) == 0 for p in parts) >= 2 and join_merged_parts(parts).equals(want), deal
