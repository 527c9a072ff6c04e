"""The window merge dealt over ranks (same_amd.merge.merged_part_rows / merge_table_part, dist.MergeChannel): every rank decides what
only it can see, one small exchange settles the seams -- and the parts are the single process's merged table
(src/helpers.py:692-815), row for row, for any rank count and either deal.  No GPU: the de-duplication step is the oracle's."""
import os
import threading

import numpy as np
import pandas as pd
import pytest

RADIUS = 25.0


def _plan(nx, ny, ws=300, overlap=80, wide=()):
    """A window plan the way windows.window_plan lays it out (column by column, central trims tiling the section); windows in `wide` get
    a trim that reaches into their neighbours', as merged windows' trims do."""
    step = ws - overlap
    plan = []
    for i in range(nx):
        for j in range(ny):
            x0, y0 = i * step, j * step
            x1, y1 = x0 + ws, y0 + ws
            trim = [x0 if i == 0 else x0 + overlap / 2, x1 if i == nx - 1 else x1 - overlap / 2,
                    y0 if j == 0 else y0 + overlap / 2, y1 if j == ny - 1 else y1 - overlap / 2]
            if len(plan) in wide:
                trim = [trim[0] - 30, trim[1] + 30, trim[2] - 30, trim[3] + 30]
            plan.append({"window_id": nx * j + i, "box": (x0, x1, y0, y1), "trim": tuple(trim), "n_mov": 100 + 7 * ((i * 5 + j) % 9), "n_ref": 100})
    return plan, (nx - 1) * step + ws, (ny - 1) * step + ws


def _window_tables(plan, width, height, seed, n_cells=6000, string_ids=False):
    """Per-window match tables with the structure the window loop gives them: a window's aligned cells lie in its trim, each matched to
    ONE reference cell within RADIUS, references unique inside a window -- so only windows near each other can disagree."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    ref = np.column_stack((rng.uniform(0, width, n_cells), rng.uniform(0, height, n_cells)))
    mov = ref[rng.permutation(n_cells)[: n_cells * 9 // 10]] + rng.normal(0, 6.0, (n_cells * 9 // 10, 2))
    tree = cKDTree(ref)
    a_id = rng.permutation(10 * len(mov))[: len(mov)] if not string_ids else np.array([f"a{v:05d}" for v in rng.permutation(len(mov))], dtype=object)
    r_id = rng.permutation(10 * len(ref))[: len(ref)] if not string_ids else np.array([f"r{v:05d}" for v in rng.permutation(len(ref))], dtype=object)
    tables = []
    for pos, w in enumerate(plan):
        x0, x1, y0, y1 = w["trim"]
        inside = np.flatnonzero((mov[:, 0] >= x0) & (mov[:, 0] < x1) & (mov[:, 1] >= y0) & (mov[:, 1] < y1))
        inside = inside[rng.random(len(inside)) < 0.9]
        taken, rows = set(), []
        for a in inside.tolist():
            near = [r for r in tree.query_ball_point(mov[a], RADIUS) if r not in taken]
            if near:
                r = near[int(rng.integers(0, min(2, len(near))))]       # one of the two nearest free ones: neighbours disagree often
                taken.add(r)
                rows.append((a, r))
        rows = np.array(rows, dtype=np.int64).reshape(-1, 2)
        a, r = rows[:, 0], rows[:, 1]
        fv = rng.random(len(a)) < 0.3
        tables.append(pd.DataFrame({"aligned_idx": np.arange(len(a)), "X": mov[a, 0], "Y": mov[a, 1], "ref_X": ref[r, 0], "ref_Y": ref[r, 1],
                                    "Ref_Cell_Num_Old": r_id[r], "Aligned_Cell_Num_Old": a_id[a], "filtered_violation": fv,
                                    "payload": rng.random(len(a)), "window_id": w["window_id"], "__plan_pos": pos}))
    return tables


class _Hub:
    def __init__(self, world):
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)


class _ThreadChannel:
    """dist.MergeChannel's interface between threads of one process"""

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world, self.sent_rows = hub, rank, hub.world, 0

    def _all(self, v):
        self.hub.slots[self.rank] = v
        self.hub.barrier.wait()
        out = list(self.hub.slots)
        self.hub.barrier.wait()
        return out

    def tables(self, table):
        self.sent_rows += len(table["row"])
        return self._all(table)

    def max(self, v):
        return max(self._all(float(v)))


def _run_ranks(world, fn):
    out, errors = [None] * world, []
    hub = _Hub(world)

    def body(rank):
        try:
            out[rank] = fn(rank, _ThreadChannel(hub, rank))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            hub.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    if errors:
        raise errors[0]
    return out


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("deal", ["block", "round_robin"])
def test_merged_parts_make_the_single_process_merge(oracle, world, deal):
    from same_amd import merge as M
    from same_amd.windows import deal_windows

    sent_share = []
    for seed, (nx, ny, wide, strings) in enumerate([(6, 5, (), False), (4, 4, (5, 6), False), (5, 3, (7,), True), (8, 8, (), False)]):
        plan, width, height = _plan(nx, ny, wide=wide)
        tables = _window_tables(plan, width, height, seed, string_ids=strings)
        want = M.merge_window_matches_unique_ref([t.drop(columns=["__plan_pos"]) for t in tables], _dedup=oracle.merge_dedup)
        assert len(want) > 1000
        owner = deal_windows(plan, world, deal)

        def rank_part(rank, channel):
            mine = [tables[p] for p in np.flatnonzero(owner == rank)]
            table = pd.concat(mine, ignore_index=True) if mine else None
            part = M.merge_table_part(table, plan, owner, channel, ids_unique=True, _dedup=oracle.merge_dedup)
            return part, channel.sent_rows, 0 if table is None else len(table)

        res = _run_ranks(world, rank_part)
        parts = [r[0] for r in res]
        for p in parts:
            assert "__plan_pos" not in p.columns
            ids = p["Aligned_Cell_Num_Old"].to_numpy()
            assert all(ids[q] < ids[q + 1] for q in range(len(ids) - 1))               # every part is in the merged table's order
        got = M.join_merged_parts(parts)
        assert list(got.columns) == list(want.columns) and got.equals(want), (seed, world, deal)
        sent_share.append(sum(r[1] for r in res) / max(1, sum(r[2] for r in res)))
    if world == 1:
        assert max(sent_share) == 0.0
    elif deal == "block" and world <= 3:
        assert sent_share[3] < 0.35, sent_share       # an 8 x 8 plan in 2-3 strips: the borders' rows travel, the strips' insides do not


def test_seam_rows_are_a_superset_of_what_ranks_share(oracle):
    """A row another rank's table shares a cell with is always marked; a window in the middle of a rank's block has no marked row."""
    from same_amd import merge as M
    from same_amd.windows import deal_windows

    plan, width, height = _plan(8, 6, wide=(24,))
    tables = _window_tables(plan, width, height, 11)
    for world, deal in ((2, "block"), (4, "block"), (3, "round_robin")):
        owner = deal_windows(plan, world, deal)
        whole = pd.concat(tables, ignore_index=True)
        rank_of = owner[whole["__plan_pos"].to_numpy()]
        marked = np.zeros(len(whole), bool)
        for rank in range(world):
            rows = np.flatnonzero(rank_of == rank)
            t = whole.iloc[rows]
            x, y, u, v = (t[c].to_numpy() for c in ("X", "Y", "ref_X", "ref_Y"))
            marked[rows] = M.seam_rows(t["__plan_pos"].to_numpy(), lambda b, e: (x[b:e], y[b:e], u[b:e], v[b:e]), plan, owner, rank, RADIUS)
        for col in ("Aligned_Cell_Num_Old", "Ref_Cell_Num_Old"):
            ranks_per_id = pd.DataFrame({"id": whole[col], "rank": rank_of}).groupby("id")["rank"].nunique()
            shared = whole[col].map(ranks_per_id).to_numpy() > 1
            assert shared.sum() > 0 and marked[shared].all(), (world, deal, col)
        if deal == "block":
            assert 0 < marked.mean() < 0.5
            inner = [p for p in range(len(plan)) if all(owner[q] == owner[p] for q in range(len(plan))
                                                        if abs(plan[q]["box"][0] - plan[p]["box"][0]) <= 220 and abs(plan[q]["box"][2] - plan[p]["box"][2]) <= 220)]
            assert inner and not marked[np.isin(whole["__plan_pos"].to_numpy(), inner)].any()


def test_duplicate_ids_send_every_row_to_the_common_step(oracle):
    """Where a cell id names several rows of a frame nothing can be reasoned from positions: ids_unique=False is still the merge."""
    from same_amd import merge as M
    from same_amd.windows import deal_windows

    plan, width, height = _plan(4, 3)
    tables = _window_tables(plan, width, height, 5)
    for t in tables:
        t["Aligned_Cell_Num_Old"] %= 500            # far-apart cells now share ids
        t["Ref_Cell_Num_Old"] %= 700
    want = M.merge_window_matches_unique_ref([t.drop(columns=["__plan_pos"]) for t in tables], _dedup=oracle.merge_dedup)
    owner = deal_windows(plan, 3, "block")

    def rank_part(rank, channel):
        table = pd.concat([tables[p] for p in np.flatnonzero(owner == rank)], ignore_index=True)
        return M.merge_table_part(table, plan, owner, channel, ids_unique=False, _dedup=oracle.merge_dedup)

    assert M.join_merged_parts(_run_ranks(3, rank_part)).equals(want)


def test_block_deal_is_contiguous_and_balanced():
    from same_amd.windows import assign_window_blocks, deal_windows

    plan, _w, _h = _plan(12, 12)
    for world in (1, 2, 3, 8, 200):
        shards = assign_window_blocks(plan, world)
        assert [w for s in shards for w in s] == list(range(len(plan)))                 # runs of the plan, in rank order
        loads = [sum(plan[w]["n_mov"] for w in s) for s in shards]
        if world <= 8:
            assert max(loads) - min(loads) <= 2 * max(w["n_mov"] for w in plan)
    assert assign_window_blocks([], 3) == [[], [], []]
    assert deal_windows(plan, 8, "round_robin").tolist() != deal_windows(plan, 8, "block").tolist()
    with pytest.raises(ValueError):
        deal_windows(plan, 2, "diagonal")


def _host_group_worker(rank, world, rdv, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from oracle import same_oracle
    from same_amd import merge as M
    from same_amd.dist import MergeChannel
    from same_amd.rendezvous import HostGroup
    from same_amd.windows import deal_windows
    from test_merge_parts_cpu import _plan, _window_tables

    plan, width, height = _plan(8, 8, wide=(20,))
    tables = _window_tables(plan, width, height, 3)
    owner = deal_windows(plan, world, "block")
    with HostGroup(rank, world, rdv_dir=rdv, timeout=120) as g:
        channel = MergeChannel(g)                                         # no context, no communicator: the host group carries the seam rows
        mine = [tables[p] for p in np.flatnonzero(owner == rank)]
        part = M.merge_table_part(pd.concat(mine, ignore_index=True) if mine else None, plan, owner, channel, _dedup=same_oracle.merge_dedup)
        g.barrier()
    part.to_pickle(os.path.join(out_dir, f"part{rank}.pkl"))


def test_merged_parts_over_the_host_group_world_8(tmp_path, oracle):
    """The same through the product's own channel (dist.MergeChannel over a HostGroup: allgather_table's byte blocks), eight processes."""
    import multiprocessing as mp

    from same_amd import merge as M

    world = 8
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_host_group_worker, args=(r, world, str(tmp_path / "rdv"), str(tmp_path))) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    plan, width, height = _plan(8, 8, wide=(20,))
    tables = _window_tables(plan, width, height, 3)
    want = M.merge_window_matches_unique_ref([t.drop(columns=["__plan_pos"]) for t in tables], _dedup=oracle.merge_dedup)
    got = M.join_merged_parts([pd.read_pickle(tmp_path / f"part{r}.pkl") for r in range(world)])
    assert got.equals(want)


def test_rows_connected_to_a_seam_row_travel_with_it(oracle):
    """A chain a1-r1-a2-r2-a3 with ONE seam row goes to the common step whole; a contested pair elsewhere and a lone row are decided here."""
    from same_amd.merge import _resolve_rows

    a = np.array([1, 2, 2, 3, 4, 5, 6, 7])
    r = np.array([1, 1, 2, 2, 9, 5, 5, 8])
    seam = np.array([True, False, False, False, False, False, False, True])
    viol, wid = np.zeros(8, bool), np.arange(8)
    mine, common = _resolve_rows(a, r, viol, wid, oracle.merge_dedup, seam=seam)
    assert sorted(common.tolist()) == [0, 1, 2, 3, 7]                     # the chain + the lone seam row
    assert mine.tolist() == [4, 5]                                         # the lone row, and aligned 5 wins reference 5 (ids in order)
    assert _resolve_rows(a, r, viol, wid, oracle.merge_dedup).tolist() == [0, 2, 4, 5, 7]
    none, every = _resolve_rows(a, r, viol, wid, oracle.merge_dedup, seam=np.ones(8, bool))
    assert len(none) == 0 and sorted(every.tolist()) == list(range(8))
    empty = np.zeros(0, np.int64)
    got = _resolve_rows(empty, empty, np.zeros(0, bool), empty, oracle.merge_dedup, seam=np.zeros(0, bool))
    assert len(got[0]) == 0 and len(got[1]) == 0


def test_matching_of_components_equals_the_whole():
    """What the per-rank merge rests on: scipy's Hopcroft-Karp gives a connected component the same matching whether it is matched
    alone or inside a larger graph (nodes in the same relative order, adjacency lists sorted)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components, maximum_bipartite_matching

    rng = np.random.default_rng(7)
    checked = 0
    for case in range(300):
        n1, n2 = int(rng.integers(2, 60)), int(rng.integers(2, 60))
        m = int(rng.integers(1, 3 * max(n1, n2)))
        g = csr_matrix((np.ones(m), (rng.integers(0, n1, m), rng.integers(0, n2, m))), shape=(n1, n2))
        g.sum_duplicates()
        g.sort_indices()
        whole = maximum_bipartite_matching(g, perm_type="column")
        coo = g.tocoo()
        both = csr_matrix((np.ones(len(coo.row)), (coo.row, n1 + coo.col)), shape=(n1 + n2, n1 + n2))
        n_comp, label = connected_components(both, directed=False)
        take = rng.random(n_comp) < 0.5                                    # any subset of components, matched without the others
        rows, cols = np.flatnonzero(take[label[:n1]]), np.flatnonzero(take[label[n1:]])
        if len(rows) == 0 or len(cols) == 0:
            continue
        sub = g[rows][:, cols]
        sub.sort_indices()
        part = maximum_bipartite_matching(sub, perm_type="column")
        want = whole[rows]
        assert np.array_equal(np.where(part >= 0, cols[np.maximum(part, 0)], -1), want), case
        checked += 1
    assert checked > 200


def test_sharded_window_tables_come_back_in_plan_order_without_a_sort():
    """dist._sharded_windows: the ranks' tables are put in plan order by whole windows (runs of the plan under the block deal: nothing
    to do at all), on every rank (`gather='all'`), on rank 0 only (`'root'`), or not at all (`'none'`)."""
    from same_amd import dist
    from same_amd.windows import deal_windows

    plan, _w, _h = _plan(5, 4)
    rng = np.random.default_rng(0)
    tables = [pd.DataFrame({"v": rng.random(int(rng.integers(0, 7))), "window_id": w["window_id"], "__plan_pos": pos}) for pos, w in enumerate(plan)]
    whole = pd.concat(tables, ignore_index=True).drop(columns=["__plan_pos"])
    for deal in ("block", "round_robin"):
        for world in (1, 2, 3):
            owner = deal_windows(plan, world, deal)

            def run(ref, moving, commonCT=None, _shard=None, **_kw):
                assert _shard[2] == deal
                mine = [tables[p] for p in np.flatnonzero(owner == _shard[0])]
                return pd.concat(mine, ignore_index=True) if mine else pd.DataFrame()

            parts = [run(None, None, _shard=(r, world, deal)) for r in range(world)]
            for rank in range(world):
                for gather in ("all", "root", "none"):
                    got = dist._sharded_windows(run, None, None, None, None, lambda part: parts, rank, world, deal, gather, {})
                    if gather == "none" or (gather == "root" and rank != 0):
                        assert got.equals(parts[rank])
                    else:
                        assert list(got.columns) == ["v", "window_id"] and got.equals(whole), (deal, world, rank, gather)
    with pytest.raises(ValueError, match="gather"):
        dist._sharded_windows(run, None, None, None, None, lambda part: parts, 0, 3, "block", "some", {})
    assert dist._plan_order([tables[3], tables[1], tables[7]])["window_id"].tolist() == \
        pd.concat([tables[1], tables[3], tables[7]])["window_id"].tolist()
