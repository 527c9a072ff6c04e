"""Host logic of the window API's device pipeline that needs no GPU: the frames a window's run_same body reads, which inputs the
sections refuse, the solver-free table builder, the pipeline switch, the timing double."""
import types

import numpy as np
import pandas as pd
import pytest


def _frame(n=400, T=3, seed=0, index=None):
    rng = np.random.default_rng(seed)
    df = pd.DataFrame(rng.random((n, T)) * 100, columns=[f"c{t}" for t in range(T)])
    df.insert(0, "Y", rng.uniform(0, 500, n))
    df.insert(0, "X", rng.uniform(0, 500, n))
    df["cell_type"] = rng.choice(np.array(["a", "b", "c"], dtype=object), n)
    df["Cell_Num_Old"] = np.arange(n) * 3 + 1
    if index is not None:
        df.index = index
    return df


def test_window_frame_is_the_frame_run_same_holds_after_its_prune():
    """window_api._window_frame(df, rows): rows of the caller's frame with the helper columns of src/same.py:934-970 and the renumbering
    of src/utils.py:739-740 -- what the frame pipeline makes with iloc / column assignments / reset_index, for any index labels."""
    from same_amd.window_api import _window_frame

    for index in (None, np.arange(400)[::-1] * 7, np.array([f"id{q}" for q in range(400)], dtype=object)):
        df = _frame(index=index)
        rows = np.array([3, 17, 18, 250, 399])
        for aligned, vertex_col in ((True, None), (True, "Cell_Num_Old"), (False, None)):
            want = df.iloc[rows].copy()
            want["size"] = 1
            want["__orig_idx"] = want.index.to_numpy()
            if aligned:
                want["__tri_vid"] = want.index.to_numpy() if vertex_col is None else want[vertex_col].to_numpy()
            want = want.reset_index(drop=True)
            got = _window_frame(df, rows, vertex_col, aligned=aligned)
            assert list(got.columns) == list(want.columns) and got.equals(want) and isinstance(got.index, pd.RangeIndex)
    sized = _frame()
    sized["size"] = np.arange(400) % 4 + 1.5                      # an existing size column (and __orig_idx) is kept, not overwritten
    sized["__orig_idx"] = -np.arange(400)
    got = _window_frame(sized, np.array([5, 6]), None, aligned=True)
    assert got["size"].tolist() == [2.5, 3.5] and got["__orig_idx"].tolist() == [-5, -6] and got["__tri_vid"].tolist() == [5, 6]
    with pytest.raises(ValueError, match="aligned_delaunay_vertex_col='nope' not in aligned_df"):
        _window_frame(df, rows, "nope", aligned=True)
    frame0 = _frame()
    _window_frame(frame0, np.array([1, 2]), None, aligned=True)
    assert "__orig_idx" not in frame0.columns and "size" not in frame0.columns          # the caller's frame is never touched


def test_sections_refuse_what_the_frame_pipeline_must_raise_for():
    from same_amd.params import init_optim_params
    from same_amd.window_api import _DeviceFrames, window_pipeline

    ref, mov, cts = _frame(seed=1), _frame(seed=2), ["c0", "c1", "c2"]
    op = init_optim_params()
    assert _DeviceFrames.refusal(ref, mov, cts, op) is None
    assert _DeviceFrames.refusal(ref, mov, cts, init_optim_params(hip_cost_dtype="float32")) is None
    assert _DeviceFrames.refusal(ref, mov, cts, dict(op, hip_cost_dtype="float16")) == "hip_cost_dtype"
    assert "missing" in _DeviceFrames.refusal(ref, mov, cts + ["c9"], op)
    assert "not numeric" in _DeviceFrames.refusal(ref.assign(c1=ref["c1"].astype(str)), mov, cts, op)
    assert "size is not numeric" in _DeviceFrames.refusal(ref, mov.assign(size="big"), cts, op)
    assert "cell_type missing" in _DeviceFrames.refusal(ref, mov.drop(columns=["cell_type"]), cts, op)
    assert _DeviceFrames.refusal(ref, mov.drop(columns=["cell_type"]), cts, dict(op, ignore_same_type_triangles=False)) is None
    assert "priority filter" in _DeviceFrames.refusal(ref.drop(columns=["cell_type"]), mov, cts, dict(op, ignore_knn_if_matched=True))
    assert "vertex_col" in _DeviceFrames.refusal(ref, mov, cts, op, vertex_col="mc_id") and _DeviceFrames.refusal(ref, mov, cts, op, "Cell_Num_Old") is None
    assert window_pipeline() == "device" and window_pipeline("FRAMES") == "frames"
    with pytest.raises(ValueError):
        window_pipeline("columns")


def test_pipeline_switch_reads_the_environment(monkeypatch):
    from same_amd.window_api import window_pipeline

    monkeypatch.setenv("SAME_WINDOW_PIPELINE", "frames")
    assert window_pipeline() == "frames" and window_pipeline("device") == "device"


def _fake_window(rng, n_rows, n_ref, pos, trim):
    rows_m = np.sort(rng.choice(n_rows, 120, replace=False)).astype(np.int32)
    match = np.where(rng.random(120) < 0.8, rng.integers(0, n_ref, 120), -1).astype(np.int32)
    return types.SimpleNamespace(rows_m=rows_m, match_row=match, point_flag=(rng.random(120) < 0.3).astype(np.uint8),
                                 flip_flag=(rng.random(120) < 0.1).astype(np.uint8)), dict(trim=trim, window_id=100 + pos)


@pytest.mark.parametrize("float_columns", [True, False, "strided columns, string ids"])
def test_table_builder_makes_the_reference_table(float_columns, monkeypatch):
    """incumbent._TableBuilder (the device route's result table, gathered when the pass is over) against the same table made the way run_same's
    post-solve makes it -- `.map` of the source columns by matched row (src/same.py:1264-1278), central trim (:565-582), window id --
    with float64 columns (row-major block gathers) and with integer type / coordinate columns (per-column gathers, dtypes kept)."""
    from same_amd.incumbent import _TableBuilder
    from same_amd.windows import Section

    ref, mov, cts = _frame(300, seed=3), _frame(400, seed=4), ["c0", "c1", "c2"]
    if not float_columns:
        for df in (ref, mov):
            df["c1"] = (df["c1"] * 10).astype(np.int64)
            df["X"] = df["X"].round().astype(np.int64)
    if isinstance(float_columns, str):
        # frames made from ONE 2-D array: every numeric column is a strided view of the block; ids that are strings (object columns)
        def restride(df):
            num = [c for c in df.columns if df[c].dtype.kind in "fi" and c != "Cell_Num_Old"]
            out = pd.DataFrame(np.ascontiguousarray(df[num].to_numpy(dtype=np.float64)), columns=num)
            out["cell_type"] = df["cell_type"].to_numpy()
            out["Cell_Num_Old"] = np.array([f"cell-{v}" for v in df["Cell_Num_Old"]], dtype=object)
            assert not out["c1"].to_numpy().flags.c_contiguous
            return out
        ref, mov = restride(ref), restride(mov)
    mov["size"] = np.arange(400) % 3 + 1
    job = types.SimpleNamespace(ref=ref, moving=mov, commonCT=cts, optim_params={"cell_id_col": "Cell_Num_Old"}, mine=None)
    sections = (Section.from_frame(ref, cts), Section.from_frame(mov, cts))
    rng = np.random.default_rng(9)
    b = _TableBuilder(job, sections, with_ref_idx=False)
    want = []
    for pos in range(11):
        dw, w = _fake_window(rng, 400, 300, pos, (50.0, 450.0, 100.0, 500.0))
        dw.axy = mov[["X", "Y"]].to_numpy(dtype=np.float64)[dw.rows_m]
        b.add(pos, w, dw)
        ai = np.flatnonzero(dw.match_row >= 0)
        t = pd.DataFrame({"aligned_idx": ai})
        a_df, r_df = mov.iloc[dw.rows_m].reset_index(drop=True), ref
        for ct in cts + ["X", "Y"]:
            t[ct] = t["aligned_idx"].map(a_df[ct])
        rj = dw.match_row[ai]
        for ct in ("X", "Y"):
            t[f"ref_{ct}"] = r_df[ct].to_numpy()[rj]
        t["size"] = t["aligned_idx"].map(a_df["size"])
        t["ref_size"] = 1
        t["Ref_Cell_Num_Old"] = r_df["Cell_Num_Old"].to_numpy()[rj]
        t["Aligned_Cell_Num_Old"] = t["aligned_idx"].map(a_df["Cell_Num_Old"])
        t["time_limit_reached"] = False
        t["triangle_violation"] = dw.flip_flag[ai].astype(bool)
        t["filtered_violation"] = dw.point_flag[ai].astype(bool)
        t["run_time"] = 0.0
        central = t[(t["X"] >= 50.0) & (t["X"] < 450.0) & (t["Y"] >= 100.0) & (t["Y"] < 500.0)].copy()
        central["window_id"] = w["window_id"]
        want.append(central)
    got, want = _TableBuilder.table([b]), pd.concat(want, ignore_index=True)
    assert list(got.columns) == list(want.columns) and len(got) == len(want) > 300
    for c in got.columns:
        assert got[c].dtype == want[c].dtype and np.array_equal(got[c].to_numpy(), want[c].to_numpy()), c
    monkeypatch.setattr(_TableBuilder, "SLICE", 37)              # many slices: the fill runs on the gather threads
    sliced = _TableBuilder.table([b])
    assert list(sliced.columns) == list(got.columns) and all(sliced[c].dtype == got[c].dtype for c in got.columns) and sliced.equals(got)


def test_timing_double_installs_and_restores_gurobipy():
    """same_amd.bench_solver_double (bench.py's `api_path`, part b): a do-nothing stand-in for the gurobipy surface run_same touches --
    takes the MIP start as the incumbent, calls the lazy callback once, reports OPTIMAL; install / uninstall leave sys.modules as found."""
    import sys

    from same_amd import bench_solver_double as dbl

    before = sys.modules.get("gurobipy")
    gp = dbl.install()
    try:
        assert sys.modules["gurobipy"] is gp
        m = gp.Model("m", env=gp.Env(params={}))
        x = m.addVars(5, vtype=gp.GRB.BINARY, name="x")
        m.addConstr(gp.quicksum(x[i] for i in range(5)) <= 2, name="c")
        m.setObjective(gp.quicksum(3.0 * x[i] for i in range(5)) + 2 * x[0] - x[1], gp.GRB.MINIMIZE)
        x[2].Start, x[4].Start = 1.0, 0.0
        seen = []
        m.optimize(lambda model, where: seen.append((where, model.cbGetSolution([x[i] for i in range(5)]))))
        assert m.status == gp.GRB.OPTIMAL and [v.x for v in x.values()] == [0.0, 0.0, 1.0, 0.0, 0.0] and m.n_constrs == 1
        assert seen == [(gp.GRB.Callback.MIPSOL, [0.0, 0.0, 1.0, 0.0, 0.0])]
    finally:
        dbl.uninstall()
    assert sys.modules.get("gurobipy") is before
