"""run_same / sliding_window_matching end to end on the GPU with a mock solver (tests/fake_gurobipy.py):
the model the solver would receive, the lazy cuts the callback adds, and the post-solve tables."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import frames_from_golden, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gp(monkeypatch, tmp_path):
    import sys
    import fake_gurobipy

    monkeypatch.chdir(tmp_path)
    saved = sys.modules.get("gurobipy")
    mod = fake_gurobipy.install()
    yield mod
    if saved is None:
        sys.modules.pop("gurobipy", None)
    else:
        sys.modules["gurobipy"] = saved


@pytest.mark.parametrize("case", ["synthetic_example", "cfg1_500"])
def test_run_same_with_mock_solver(gp, case, tmp_path):
    import same_amd

    g = load_golden(case)
    a_df, r_df, cols = frames_from_golden(g)
    mad = None if g["params"][2] < 0 else g["params"][2]
    op = dict(radius=g["params"][0], knn=int(g["params"][1]), min_angle_deg=mad, ignore_same_type_triangles=False,
              dist_ct_coeff=g["params"][3], no_match_penalty=g["params"][4])
    gpar = dict(init_method="greedy", lazy_allowed_flip_fraction=0.0, lazy_max_cuts_per_incumbent=50)
    out_df, var_out = same_amd.run_same(r_df, a_df, cols, outprefix=str(tmp_path / "out"), optim_params=op, gurobi_params=gpar)
    model = gp.Model.last
    P, n_a, n_r, n_t = len(g["pairs"]), len(g["kept_aligned"]), len(g["kept_ref"]), len(g["tri_plain"])
    # what the solver consumed: variable counts/order and the assignment constraints of src/helpers.py:102-161
    assert len(model.vars) == P + n_r + n_a + n_t
    names = [n for n, _ in model.constrs]
    n_ref_used, n_al_used = len(np.unique(g["pairs"][:, 1])), len(np.unique(g["pairs"][:, 0]))
    assert len(names) == 2 * n_ref_used + 2 * n_al_used
    assert names[0].startswith("max_matches_") and names[n_ref_used].startswith("one_match_") and names[-1].startswith("no_match_")
    # objective coefficients on x are the pair costs, in pair order
    xs = model._x
    coefs = np.array([model.objective.terms[xs[i]] for i in range(P)])
    assert np.array_equal(coefs, g["all_costs"])
    # lazy cuts: first 50 flipped triangles (ascending) under the greedy incumbent, x_a + x_b + x_c <= 2 + q_t
    want = g["lazy_violating"][:50]
    assert model._cuts_added == len(want) == len(model.lazy)
    for cut, t in zip(model.lazy, want):
        q = [v for v in cut.expr.terms if v.VarName.startswith("q_tri")]
        assert len(q) == 1 and q[0].VarName == f"q_tri[{t}]" and cut.sense == "<=" and cut.expr.const == -2.0
        picked = sorted(int(v.VarName[2:-1]) for v in cut.expr.terms if v.VarName.startswith("x["))
        tri = g["tri_plain"][t]
        assert sorted(g["pairs"][picked][:, 0].tolist()) == sorted(tri.tolist())
    # match table = the incumbent; violation columns from the device sweeps
    ch = g["greedy_chosen"]
    got = out_df[["aligned_idx", "ref_idx"]].to_numpy()
    assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], ch[np.lexsort((ch[:, 1], ch[:, 0]))][:, :2])
    flipped_nodes = set(g["tri_plain"][g["area_flipped"]].reshape(-1).tolist())
    assert out_df["triangle_violation"].tolist() == [a in flipped_nodes for a in out_df["aligned_idx"]]
    s = var_out["violations"]["violation_summary"]
    assert [s["total_triangles"], s["violated_triangles"], s["total_comparisons"], s["total_violations"]] == g["viol_summary"].tolist()
    assert var_out["triangle_data"]["flipped_triangles"] == g["area_flipped"].tolist()
    assert var_out["lazy_cuts_added"] == len(want) and var_out["lazy_constraints"] is True
    for f in ("matches_df.csv", "aligned_df.csv", "ref_df.csv", "var_out.json", "var_out.npz", "var_out.npy", "matching_model.lp"):
        assert (tmp_path / "out" / f).exists()                       # var_out.npy: the reference's own file, for its readers
    # the pickle-free pair holds the whole of var_out (keys in order, triangle_info insertion order included)
    from same_amd import varout
    from test_varout import assert_same
    assert_same(var_out, varout.load(str(tmp_path / "out")))
    # the reference's file reads back through the numpy-only unpickler; SAME_LEGACY_VAR_OUT=0 leaves it out
    legacy = varout.load_legacy_npy(str(tmp_path / "out" / "var_out.npy"))
    assert list(legacy.keys()) == list(var_out.keys()) and legacy["x"] == var_out["x"]
    os.environ["SAME_LEGACY_VAR_OUT"] = "0"
    try:
        same_amd.run_same(r_df, a_df, cols, outprefix=str(tmp_path / "nopickle"), optim_params=op, gurobi_params=gpar)
    finally:
        del os.environ["SAME_LEGACY_VAR_OUT"]
    assert (tmp_path / "nopickle" / "var_out.json").exists() and not (tmp_path / "nopickle" / "var_out.npy").exists()
    # allowed flip fraction above the observed rate -> no cuts (src/same.py:674-679)
    same_amd.run_same(r_df, a_df, cols, optim_params=op, gurobi_params=dict(init_method="greedy", lazy_allowed_flip_fraction=1.0))
    assert gp.Model.last._cuts_added == 0
    # a global cap
    same_amd.run_same(r_df, a_df, cols, optim_params=op,
                      gurobi_params=dict(init_method="greedy", lazy_allowed_flip_fraction=0.0, lazy_max_cuts=5))
    assert gp.Model.last._cuts_added == 5


def test_run_same_errors(gp):
    import same_amd

    g = load_golden("cfg1_500")
    a_df, r_df, cols = frames_from_golden(g)
    with pytest.raises(ValueError, match="No valid_pairs"):
        same_amd.run_same(r_df, a_df.assign(X=a_df["X"] + 1e7), cols, optim_params=dict(radius=1.0))
    with pytest.raises(ValueError, match="aligned_delaunay_vertex_col"):
        same_amd.run_same(r_df, a_df, cols, aligned_delaunay=np.zeros((0, 3)), aligned_delaunay_vertex_col="nope")


def test_error_conventions_of_the_boundary(gp):
    """the ValueErrors the reference raises at the boundary (src/same.py:258, :451-457, :463-481)."""
    import same_amd

    g = load_golden("cfg1_500")
    a_df, r_df, cols = frames_from_golden(g)
    with pytest.raises(ValueError, match="must have shape"):                       # bad triangle shape
        same_amd.run_same(r_df, a_df, cols, aligned_delaunay=np.zeros((4, 2), dtype=int), optim_params=dict(radius=10))
    other = r_df.copy()
    other["cell_type"] = np.where(np.arange(len(other)) % 2 == 0, "only_in_ref", other["cell_type"])
    with pytest.raises(ValueError, match="Cell type categories differ"):           # sliding windows check the category sets
        same_amd.sliding_window_matching(other, a_df, commonCT=cols, optim_params=dict(radius=10))
    with pytest.raises(ValueError, match="not present as probability/one-hot columns"):   # commonCT inferred from cell_type names
        same_amd.sliding_window_matching(r_df.assign(cell_type="zz"), a_df.assign(cell_type="zz"), optim_params=dict(radius=10))
    with pytest.raises(ValueError, match="cell_type columns were not found"):
        same_amd.sliding_window_matching(r_df.drop(columns=["cell_type"]), a_df.drop(columns=["cell_type"]), optim_params=dict(radius=10))
    with pytest.raises(ValueError, match="Unknown init_method"):
        same_amd.run_same(r_df, a_df, cols, optim_params=dict(radius=10), gurobi_params=dict(init_method="bogus"))


def test_precomputed_triangulation_and_unconstrained_nodes(gp):
    """Caller-supplied triangles in vertex-id space (MetaCell duck type): remap, filter, drop unconstrained nodes."""
    import same_amd
    from scipy.spatial import Delaunay

    g = load_golden("cfg1_500")
    a_df, r_df, cols = frames_from_golden(g)
    a_df = a_df.assign(mc_id=np.arange(len(a_df)) * 7 + 3)
    tris_ids = a_df["mc_id"].to_numpy()[Delaunay(a_df[["X", "Y"]].to_numpy()).simplices]

    class MC:  # what run_same duck-types (src/same.py:891-899)
        metacell_df = a_df
        metacell_delaunay = tris_ids
        metacell_idx_col = "mc_id"

    prep = same_amd.prepare_same_inputs(r_df, MC(), cols, optim_params=dict(radius=10, knn=8, cell_id_col=None), verbose=False)
    assert prep.using_precomputed and prep.optim_params["cell_id_col"] == "mc_id"
    n_after_knn = len(g["kept_aligned"])
    assert prep.n_aligned == n_after_knn - len(prep.unconstrained_nodes) and len(prep.unconstrained_nodes) > 0
    tri = np.asarray(prep.aligned_delaunay)
    assert tri.max() < prep.n_aligned and isinstance(prep.valid_pairs, list)
    assert max(i for i, _ in prep.valid_pairs) == prep.n_aligned - 1
    # every remaining node has at least one triangle that passed radius+angle, by construction
    assert len(prep.costs) == len(prep.valid_pairs) and len(prep.triangle_weights) == len(tri)

    # a triangulation under which EVERY node is unconstrained: nothing is left to match; run_same returns its no-solution
    # pair instead of the reference's IndexError (src/same.py:1253), so a window pipeline carries on
    class Empty(MC):
        metacell_delaunay = np.zeros((0, 3), dtype=np.int64)

    prep0 = same_amd.prepare_same_inputs(r_df, Empty(), cols, optim_params=dict(radius=10, knn=8, cell_id_col=None), verbose=False)
    assert len(prep0.valid_pairs) == 0 and prep0.n_aligned == 0 and len(prep0.costs) == 0 and len(prep0.aligned_delaunay) == 0
    out_df, var_out = same_amd.run_same(r_df, Empty(), cols, optim_params=dict(radius=10, knn=8, cell_id_col=None))
    assert len(out_df) == 0 and var_out == {}


def test_sliding_window_matching(gp, tmp_path):
    import same_amd
    from same_amd import synth

    ref = synth.make_cells(3000, 3, seed=0, side=600.0)
    mov = synth.make_jittered(ref, seed=1)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    op = dict(window_size=300, overlap=100, min_cells_per_window=20, radius=25, knn=6, no_match_penalty=100)
    gpar = dict(init_method="greedy")
    res = same_amd.sliding_window_matching(r_df, m_df, outprefix=str(tmp_path / "sw"), optim_params=op, gurobi_params=gpar)
    plan = same_amd.window_plan(ref["xy"], mov["xy"], 300, 100, 20)
    assert set(res["window_id"]) <= {w["window_id"] for w in plan} and len(res) > 0
    # central-region trim: every kept match lies inside its window's trim box
    for w in plan:
        sub = res[res["window_id"] == w["window_id"]]
        x0, x1, y0, y1 = w["trim"]
        assert ((sub["X"] >= x0) & (sub["X"] < x1) & (sub["Y"] >= y0) & (sub["Y"] < y1)).all()
    assert (tmp_path / "sw" / "matchedDF.csv").exists()
    # resume: nothing is re-run when every window id is already on disk
    calls = []
    again = same_amd.sliding_window_matching(r_df, m_df, outprefix=str(tmp_path / "sw"), optim_params=op, gurobi_params=gpar,
                                             _run_window=lambda **k: calls.append(1) or (pd.DataFrame(), {}))
    done = {w["grid_id"] for w in plan} - set(pd.read_csv(tmp_path / "sw" / "matchedDF.csv")["window_id"])
    assert len(calls) <= len(done) + len(plan) and len(again) == len(res)
    # cell-type sets must agree (src/same.py:446-457)
    with pytest.raises(ValueError, match="Cell type categories differ"):
        same_amd.sliding_window_matching(r_df, m_df.assign(cell_type="zzz"), optim_params=op)


@pytest.mark.parametrize("case", ["synthetic_example", "cfg1_500", "cfg2_small"])
def test_prepare_default_flags_and_priority_filter(case, oracle):
    """The reference's default flags (same-type triangles ignored, re-add pass) and the cell-type-priority prune."""
    import same_amd

    g = load_golden(case)
    a_df, r_df, cols = frames_from_golden(g)
    mad = None if g["params"][2] < 0 else g["params"][2]
    base = dict(radius=g["params"][0], knn=int(g["params"][1]), min_angle_deg=mad, dist_ct_coeff=g["params"][3])
    # ignore_same_type_triangles=True
    prep = same_amd.prepare_same_inputs(r_df, a_df.drop(columns=["size"]), cols, optim_params=base, verbose=False)
    assert np.array_equal(np.array(prep.aligned_delaunay, dtype=np.int64).reshape(-1, 3), g["tri_type"])
    assert (prep.aligned_df["size"] == 1).all()                                  # default size column (src/same.py:934-939)
    w, s = oracle.tri_sign_weight(prep.aligned_df[["X", "Y"]].to_numpy(), np.ones(prep.n_aligned), g["tri_type"])
    assert np.array_equal(np.array(prep.source_signs), w.astype(np.float64)) and list(prep.triangle_weights) == [3] * len(g["tri_type"])
    assert list(prep.triangle_info.keys()) == list(oracle.precompute_triangle_info(
        prep.aligned_df, g["tri_type"], oracle.simplex_map(prep.n_aligned, g["tri_type"])).keys())
    # ignore_knn_if_matched=True (src/same.py:974-976): pairs come from the priority filter, as a list of tuples
    prep2 = same_amd.prepare_same_inputs(r_df, a_df, cols, optim_params=dict(ignore_knn_if_matched=True, **base), verbose=False)
    assert isinstance(prep2.valid_pairs, list)
    assert np.array_equal(np.asarray(prep2.valid_pairs, dtype=np.int64), g["pairs_priority"])
    want = oracle.pair_costs(prep2.aligned_df, prep2.ref_df, g["pairs_priority"], cols, g["params"][3])
    assert prep2.costs == want
    assert prep2.valid_pairs_map[int(g["pairs_priority"][0, 0])][0] == (0, int(g["pairs_priority"][0, 1]))
    # a caller-supplied triangulation as a DataFrame in index space (aligned_delaunay_vertex_col=None -> frame index ids)
    from scipy.spatial import Delaunay
    tri_df = pd.DataFrame(Delaunay(a_df[["X", "Y"]].to_numpy()).simplices)
    prep3 = same_amd.prepare_same_inputs(r_df, a_df, cols, aligned_delaunay=tri_df, optim_params=base, verbose=False)
    assert prep3.using_precomputed and len(prep3.aligned_delaunay) > 0
    prep4 = same_amd.prepare_same_inputs(r_df, a_df, cols, aligned_delaunay=tri_df, ignore_precomputed_triangulation=True,
                                         optim_params=base, verbose=False)
    assert not prep4.using_precomputed and np.array_equal(np.asarray(prep4.aligned_delaunay), np.asarray(prep.aligned_delaunay))


def test_run_same_eager_mode_matches_reference_model(gp, tmp_path):
    """lazy_constraints=False: every variable and constraint of the assembled model equals, one by one and in order, what the
    reference's add_basic_constraints_optimized + add_spatial_constraints_triangle_based emitted into the same recording
    solver double (tests/golden/eager_model.npz; some reference rows are metacells so both match limits occur)."""
    import fake_gurobipy
    import same_amd
    from same_amd import synth

    g = load_golden("eager_model")
    radius, knn = float(g["params"][0]), int(g["params"][1])
    cells = synth.make_cells(140, 3, seed=31)
    r_df = synth.to_frame(cells)
    r_df.loc[r_df.index % 7 == 0, "size"] = 3.0
    a_df = synth.to_frame(synth.make_jittered(cells, seed=32))
    out_df, var_out = same_amd.run_same(r_df, a_df, synth.type_columns(3), outprefix=str(tmp_path),
                                        optim_params=dict(radius=radius, knn=knn, min_angle_deg=15, lazy_constraints=False),
                                        gurobi_params=dict(init_method="greedy"))
    model = gp.Model.last
    assert [v.VarName for v in model.vars] == list(g["var_names"])              # creation order and names
    assert [float(v.lb) for v in model.vars] == list(g["var_lb"])
    assert [np.inf if v.ub is None else float(v.ub) for v in model.vars] == list(g["var_ub"])
    assert [n or "" for n, _ in model.constrs] == list(g["constr_names"])
    got = fake_gurobipy.canonical_constraints(model.constrs)
    assert len(got) == len(g["sense"])
    sense = {-1: "<=", 0: "==", 1: ">="}
    for q, (sn, const, terms) in enumerate(got):
        want_terms = tuple(sorted((str(g["var_names"][v]), float(k)) for v, k in zip(g["term_var"][q], g["term_coef"][q]) if v >= 0))
        assert (sn, const, terms) == (sense[int(g["sense"][q])], float(g["const"][q]), want_terms), q
    assert any(c[1] == -3.0 for c in got[:50]) and any(c[1] == -1.0 for c in got[:50])     # metacell and single-cell limits
    # eager mode: no callback, no lazy bookkeeping; post-solve tables still come out
    assert var_out["lazy_constraints"] is False and var_out["lazy_cuts_added"] == 0
    assert len(var_out["area_penalty_vars"]) == len(g["area_penalty_names"]) == len(g["triangles"])
    assert not hasattr(model.Params, "LazyConstraints")
    assert len(out_df) > 0 and (tmp_path / "var_out.npz").exists() and (tmp_path / "var_out.npy").exists()


@pytest.mark.parametrize("force_comm", [False, True])
def test_bench_contract_line(force_comm):
    """bench.py prints ONE JSON line carrying the driver's keys plus roofline and cpu_baseline (tiny workload; with the
    RCCL branch forced through a size-1 communicator in the second case)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if force_comm:
        env["SAME_BENCH_FORCE_COMM"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1"]
    if force_comm:
        cmd.append("--no-cpu-baseline")
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["metric"] == json.load(open(os.path.join(root, "BASELINE.json"), encoding="utf-8"))["metric"]
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["dtype"] == "f64"
    assert out["value"] > 0 and out["vs_baseline"] is None and "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    if force_comm:
        assert out["cpu_baseline"] is None and out["parity_spot_check"].startswith("transport: rows [0,2000) of rank 0's block")
        assert "RCCL" in out["config"]["parallelism"]
        _check_multi_rank_keys(out, world=1, kind="rccl")
    else:
        assert "rccl" not in out and "gather" not in out
        # the bound as fields, from this run's telemetry; and the kernel against the measured device copy
        if rf["telemetry"].get("available") and rf["telemetry"].get("sclk_steady"):
            assert 0.0 < rf["valu_busy_frac"] <= 1.05 and rf["valu_floor_ms_at_held_clock"] > 0 and rf["held_clock_mhz"] > 500
        assert rf["measured_ceilings"]["device_copy_GBs"] > 0 and rf["frac_of_measured_copy_bw"] > 0
        # T = 20 is bound by fp64 issue: the timed loop stores into a plain block (--spread auto)
        assert rf["output_buffer"]["spread"] is False
        assert all("output_buffer" in e for e in rf["sweep"] if not e.get("opt_in"))
        assert rf["pruned_path"]["cell_pairs_per_s"] > out["value"] and rf["triangle_maps_and_sweeps"]["triangles_per_s"] > 0
        rm = rf["realistic_matching"]        # jittered copy + greedy start: most rows matched, few flips among many checked triangles
        assert rm["matched_rows"] > 0.8 * 0.9 * 4000 and rm["orientation_checked"] > 1000
        assert rm["orientation_flipped"] < 0.2 * rm["orientation_checked"]
        cb = out["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
        assert "equal the oracle bit-for-bit" in out["parity_spot_check"]
        # live ceilings, telemetry and the T sweep of this very run (incl. the labelled fixed-point control)
        assert rf["measured_ceilings"]["same_kernel_T0_store_only_GBs"] > 0 and "telemetry" in rf
        kinds = [(e["dtype"], e["T"]) for e in rf["sweep"]]
        assert ("f64", 3) in kinds and ("f32", 20) in kinds and ("q32->f64", 20) in kinds
        q = [e for e in rf["sweep"] if e["dtype"] == "q32->f64"][0]
        assert q["opt_in"] is True and q["max_rel_diff_vs_exact_on_16_rows"] <= 1e-6


def _check_multi_rank_keys(out, world, kind):
    """What the N > 1 line must carry to explain itself."""
    r = out["rccl"]
    assert r["kind"] == kind and r["nranks"] == world and r["rank"] == 0 and r["consistent"] is True
    assert [e[0] for e in r["every_rank"]] == list(range(world)) and all(e[2] == world for e in r["every_rank"])
    if kind == "rccl":
        assert "ncclCommCount" in r["source"] and r["version"] and r["device"] == 0
    g = out["gather"]
    assert g["bytes_per_rank"] == 4000 * 32 * 12 // (world if out["scaling"] == "strong" else 1) and g["ms"] is not None and g["ms"] >= 0
    assert g["GBs"] is None or g["GBs"] > 0
    assert g["step_ms_with_gather"] > 0 and g["step_ms_without_gather"] > 0 and g["steps_without_gather"] >= 1
    assert abs(out["gather_hidden_ms"] - (g["step_ms_with_gather"] - g["step_ms_without_gather"])) < 1e-9
    lo, mean, hi = out["per_rank_dense_ms"]
    assert 0 < lo <= mean <= hi and len(out["per_rank"]["dense_ms_by_rank"]) == world
    s = out["strong_record"]                      # tiny has no BASELINE-named strong twin: the same shape as ONE problem
    assert s["scaling"] == "strong" and s["value"] > 0 and s["steps"] >= 1 and "ONE problem" in s["config"]["workload"]
    assert s["gather"]["bytes_per_rank"] == -(-4000 // world) * 32 * 12 and s["gather_hidden_ms"] is not None
    assert s["parity_spot_check"].startswith("transport: rows [0,") and len(s["dense_ms_by_rank"]) == world
    assert s["sweep_outputs"]["checked"] > 0


def test_bench_two_ranks_on_one_gpu_carry_the_diagnosis():
    """`bench.py --gpus 2` on this one GPU (RCCL refuses two ranks on one device, so the exchanges go through the host
    transport): the line carries every key the 8-GPU run needs to explain itself, and the embedded ONE-problem record."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAME_RDV_DIR")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup",
                          "1",
                          "--no-cpu-baseline"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "HOST" in out["config"]["parallelism"]
    _check_multi_rank_keys(out, world=2, kind="host")
    assert out["per_rank"]["device_by_rank"] == [0, 0]


@pytest.mark.parametrize("pipeline", ["device", "frames"])
@pytest.mark.parametrize("cost_dtype", ["float64", "float32"])
def test_window_arrays_equal_prepared_windows(cost_dtype, pipeline):
    """The column pipeline (windows.iter_window_arrays: no DataFrame per window; what bench.py --workload cfg5 runs) yields, for
    every window of a plan -- thin edge strips, an empty window and integer / float size columns included -- exactly the
    artefacts the frame pipeline (api.iter_prepared_windows, itself pinned against the reference's run_same) computes: the
    same cells, pairs, costs, kept triangles in the same order, weights and signs."""
    import same_amd
    from same_amd import synth
    from same_amd.windows import Section, iter_window_arrays, window_plan

    T = 5
    ref = synth.make_cells(40_000, T, seed=30)
    mov = synth.make_jittered(ref, seed=31)
    mov["xy"][:300] += 5000.0                                    # a far-away clump: windows whose prune finds nothing
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    m_df["size"] = np.where(np.arange(len(m_df)) % 3 == 0, 2, 1) if cost_dtype == "float32" else m_df["size"].astype(float) * 1.5
    cols = synth.type_columns(T)
    plan = window_plan(r_df[["X", "Y"]].to_numpy(), m_df[["X", "Y"]].to_numpy(), 700, 200, 10)
    # + a window over the clump: aligned cells, no reference cells
    plan = plan[::2] + [dict(plan[0], box=(5000.0, 7100.0, 5000.0, 7100.0))]
    op = dict(radius=30, knn=6, min_angle_deg=12, dist_ct_coeff=1.5, hip_cost_dtype=cost_dtype)
    frames = list(same_amd.iter_prepared_windows(r_df, m_df, cols, plan, optim_params=op, pipeline=pipeline))
    arrays = list(iter_window_arrays(Section.from_frame(r_df, cols), Section.from_frame(m_df, cols), plan, radius=30, knn=6,
                                     dist_ct_coeff=1.5, min_angle_deg=12, ignore_same_type_triangles=True, cost_dtype=cost_dtype))
    assert len(frames) == len(arrays) == len(plan) > 10
    errors = 0
    for (w, prep), wa in zip(frames, arrays):
        assert wa.window is w
        if isinstance(prep, Exception):
            assert isinstance(wa.error, ValueError) and str(wa.error) == str(prep)
            errors += 1
            continue
        assert wa.error is None
        assert np.array_equal(m_df.index.to_numpy()[wa.rows_m], prep.aligned_df["__orig_idx"].to_numpy())
        assert np.array_equal(r_df.index.to_numpy()[wa.rows_r], prep.ref_df["__orig_idx"].to_numpy())
        assert np.array_equal(wa.pairs, np.asarray(prep.valid_pairs, dtype=np.int64).reshape(-1, 2))
        assert np.array_equal(wa.costs, prep.costs_array) and wa.costs.dtype == np.float64
        assert np.array_equal(wa.triangles, prep.triangles_array)
        assert np.array_equal(wa.signs, prep.signs_array)
        assert wa.weights.dtype == prep.weights_array.dtype and np.array_equal(wa.weights, prep.weights_array)
        assert np.array_equal(wa.axy, prep.aligned_df[["X", "Y"]].to_numpy()) and np.array_equal(wa.rxy, prep.ref_df[["X", "Y"]].to_numpy())
    assert errors >= 1


@pytest.mark.parametrize("cost_dtype", ["float64", "float32"])
def test_device_windows_equal_the_column_pipeline(cost_dtype):
    """The window path with both sections resident on the device (windows.iter_device_windows over csrc/window.hip: subsetting,
    prune, costs, compaction, signs, greedy incumbent and the three sweeps in two calls per window) against the column pipeline
    plus the host-buffer entry points -- every window of a plan, thin strips, a window without reference cells, integer and
    float sizes: the same kept cells, pairs (reference cells compared by section row: the device path does not renumber them),
    costs, triangles, signs, weights, match, per-cell flags and counters."""
    from window_check import check_window

    from same_amd import ops, synth
    from same_amd import windows as W

    T = 5
    ref = synth.make_cells(40_000, T, seed=30)
    mov = synth.make_jittered(ref, seed=31)
    mov["xy"][:300] += 5000.0
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    m_df["size"] = np.where(np.arange(len(m_df)) % 3 == 0, 2, 1) if cost_dtype == "float32" else m_df["size"].astype(float) * 1.5
    cols = synth.type_columns(T)
    plan = W.window_plan(r_df[["X", "Y"]].to_numpy(), m_df[["X", "Y"]].to_numpy(), 700, 200, 10)
    plan = plan[::2] + [dict(plan[0], box=(5000.0, 7100.0, 5000.0, 7100.0))]
    ref_sec, mov_sec = W.Section.from_frame(r_df, cols), W.Section.from_frame(m_df, cols)
    dref, dmov = W.DeviceSection(ref_sec, cost_dtype), W.DeviceSection(mov_sec, cost_dtype)
    kw = dict(radius=30, knn=6, dist_ct_coeff=1.5, min_angle_deg=12, ignore_same_type_triangles=True)
    arrays = list(W.iter_window_arrays(ref_sec, mov_sec, plan, cost_dtype=cost_dtype, **kw))
    # a penalty some rows' best pair does not beat: `prefer` is not all ones
    penalty, errors, windows, n_matched, n_rows = 0.006, 0, 0, 0, 0
    for wa, dw in zip(arrays, W.iter_device_windows(ref_sec, mov_sec, dref, dmov, plan, no_match_penalty=penalty, fetch_triangles=True,
                                                    **kw)):
        assert dw.window is wa.window
        if wa.error is not None:
            assert isinstance(dw.error, ValueError) and str(dw.error) == str(wa.error)
            errors += 1
            continue
        assert dw.error is None
        n_matched, n_rows = n_matched + check_window(W, ops, wa, dw, penalty), n_rows + wa.n_aligned
        windows += 1
    assert errors >= 1 and windows > 10 and 0.2 * n_rows < n_matched < 0.95 * n_rows
    # other filter settings (no angle rule; same-type triangles kept), triangles only
    for kw2 in (dict(kw, min_angle_deg=None), dict(kw, ignore_same_type_triangles=False), dict(kw, min_angle_deg=40, radius=18)):
        some = plan[3:9]
        for wa, dw in zip(W.iter_window_arrays(ref_sec, mov_sec, some, cost_dtype=cost_dtype, **kw2),
                          W.iter_device_windows(ref_sec, mov_sec, dref, dmov, some, no_match_penalty=penalty, fetch_triangles=True, **kw2)):
            assert (wa.error is None) == (dw.error is None)
            if wa.error is None:
                assert np.array_equal(dw.triangles, wa.triangles) and dw.n_triangles == len(wa.triangles)
                assert np.array_equal(dw.state.fetch(W._W_SIGNS), wa.signs.astype(np.int8))
    # a cosine "at" the threshold (forced here by a huge tolerance) is left to the host: nothing stays on the device for finish()
    from scipy.spatial import Delaunay

    from same_amd._lib import SameHipError
    from same_amd.triangles import cos_threshold, filter_triangles_by_radius

    st = W.DeviceWindow()
    n_m, n_r, kept, n_pairs = st.stage(dmov, dref, plan[5]["box"], 30, 6, 1.5)
    assert kept > 100 and n_pairs > kept
    axy, rows = st.fetch(W._W_ALIGNED_XY), st.fetch(W._W_ALIGNED_ROWS)
    simplices = Delaunay(axy).simplices
    en, thr = cos_threshold(12)
    k0, k1, near, m0, f0, s0 = st.filter_finish(simplices, 30, en, thr, 4.0, True, penalty)
    assert near > 0 and st.n_triangles == 0 and m0 is None and f0 is None and s0 is None
    with pytest.raises(SameHipError):
        st.fetch(W._W_TRIANGLES)                            # nothing was left on the device
    host = filter_triangles_by_radius(axy, simplices, 30, ignore_same_type_triangles=True, min_angle_deg=12, verbose=False,
                                      _rows_as_array=True,
                                      _type_id=mov_sec.type_id[rows])
    k0, k1, near, a, b, c = st.filter_finish(simplices, 30, en, thr, 0.0, True, penalty)
    assert near == 0 and k0 + k1 == len(host) and k1 > 0
    assert np.array_equal(st.fetch(W._W_TRIANGLES), host)
    a2, b2, c2 = st.finish(host, penalty)                  # the same triangles passed in from the host (prefiltered): the same answers
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and c == c2
    st.close()
    dref.close()
    dmov.close()


def test_allgather_table_over_a_size_one_rccl_communicator():
    """dist.allgather_table (the cfg 5 exchange: per-rank match tables as ONE device all-gather) through a real ncclAllGather on
    a size-1 communicator: columns of different widths and an empty table come back bit for bit; ragged or non-numeric
    columns are refused.  (Two ranks go through the same code over the host transport in test_bench_cfg5_windows_line[2].)"""
    from same_amd import _lib
    from same_amd.dist import RcclGroup, allgather_table
    from same_amd.rendezvous import HostGroup

    ctx = _lib.Context(0)
    comm = RcclGroup(ctx, 1, 0, lambda b: b)
    rng = np.random.default_rng(1)
    try:
        with HostGroup(0, 1) as group:
            for n in (0, 1, 1000, 123_457):
                t = {"a": rng.integers(-2 ** 62, 2 ** 62, n), "x": rng.random(n), "v": (rng.random(n) < 0.5).astype(np.uint8),
                     "w": rng.integers(0, 300, n).astype(np.int32)}
                (back,) = allgather_table(ctx, comm, group, t)
                assert list(back) == list(t) and all(back[c].dtype == t[c].dtype and np.array_equal(back[c], t[c]) for c in t)
            with pytest.raises(ValueError):
                allgather_table(ctx, comm, group, {"a": np.arange(3), "b": np.arange(4)})
            with pytest.raises(ValueError):
                allgather_table(ctx, comm, group, {"a": np.array(["x", "y"])})
            assert allgather_table(ctx, None, group, {"a": np.arange(3)})[0]["a"].tolist() == [0, 1, 2]
    finally:
        comm.close()
        ctx.close()


@pytest.mark.parametrize("world", [1, 2])
def test_bench_cfg5_windows_line(world):
    """`bench.py --workload cfg5` (BASELINE cfg 5 at reduced size): whole windows dealt to the ranks, fp32 costs, all sweeps per
    window, tables exchanged once and merged; the line reports windows/s per rank and the host-glue share, and at N=1 checks
    up to four windows against the oracle."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAME_RDV_DIR")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg5", "--cfg5-cells", "60000", "--steps", "1", "--warmup", "1",
           "--gpus", str(world)]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["metric"] == json.load(open(os.path.join(root, "BASELINE.json"), encoding="utf-8"))["metric"]
    assert out["n_gpus"] == world and out["dtype"] == "f32" and out["value"] > 0
    assert out["config"]["workload"].startswith("cfg5: 60000-cell section")
    pr = out["per_rank"]
    assert len(pr["windows"]) == world and sum(pr["windows"]) >= 4 and all(v > 0 for v in pr["windows_per_s"])
    assert all(0.0 <= v < 1.0 for v in pr["host_glue_share"]) and out["windows_per_s"] > 0 and out["merged_matches"] > 1000
    assert out["config"]["pipeline"].startswith("device: same_amd.sliding_window_incumbent(merge=True) on resident frames")
    am = out["amdahl"]
    assert out["amdahl_bound_at_8_ranks"] == am["value"] > 1.0 and len(am["at_8_ranks"]["seam_rows_by_rank"]) == 8
    assert 0.0 < am["at_8_ranks"]["seam_rows_share"] < 1.0 and am["at_8_ranks"]["common_seam_step_s"] > 0
    assert out["product_function"].startswith("same_amd.sliding_window_incumbent")
    assert any(k.startswith("subset + prune") for k in out["stages_rank0"]) and out["library_calls_rank0_top"][0]["seconds"] > 0
    assert {"same_window_stage", "same_window_filter_finish"} <= {e["entry_point"] for e in out["library_calls_rank0_top"]}
    # the diagnostic pass with the triangulations remembered (not a throughput; it says what is left once Qhull is out of the picture)
    assert out["windows_per_s_triangulations_given"] > 0 and len(pr["windows_per_s_triangulations_given"]) == world
    assert out["qhull"]["helpers"] >= 0 and out["qhull"]["waiting_s_per_step_rank0"] >= 0 and out["qhull"]["cpu_budget"] >= 1
    assert 0.0 <= out["python_share"] <= out["host_glue_share"] and len(pr["python_share"]) == world
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    if world == 1:
        assert out["cpu_baseline"]["kind"] == "port" and "equal the oracle bit-for-bit" in out["parity_spot_check"]
        assert "through the device-resident window path" in out["parity_spot_check"]
        # the reference's own signature, timed in the same run: all windows with the incumbent standing in for the solver half, a corner
        # of the section with a do-nothing gurobipy (run_same's own Python around the solver)
        ap = out["api_path"]
        assert out["api_path_windows_per_s"] == ap["api_path_windows_per_s"] > 0 and ap["pipeline"] == "device" and ap["matches"] > 1000
        sd = ap["with_solver_double"]
        assert sd["windows"] >= 1 and sd["seconds_per_window"] > sd["solver_side_python_s_per_window"] > 0
        # the same function's general route on host frames merges to the same table
        res = subprocess.run(cmd + ["--cfg5-pipeline", "frames", "--no-cpu-baseline"], env=env, cwd=root, capture_output=True, text=True,
                             timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        col = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
        assert col["config"]["pipeline"].startswith("frames:") and col["merged_matches"] == out["merged_matches"]
        assert any(k.startswith("prune") for k in col["stages_rank0"]) and col["api_path"]["pipeline"] == "frames"
    else:
        assert out["cpu_baseline"] is None


@pytest.mark.parametrize("world", [1, 2])
def test_bench_line_embeds_a_cfg5_record(world):
    """Every line of the default workload carries BASELINE cfg 5 as a sub-record measured IN THE SAME JOB -- its ranks, their
    contexts, the communicator; no child process -- at one rank or N (`--embed-cfg5`, automatic with the default workload; forced
    here on the tiny one at a reduced section): windows/s per rank, the pipeline, the Qhull record with the helper budget divided
    by the ranks on the host, the table all-gather's time, the merged table's size, and at one rank the oracle check of four windows."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "SAME_RDV_DIR",
                                                            "SAME_QHULL_WORKERS")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-extras",
           "--no-strong-record", "--embed-cfg5", "on", "--cfg5-cells", "60000", "--gpus", str(world)]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])["cfg5"]
    assert "error" not in rec, rec
    assert rec["n_gpus"] == world and rec["windows_per_s"] > 0 and rec["merged_matches"] > 1000 and rec["pipeline"].startswith("device:")
    assert rec["workload"].startswith("cfg5: 60000-cell section")
    pr = rec["per_rank"]
    assert len(pr["windows"]) == len(pr["windows_per_s"]) == len(pr["qhull_helpers"]) == world and sum(pr["windows"]) >= 4
    assert rec["qhull"]["ranks_on_this_host"] == world and rec["qhull"]["helpers_all_ranks"] == sum(pr["qhull_helpers"]) <= max(24, world)
    calls = rec["runtime_calls_per_window"]
    assert calls["launches"] <= 30 and calls["fills"] <= 4 and calls["copies"] <= 4 and calls["waits"] <= 3
    # the window merge is per rank: only seam rows travel; the record prices what is NOT dealt with the windows and bounds 8 ranks from it
    assert rec["deal"] == "block" and 0.0 <= rec["unsharded_s_per_step"] <= rec["serial_tail_s_per_step"] + 1e-9
    assert rec["serial_tail_s_per_step"] <= rec["after_windows_s_per_step"] + 1e-9 and rec["seam_wait_s_per_step"] >= 0.0
    assert len(pr["serial_tail_s_per_step"]) == len(pr["merged_rows"]) == world and sum(pr["merged_rows"]) == rec["merged_matches"]
    if world == 1:
        assert "through the device-resident window path" in rec["parity_spot_check"] and rec["seam_exchange"] is None
        assert rec["unsharded_s_per_step"] == 0.0
    else:
        sx = rec["seam_exchange"]
        sent = sx["rows_sent_per_step_by_rank"]
        assert len(sent) == world and all(0 < v < 0.5 * rec["merged_matches"] for v in sent)
        assert rec["unsharded_s_per_step"] > 0.0


def test_bench_step_with_the_fixed_point_dense_build():
    """`bench.py --dense q32`: the step's dense build is the opt-in fixed-point kernel; the line says so (dtype, kernel name, note)
    and its own check -- twin-equal and within 1e-6 relative of the exact costs on every sampled pair -- passed."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1",
                          "--dense", "q32",
                          "--no-extras"], cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert out["dtype"] == "u32+f64" and "dense_cost_q32_kernel" in out["roofline"]["kernel"] and "--dense q32" in out["roofline"]["note"]
    assert "within 1e-6 relative of the exact fp64 costs" in out["parity_spot_check"] and out["roofline"]["valu_fp64"] is None


# ---------------------------------------------------------------------------------------------------------------
# The whole boundary functions against the REFERENCE'S OWN run_same / sliding_window_matching, both driven through the
# same recording solver double (tests/golden/run_same_mock.npz, written by tools/gen_golden.py runsame).
def _mock_inputs():
    from scipy.spatial import Delaunay
    from same_amd import synth

    cells = synth.make_cells(300, 4, seed=41)
    r_df = synth.to_frame(cells)
    a_df = synth.to_frame(synth.make_jittered(cells, seed=42))
    ids = np.arange(len(a_df)) * 5 + 2
    a_pre = a_df.assign(mc_id=ids)
    tri_all = Delaunay(a_pre[["X", "Y"]].values).simplices
    keep = ~np.isin(tri_all, np.arange(0, len(a_pre), 9)).any(axis=1)
    cases = {
        "lazy_greedy": (a_df, dict(radius=20, knn=4), dict(init_method="greedy", lazy_allowed_flip_fraction=0.0,
                                                           lazy_max_cuts_per_incumbent=25), {}),
        "priority_hungarian": (a_df, dict(radius=20, knn=4, ignore_knn_if_matched=True, min_angle_deg=None,
                                          ignore_same_type_triangles=False,
                                          dist_ct_coeff=2.5, no_match_penalty=40, penalty_coeff=3.0, delaunay_penalty=7.0),
                               dict(init_method="hungarian", lazy_allowed_flip_fraction=0.0, lazy_max_cuts=9, time_limit=60, mip_focus=1,
                                    cuts=2, heuristics=0.2), {}),
        "eager": (a_df, dict(radius=14, knn=3, lazy_constraints=False), dict(init_method="greedy"), {}),
        "max_matches2": (a_df, dict(radius=20, knn=5, max_matches=2, min_angle_deg=0),
                         dict(init_method="greedy", lazy_allowed_flip_fraction=0.0), {}),
        "multiplier": (a_df, dict(radius=25, knn=6, ref_metacell_match_multiplier=2),
                       dict(init_method="hungarian", init_hungarian_max_n=100, lazy_allowed_flip_fraction=0.0), {}),
        "precomputed": (a_pre, dict(radius=20, knn=4), dict(init_method="greedy", lazy_allowed_flip_fraction=0.0),
                        dict(aligned_delaunay=ids[tri_all[keep]], aligned_delaunay_vertex_col="mc_id")),
    }
    return r_df, synth.type_columns(4), cases


@pytest.mark.parametrize("tag", ["lazy_greedy", "priority_hungarian", "eager", "precomputed", "max_matches2", "multiplier"])
def test_run_same_equals_reference_run_same(gp, tag, tmp_path, monkeypatch):
    import os
    import run_same_record as rec
    import same_amd

    monkeypatch.chdir(tmp_path)
    g = load_golden("run_same_mock")
    r_df, cols, cases = _mock_inputs()
    a_df, op, gpar, extra = cases[tag]
    outprefix = str(tmp_path / tag)
    if tag == "multiplier":
        r_df = r_df.copy()
        r_df.loc[r_df.index % 5 == 0, "size"] = 4.0
    out_df, var_out = same_amd.run_same(r_df.copy(), a_df.copy(), cols, outprefix=outprefix, optim_params=same_amd.init_optim_params(**op),
                                        gurobi_params=same_amd.init_gurobi_params(**gpar), **extra)
    got = rec.record_run(out_df, var_out, gp.Model.last)
    # files written: the reference's set (its pickled var_out.npy, src/same.py:1455-1462, included) plus the pickle-free
    # var_out.json + var_out.npz pair beside it (same_amd/varout.py; INTEGRATION.md "deliberately differs")
    written = sorted(os.listdir(outprefix))
    if "var_out.json" in written:
        assert "var_out.npz" in written and "var_out.npy" in written
        written = sorted(f for f in written if f not in ("var_out.json", "var_out.npz"))
    got["files"] = np.array(written, dtype=str)
    rec.assert_same_record(got, g, prefix=f"{tag}/")
    # what was written loads back through load_matching_results (src/helpers.py:667-689) and flattens to the same record
    lv, la, lr, lm = same_amd.load_matching_results(outprefix)
    if var_out:
        back = rec.record_run(lm, lv, gp.Model.last)
        for k in ("x", "no_match_vars", "area_penalty_vars", "viol_summary", "flipped_triangles", "areas_before", "info_keys",
                  "out__aligned_idx",
                  "out__ref_idx", "out__triangle_violation"):
            assert np.array_equal(np.asarray(back[k]).astype(float), np.asarray(got[k]).astype(float), equal_nan=True), k
        assert len(la) == len(var_out["no_match_vars"]) and len(lr) == len(var_out["penalty_vars"])


@pytest.mark.parametrize("pipeline", ["device", "frames"])
def test_sliding_window_equals_reference(gp, tmp_path, monkeypatch, pipeline):
    """sliding_window_matching against the reference's own (recording solver double on both sides), on the default pipeline -- both
    frames resident on the device for the whole loop, a window's frames made from the device's row lists -- and on the host-frame one."""
    import os
    import pandas as pd
    import run_same_record as rec
    import same_amd
    from same_amd import synth

    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SAME_WINDOW_PIPELINE", pipeline)
    assert same_amd.window_api.window_pipeline() == pipeline
    g = load_golden("run_same_mock")
    cells = synth.make_cells(1500, 3, seed=51)
    r_big = synth.to_frame(cells)
    m_big = synth.to_frame(synth.make_jittered(cells, seed=52))
    m_big = m_big[~((m_big["X"] < 120) & (m_big["Y"] < 170) & (np.arange(len(m_big)) % 4 != 0))].reset_index(drop=True)
    swp = dict(radius=20, knn=4, window_size=150, overlap=40, min_cells_per_window=60)
    gpar = dict(init_method="greedy", lazy_allowed_flip_fraction=0.0)
    sw = str(tmp_path / "sw")
    res = same_amd.sliding_window_matching(r_big.copy(), m_big.copy(), commonCT=synth.type_columns(3), outprefix=sw, optim_params=dict(swp),
                                           gurobi_params=dict(gpar))
    rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("res", res).items()}, g, prefix="sw/res_")
    assert sorted(d for d in os.listdir(sw) if d.startswith("window_")) == list(g["sw/dirs"])
    assert len(pd.read_csv(os.path.join(sw, "matchedDF.csv"))) == int(g["sw/csv_rows"][0])
    res2 = same_amd.sliding_window_matching(r_big.copy(), m_big.copy(), commonCT=synth.type_columns(3), outprefix=sw,
                                            optim_params=dict(swp),
                                            gurobi_params=dict(gpar))
    assert len(res2) == int(g["sw/resume_rows"][0])
    res3 = same_amd.sliding_window_matching(r_big.copy(), m_big.copy(), optim_params=dict(swp, window_size=220, overlap=60),
                                            gurobi_params=dict(gpar))
    rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("res", res3).items()}, g, prefix="sw_infer/res_")


def _sw_inputs():
    from same_amd import synth

    cells = synth.make_cells(1500, 3, seed=51)
    r_big = synth.to_frame(cells)
    m_big = synth.to_frame(synth.make_jittered(cells, seed=52))
    m_big = m_big[~((m_big["X"] < 120) & (m_big["Y"] < 170) & (np.arange(len(m_big)) % 4 != 0))].reset_index(drop=True)
    return r_big, m_big, synth.type_columns(3)


def _sw_inputs_crowded():
    """The same reference section against TWO jittered copies of it: every reference cell has two suitors, and where they sit in
    different windows' central regions both windows propose it -- the overlaps disagree, which is what the window merge settles."""
    from same_amd import synth

    r_big, m_big, cols = _sw_inputs()
    other = synth.to_frame(synth.make_jittered(synth.make_cells(1500, 3, seed=51), seed=53))
    m2 = pd.concat([m_big, other], ignore_index=True)
    m2["Cell_Num_Old"] = np.arange(len(m2)) * 3 + 1
    return r_big, m2, cols


def _assert_incumbent_equals_golden(res, g, prefix, with_ref_idx=True):
    """Every column of the reference's result table that is defined without a solver, and the column order.  `filtered_violation`
    is the XY-order flag here and the flag intersected with the solver's penalised triangles there; `run_time` is the double's."""
    want_cols = [str(c) for c in g[f"{prefix}/res_columns"]]
    assert list(res.columns) == [c for c in want_cols if with_ref_idx or c != "ref_idx"], prefix
    for c in want_cols:
        if c in ("filtered_violation", "run_time") or (c == "ref_idx" and not with_ref_idx):
            continue
        want, got = g[f"{prefix}/res__{c}"], res[c].to_numpy()
        assert np.array_equal(got.astype(want.dtype) if want.dtype.kind in "fiub" else got.astype(str), want), (prefix, c)
    assert len(res) > 500


@pytest.mark.parametrize("route,pipeline", [("device", "device"), ("general", "device"), ("general", "frames")])
def test_incumbent_table_equals_the_reference_window_loop(route, pipeline, tmp_path):
    """The solver-free product function END TO END against the reference's own sliding_window_matching (tests/golden/run_same_mock.npz,
    `sw` and `sw_infer`: run there with the recording solver double, which takes the greedy MIP start as the incumbent): per window
    stage -> Qhull -> filter -> greedy incumbent -> sweeps, matched cells inside the central trim -- every column of the reference's
    result table that does not need a solver, row for row, through the device route (two library calls per window, one gather at the
    end) and through the general route on either pipeline."""
    import same_amd

    g = load_golden("run_same_mock")
    r_big, m_big, cols = _sw_inputs()
    for prefix, ws, ov, ct in (("sw", 150, 40, cols), ("sw_infer", 220, 60, None)):
        op = dict(radius=20, knn=4, window_size=ws, overlap=ov, min_cells_per_window=60)
        res, stats = same_amd.sliding_window_incumbent(r_big.copy(), m_big.copy(), commonCT=ct, optim_params=dict(op),
                                                       window_local_indices=True,
                                                       return_stats=True, _route=route, _pipeline=pipeline)
        _assert_incumbent_equals_golden(res, g, prefix)
        assert len(stats) == res["window_id"].nunique() and all(s["pairs"] > 0 and s["triangles"] > 0 for s in stats)
    if route == "device":
        # without the window-local reference index (the default: no pair list comes back), with two worker threads, into a directory
        res2 = same_amd.sliding_window_incumbent(r_big.copy(), m_big.copy(), optim_params=dict(op), workers=2,
                                                 outprefix=str(tmp_path / "inc"))
        assert "ref_idx" not in res2.columns and res2.equals(res.drop(columns=["ref_idx"]))
        assert len(pd.read_csv(tmp_path / "inc" / "matchedDF.csv")) == len(res2)
        # a second call finds every window in the file and runs none
        res3 = same_amd.sliding_window_incumbent(r_big.copy(), m_big.copy(), optim_params=dict(op), outprefix=str(tmp_path / "inc"))
        assert len(res3) == len(res2)


def test_incumbent_table_routes_agree():
    """The two routes of same_amd.incumbent on a plan with thin edge strips, fp32 costs, integer sizes and a penalty some rows do not beat:
    the same table, bit for bit; the cell-type-priority filter (general route only) runs on both pipelines alike; a window whose prune
    finds nothing raises run_same's ValueError on every route."""
    import same_amd
    from same_amd import synth

    ref = synth.make_cells(30_000, 5, seed=30)
    mov = synth.make_jittered(ref, seed=31)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    m_df["size"] = np.where(np.arange(len(m_df)) % 3 == 0, 2, 1)
    op = dict(radius=30, knn=6, min_angle_deg=12, dist_ct_coeff=1.5, hip_cost_dtype="float32", window_size=700, overlap=200,
              no_match_penalty=0.006,
              min_cells_per_window=10)
    cols = synth.type_columns(5)
    fast = same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op), window_local_indices=True, _route="device")
    for pipeline in ("device", "frames"):
        slow = same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op), window_local_indices=True,
                                                 _route="general",
                                                 _pipeline=pipeline)
        assert list(fast.columns) == list(slow.columns) and len(fast) == len(slow) > 5000
        for c in fast.columns:
            assert np.array_equal(fast[c].to_numpy(), slow[c].to_numpy()), (pipeline, c)
    assert 0.2 * len(m_df) < fast["Aligned_Cell_Num_Old"].nunique() < 0.98 * len(m_df)
    assert fast["triangle_violation"].any() and fast["filtered_violation"].any()
    pri = [same_amd.sliding_window_incumbent(r_df, m_df, commonCT=cols, optim_params=dict(op, ignore_knn_if_matched=True), _pipeline=p)
           for p in ("device", "frames")]
    assert pri[0].equals(pri[1]) and len(pri[0]) > 5000 and not pri[0].equals(fast.drop(columns=["ref_idx"]))
    # a clump of aligned cells and a clump of reference cells that share a window 50 000 units away, farther apart than the radius
    rng = np.random.default_rng(5)
    far = m_df.copy()
    far.loc[far.index[:300], ["X", "Y"]] = 50_000.0 + rng.uniform(0, 300, (300, 2))
    near_refs = r_df.iloc[:12].copy()
    near_refs[["X", "Y"]] = 50_500.0 + rng.uniform(0, 50, (12, 2))
    r_far = pd.concat([r_df, near_refs], ignore_index=True)
    for kw in (dict(_route="device"), dict(_route="general", _pipeline="frames")):
        with pytest.raises(ValueError, match="No valid_pairs after KNN filtering"):
            same_amd.sliding_window_incumbent(r_far, far, commonCT=cols, optim_params=dict(op), **kw)


def test_resident_frames_serve_several_jobs(gp, tmp_path, monkeypatch):
    """`resident_frames`: the two frames uploaded and binned once, then jobs with other radii, penalties, window sizes and cost types over
    them -- each result equals the one-off call's; sliding_window_matching takes the same object; foreign frames are refused."""
    import same_amd
    from same_amd import _lib

    monkeypatch.chdir(tmp_path)
    r_big, m_big, cols = _sw_inputs()
    jobs = [dict(radius=20, knn=4, window_size=150, overlap=40, min_cells_per_window=60),
            dict(radius=14, knn=6, window_size=150, overlap=40, min_cells_per_window=60, no_match_penalty=5.0),
            dict(radius=20, knn=4, window_size=220, overlap=60, min_cells_per_window=60, hip_cost_dtype="float32")]
    ctx = _lib.default_context()
    with same_amd.resident_frames(r_big, m_big) as res:
        for op in jobs:
            got = same_amd.sliding_window_incumbent(res, res, commonCT=cols, optim_params=dict(op), window_local_indices=True)
            want = same_amd.sliding_window_incumbent(r_big, m_big, commonCT=cols, optim_params=dict(op), window_local_indices=True)
            assert len(got) > 300 and got.equals(want), op
        # sections per (cost type, window grid); plans per (size, overlap, min cells)
        assert len(res._frames) == 2 and len(res.plans) == 2
        before = ctx.stats()
        again = same_amd.sliding_window_incumbent(res, m_big, commonCT=cols, optim_params=dict(jobs[0]), window_local_indices=True)
        assert again.equals(same_amd.sliding_window_incumbent(r_big, m_big, commonCT=cols, optim_params=dict(jobs[0]),
                                                              window_local_indices=True))
        assert len(res._frames) == 2 and ctx.stats()["launches"] > before["launches"]
        # the reference's signature on the same resident frames
        g = load_golden("run_same_mock")
        import run_same_record as rec
        out = same_amd.sliding_window_matching(res, res, commonCT=cols, optim_params=dict(jobs[0]),
                                               gurobi_params=dict(init_method="greedy", lazy_allowed_flip_fraction=0.0))
        rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("res", out).items()}, g, prefix="sw/res_")
        with pytest.raises(ValueError, match="other ref / moving objects"):
            same_amd.sliding_window_incumbent(res, m_big.copy(), commonCT=cols, optim_params=dict(jobs[0]))
    assert res._frames == {}


def _sharded_incumbent_worker(rank, world, deal, out_dir):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from test_gpu_run_same import _sw_inputs, _sw_inputs_crowded
    import same_amd
    from same_amd.dist import MergeChannel, sharded_merged_window_incumbent, sharded_sliding_window_incumbent
    from same_amd.rendezvous import HostGroup

    r_big, m_big, cols = _sw_inputs()
    _r, m_two, _c = _sw_inputs_crowded()
    op = dict(radius=20, knn=4, window_size=150, overlap=40, min_cells_per_window=60)
    # an overlap far below the prune's reach: neighbouring windows see different suitors of a cell and disagree
    op2 = dict(radius=25, knn=6, window_size=100, overlap=4, min_cells_per_window=20)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), SAME_RDV_DIR=os.path.join(out_dir, "rdv"))
    if world == 2:     # the package's own wrappers over its plain-Python host group (RANK / WORLD_SIZE / SAME_RDV_DIR)
        # every rank: the whole table
        part = sharded_sliding_window_incumbent(r_big, m_big, commonCT=cols, optim_params=dict(op), deal=deal)
        # every rank: its part of the merge
        merged = sharded_merged_window_incumbent(r_big, m_two, commonCT=cols, optim_params=dict(op2), deal=deal)
    else:              # the share of one rank, as a launcher with its own exchange would take it
        part = same_amd.sliding_window_incumbent(r_big, m_big, commonCT=cols, optim_params=dict(op), _shard=(rank, world, deal))
        with HostGroup() as g:
            merged = same_amd.sliding_window_incumbent(r_big, m_two, commonCT=cols, optim_params=dict(op2), merge=True, _route="general",
                                                       _pipeline="frames" if rank == 1 else None, _shard=(rank, world, deal),
                                                       _merge_channel=MergeChannel(g))
            g.barrier()
    part.to_pickle(os.path.join(out_dir, f"part{rank}.pkl"))
    merged.to_pickle(os.path.join(out_dir, f"merged{rank}.pkl"))
    # the solver loop's form: sliding_window_matching per rank (the incumbent standing in for the solver half), then the merge
    if world == 2:
        from same_amd.dist import sharded_merged_window_matches
        from same_amd.incumbent import incumbent_of_prepared

        stand_in = lambda prep, _o: (incumbent_of_prepared(prep, cols, False)[0], {})
        solved = sharded_merged_window_matches(r_big, m_two, commonCT=cols, optim_params=dict(op2), deal=deal, _solve=stand_in)
        solved.to_pickle(os.path.join(out_dir, f"solved{rank}.pkl"))


@pytest.mark.parametrize("world,deal", [(2, "block"), (2, "round_robin"), (3, "block"), (3, "round_robin")])
def test_sharded_incumbent_parts_make_the_single_process_table(tmp_path, world, deal):
    """sliding_window_incumbent over several processes on the one GPU, each running its share of the plan under either deal: through
    dist.sharded_sliding_window_incumbent (world 2: every rank ends with the whole table) and through `_shard=(rank, world, deal)` by hand
    (world 3: the parts, put back into plan order by their `__plan_pos`) -- the single process's table row for row.  And the window
    merge dealt the same way (dist.sharded_merged_window_incumbent on the device route; `merge=True` with a MergeChannel on the general
    route, one rank on the frames pipeline): the parts, joined, are merge_window_matches_unique_ref of the single process's table
    (src/helpers.py:692-815), and every part is in its order."""
    import multiprocessing as mp
    import same_amd
    from same_amd.merge import join_merged_parts, merge_window_matches_unique_ref

    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sharded_incumbent_worker, args=(rank, world, deal, str(tmp_path))) for rank in range(world)]
    [p.start() for p in procs]
    [p.join(600) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    parts = [pd.read_pickle(tmp_path / f"part{rank}.pkl") for rank in range(world)]
    r_big, m_big, cols = _sw_inputs()
    op = dict(radius=20, knn=4, window_size=150, overlap=40, min_cells_per_window=60)
    whole = same_amd.sliding_window_incumbent(r_big, m_big, commonCT=cols, optim_params=dict(op))
    if world == 2:
        assert all(p.equals(whole) for p in parts)
    else:
        assert all(len(p) > 50 and "__plan_pos" in p.columns for p in parts)
        merged = pd.concat(parts, ignore_index=True).sort_values("__plan_pos", kind="stable")
        merged = merged.drop(columns=["__plan_pos"]).reset_index(drop=True)
        assert merged.equals(whole)
    _assert_incumbent_equals_golden(whole, load_golden("run_same_mock"), "sw", with_ref_idx=False)
    _r, m_two, _c = _sw_inputs_crowded()
    op2 = dict(radius=25, knn=6, window_size=100, overlap=4, min_cells_per_window=20)
    crowded = same_amd.sliding_window_incumbent(r_big, m_two, commonCT=cols, optim_params=dict(op2))
    want = merge_window_matches_unique_ref([crowded])
    print('window disagreements settled by the merge:', len(crowded) - len(want))
    assert 300 < len(want) <= len(crowded) - 5                                # the overlaps do disagree in this job
    one = same_amd.sliding_window_incumbent(r_big, m_two, commonCT=cols, optim_params=dict(op2), merge=True)
    assert list(one.columns) == list(want.columns) and one.equals(want)       # one process: the merge without the pre-merge table
    mparts = [pd.read_pickle(tmp_path / f"merged{rank}.pkl") for rank in range(world)]
    assert all(0 < len(p) < len(want) and "__plan_pos" not in p.columns for p in mparts)
    assert all(np.all(np.diff(p["Aligned_Cell_Num_Old"].to_numpy()) > 0) for p in mparts)
    assert join_merged_parts(mparts).equals(want)
    if world == 2:     # dist.sharded_merged_window_matches: the reference's own loop per rank, merged the same way
        sparts = [pd.read_pickle(tmp_path / f"solved{rank}.pkl") for rank in range(world)]
        stand_in = lambda prep, _o: (same_amd.incumbent.incumbent_of_prepared(prep, cols, False)[0], {})
        solver_table = same_amd.sliding_window_matching(r_big, m_two, commonCT=cols, optim_params=dict(op2), _solve=stand_in)
        assert join_merged_parts(sparts).equals(merge_window_matches_unique_ref([solver_table]))


def test_metacell_flow_equals_reference(gp, tmp_path, monkeypatch):
    """collapse both sections -> run_same on a MetaCell object -> unpack with per-match assignments -> windows over MetaCell
    objects: the reference's own pipeline, run there through the solver double, reproduced array for array."""
    import run_same_record as rec
    import same_amd
    from same_amd import synth

    monkeypatch.chdir(tmp_path)
    g = load_golden("run_same_mock")
    cells = synth.make_cells(900, 3, seed=61)
    r_c = synth.to_frame(cells)
    a_c = synth.to_frame(synth.make_jittered(cells, seed=62))
    a_c["Cell_Num_Old"] = np.arange(len(a_c)) * 2 + 7
    mc_a = same_amd.greedy_triangle_collapse(a_c, max_metacell_size=4, r_max=40, min_angle_deg=10, return_object=True, verbose=False)
    mc_r = same_amd.greedy_triangle_collapse(r_c, max_metacell_size=3, r_max=40, min_angle_deg=10, return_object=True, verbose=False)
    # the MetaCell container's own helpers, against what the reference object returned
    import json
    probe = np.vstack([mc_a.original_delaunay[:5], [[7, 9, 123456789]], mc_a.original_delaunay[5:8]])
    assert np.array_equal(np.asarray(mc_a.original_delaunay), g["mc_helpers/original_delaunay"])
    members = [json.dumps([int(v) for v in mc_a.metacell_members(k)]) for k in (0, 5, len(mc_a.metacell_df) - 1)]
    assert members == list(g["mc_helpers/members_0_5_last"])
    assert np.array_equal(mc_a.original_delaunay_to_row_indices(), g["mc_helpers/rows"])
    assert np.array_equal(mc_a.original_delaunay_to_pos(probe), g["mc_helpers/rows_probe_drop"])
    assert np.array_equal(mc_a.original_delaunay_to_xy(), g["mc_helpers/xy"])
    assert np.array_equal(mc_a.original_delaunay_to_xy(probe, on_missing="drop"), g["mc_helpers/xy_probe"])
    assert np.array_equal(mc_a.metacell_delaunay_to_xy(), g["mc_helpers/mc_xy"])
    assert json.dumps(mc_a.to_summary_dict(), sort_keys=True, default=str) == str(g["mc_helpers/summary"][0])
    with pytest.raises(KeyError):
        mc_a.original_delaunay_to_row_indices(probe, on_missing="error")
    mop = dict(radius=30, knn=4)
    mgp = dict(init_method="greedy", lazy_allowed_flip_fraction=0.0, lazy_max_cuts_per_incumbent=40)
    out_df, var_out = same_amd.run_same(mc_r.metacell_df, mc_a, synth.type_columns(3), outprefix=str(tmp_path / "mc"),
                                        optim_params=dict(mop), gurobi_params=dict(mgp))
    rec.assert_same_record(rec.record_run(out_df.drop(columns=["members"], errors="ignore"), var_out, gp.Model.last), g,
                           prefix="metacell_flow/")
    indiv = same_amd.unpack_metacell_matches(out_df, mc_a.metacell_df, mc_r.metacell_df, aligned_df=a_c, ref_df=r_c, strategy="nearest",
                                             aligned_original_idx_col="Cell_Num_Old", ref_original_idx_col="Cell_Num_Old")
    assert np.array_equal(indiv[["Aligned_cell_id", "Ref_cell_id"]].to_numpy(dtype=np.int64), g["metacell_flow_unpacked"])
    res = same_amd.sliding_window_matching(mc_r, mc_a, commonCT=synth.type_columns(3),
                                           optim_params=dict(mop, window_size=200, overlap=50, min_cells_per_window=20),
                                               gurobi_params=dict(mgp))
    rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("res", res).items()}, g, prefix="sw_metacell/res_")
    # the solver-free product function on the same MetaCell objects (caller's triangulation -> its general route): the reference's table
    inc = same_amd.sliding_window_incumbent(mc_r, mc_a, commonCT=synth.type_columns(3), window_local_indices=True,
                                            optim_params=dict(mop, window_size=200, overlap=50, min_cells_per_window=20),
                                                gurobi_params=dict(mgp))
    want_cols = [str(c) for c in g["sw_metacell/res_columns"]]
    assert [c for c in inc.columns] == [c for c in want_cols if c in inc.columns] and set(want_cols) - set(inc.columns) <= {"members"}
    for c in inc.columns:
        if c not in ("filtered_violation", "run_time"):
            want, got = g[f"sw_metacell/res__{c}"], inc[c].to_numpy()
            assert np.array_equal(got.astype(want.dtype) if want.dtype.kind in "fiub" else got.astype(str), want), c


def test_shipped_example_flow_equals_reference(gp, tmp_path, monkeypatch):
    """The reference's own CPU-runnable case (BASELINE configs[0]): examples/synthetic driven as its run_same.sh drives it --
    size-1 metacells, MetaCell objects into sliding_window_matching, commonCT inferred from cell_type, paper parameters."""
    import pandas as pd
    import run_same_record as rec
    import same_amd

    monkeypatch.chdir(tmp_path)
    g = load_golden("run_same_mock")

    def frame(prefix):
        cols = [str(c) for c in g[f"example/{prefix}_columns"]]
        df = pd.DataFrame({c: g[f"example/{prefix}__{c}"] for c in cols})
        df["cell_type"] = df["cell_type"].astype(object)
        df["quadrant"] = df["quadrant"].astype(object)
        return df

    ex_ref, ex_query = frame("ref"), frame("query")
    mck = dict(cell_type_col="cell_type", original_idx_col="cell_idx", x_col="X", y_col="Y", max_metacell_size=1, r_max=5, min_angle_deg=5,
               use_alpha_shape=False, alpha=None, return_object=True, verbose=False)
    mc_a = same_amd.greedy_triangle_collapse(ex_query, **mck)
    mc_r = same_amd.greedy_triangle_collapse(ex_ref, **mck)
    ex_gp = same_amd.init_gurobi_params()
    ex_gp.update(mip_gap=0.025, lazy_allowed_flip_fraction=0.0, time_limit=7200, mip_focus=2, init_method="greedy")
    ex_op = same_amd.init_optim_params()
    ex_op.update({"window_size": 100, "overlap": 0, "min_cells_per_window": 30, "max_matches": 2, "radius": 5, "knn": 8,
                  "no_match_penalty": 10000, "dist_ct_coeff": 1, "min_angle_deg": 5, "penalty_coeff": 100, "delaunay_penalty": 10,
                  "cell_id_col": "metacell_id", "ref_metacell_match_multiplier": 1, "ignore_same_type_triangles": False,
                  "lazy_constraints": True})
    res = same_amd.sliding_window_matching(mc_r, mc_a, outprefix=str(tmp_path / "example"), optim_params=ex_op, gurobi_params=ex_gp,
                                           ignore_precomputed_triangulation=False)
    got = {k[4:]: v for k, v in rec.record_frame("res", res).items()}
    want_keys = {k: v for k, v in g.items() if k.startswith("example/res_")}
    rec.assert_same_record(got, want_keys, prefix="example/res_")
    model_keys = rec.record_model(gp.Model.last)
    rec.assert_same_record(model_keys, {f"m/{k}": g[f"example/{k}"] for k in model_keys}, prefix="m/")
    assert len(res) == 351 and len(gp.Model.last.lazy) == 178      # 351 matched under the greedy start, 178 flipped triangles


def test_window_tiler_equals_reference_loop():
    """Tiling, right/down merges of under-populated windows, window ids, central trimming, CSV resume: ten seeded
    configurations, each compared call by call with what the reference's own sliding_window_matching loop did when its
    run_same was replaced by a recorder (tests/golden/window_tiler.npz)."""
    import os
    import tempfile
    import pandas as pd
    import same_amd
    from run_same_record import tiler_inputs

    g = load_golden("window_tiler")
    calls = []

    def recorder(aligned_df, ref_df, commonCT, optim_params, gurobi_params, outprefix, aligned_delaunay, aligned_delaunay_vertex_col,
                 ignore_precomputed_triangulation):
        calls.append((os.path.basename(outprefix) if outprefix else "", aligned_df["Cell_Num_Old"].to_numpy().copy(),
                      ref_df["Cell_Num_Old"].to_numpy().copy()))
        return pd.DataFrame({"X": aligned_df["X"].to_numpy(), "Y": aligned_df["Y"].to_numpy(),
                             "Aligned_Cell_Num_Old": aligned_df["Cell_Num_Old"].to_numpy()}), {}

    def check(tag, res):
        assert [c[0] for c in calls] == list(g[f"{tag}/prefix"]), tag
        a_off, r_off = g[f"{tag}/a_off"], g[f"{tag}/r_off"]
        for q, c in enumerate(calls):
            assert np.array_equal(c[1], g[f"{tag}/a_ids"][a_off[q]:a_off[q + 1]]), (tag, q)
            assert np.array_equal(c[2], g[f"{tag}/r_ids"][r_off[q]:r_off[q + 1]]), (tag, q)
        got = res[["Aligned_Cell_Num_Old", "window_id"]].to_numpy(dtype=np.int64) if len(res) else np.zeros((0, 2), np.int64)
        assert np.array_equal(got, g[f"{tag}/res"]), tag
        calls.clear()

    with tempfile.TemporaryDirectory() as work:
        for q, cfg in enumerate(g["cfgs"]):
            r_df, m_df = tiler_inputs(cfg)
            op = dict(window_size=int(cfg[4]), overlap=int(cfg[5]), min_cells_per_window=int(cfg[6]))
            res = same_amd.sliding_window_matching(r_df.copy(), m_df.copy(), commonCT=["a"], optim_params=dict(op), _run_window=recorder)
            check(f"c{q}/plain", res)
            pre = os.path.join(work, f"c{q}")
            res = same_amd.sliding_window_matching(r_df.copy(), m_df.copy(), commonCT=["a"], outprefix=pre, optim_params=dict(op),
                                                   _run_window=recorder)
            check(f"c{q}/out", res)
            csv = os.path.join(pre, "matchedDF.csv")
            if f"c{q}/kept_ids" in g:
                full = pd.read_csv(csv)
                full[full["window_id"].isin(g[f"c{q}/kept_ids"].tolist())].to_csv(csv, index=False)
                res = same_amd.sliding_window_matching(r_df.copy(), m_df.copy(), commonCT=["a"], outprefix=pre, optim_params=dict(op),
                                                       _run_window=recorder)
                check(f"c{q}/resume", res)
            else:
                assert not os.path.exists(csv)


def test_device_window_rows_equal_the_reference_loops_frames():
    """a13 on the device against the REFERENCE's own loop: for the ten seeded layouts of tests/golden/window_tiler.npz (merges of
    under-populated windows included) the rows same_window_stage finds for every window of the plan -- sections binned on the window
    grid -- are the cells of the frames the reference's sliding_window_matching handed its run_same, call by call, in order."""
    from run_same_record import tiler_inputs

    from same_amd import windows as W

    g = load_golden("window_tiler")
    st = W.DeviceWindow()
    windows = 0
    for q, cfg in enumerate(g["cfgs"]):
        r_df, m_df = tiler_inputs(cfg)
        ws, ov, mc = int(cfg[4]), int(cfg[5]), int(cfg[6])
        rxy, mxy = r_df[["X", "Y"]].to_numpy(), m_df[["X", "Y"]].to_numpy()
        plan = W.window_plan(rxy, mxy, ws, ov, mc)
        xs, ys, _ = W.window_grid(rxy, mxy, ws, ov)
        grid = W.window_cell_grid((xs, ys), ws, ov)
        dref = W.DeviceSection(W.Section(rxy, np.ones((len(rxy), 1)), None, None), "float64").bin(*grid)
        dmov = W.DeviceSection(W.Section(mxy, np.ones((len(mxy), 1)), None, None), "float64").bin(*grid)
        a_off, r_off = g[f"c{q}/plain/a_off"], g[f"c{q}/plain/r_off"]
        assert len(plan) == len(a_off) - 1, q
        r_ids, m_ids = r_df["Cell_Num_Old"].to_numpy(), m_df["Cell_Num_Old"].to_numpy()
        for c, w in enumerate(plan):
            n_m, n_r, _kept, _pairs = st.stage(dmov, dref, w["box"], 1.0, 1, 1.0)
            assert np.array_equal(m_ids[st.fetch(W._W_ROWS_M)], g[f"c{q}/plain/a_ids"][a_off[c]:a_off[c + 1]]), (q, c)
            assert np.array_equal(r_ids[st.fetch(W._W_ROWS_R)], g[f"c{q}/plain/r_ids"][r_off[c]:r_off[c + 1]]), (q, c)
            assert (n_m, n_r) == (w["n_mov"], w["n_ref"])
            windows += 1
        dref.close()
        dmov.close()
    st.close()
    assert windows > 60


@pytest.mark.parametrize("ms", [1, 3])
def test_real_tongue_flow_equals_reference(gp, ms, tmp_path, monkeypatch):
    """Real data (examples/tongue: 4 671 protein cells with UUID string ids against 3 608 RNA cells with 64-bit integer ids,
    proportions with exact zeros) driven as the example's run_same.sh drives it, with MS=1 and with MS=3 metacells + unpacking:
    match table, last model and unpacked cell matches equal the reference's run through the same solver double."""
    import pandas as pd
    import run_same_record as rec
    import same_amd

    monkeypatch.chdir(tmp_path)
    g = load_golden("real_tongue")
    types = ["Endothelial cells", "Epithelial cells", "Fibroblasts", "Lymphoid cells", "Myeloid cells"]

    def frame(prefix):
        cols = [str(c) for c in g[f"{prefix}_columns"]]
        df = pd.DataFrame({c: g[f"{prefix}__{c}"] for c in cols})
        for c in cols:
            if df[c].dtype.kind in "US":
                df[c] = df[c].astype(object)
        return df

    a_df, r_df = frame("prot"), frame("mer")
    assert r_df["Cell_Num"].dtype == np.int64 and int(r_df["Cell_Num"].max()) > 2 ** 53 and a_df["Cell_Num"].dtype == object
    for df in (a_df, r_df):
        df["X"] = df["transformed_x"]
        df["Y"] = df["transformed_y"]
        df[types] = df[types] * 100
        df["cell_type"] = df[types].idxmax(axis=1)
    mck = dict(cell_type_col="cell_type", original_idx_col="Cell_Num", x_col="X", y_col="Y", max_metacell_size=ms, r_max=300,
               min_angle_deg=15, use_alpha_shape=False, return_object=True, verbose=False)
    mc_a = same_amd.greedy_triangle_collapse(a_df, **mck)
    mc_r = same_amd.greedy_triangle_collapse(r_df, **mck)
    assert [len(mc_a.metacell_df), len(mc_r.metacell_df)] == g[f"ms{ms}/n_metacells"].tolist()
    gpar = same_amd.init_gurobi_params()
    gpar.update(mip_gap=0.05, lazy_allowed_flip_fraction=0.05, init_method="greedy")
    op = same_amd.init_optim_params()
    op.update({"window_size": 4000, "overlap": 300, "min_cells_per_window": 30, "max_matches": 1, "radius": 300, "knn": 8,
               "no_match_penalty": 10000, "penalty_coeff": 100, "dist_ct_coeff": 1, "delaunay_penalty": 10,
               "cell_id_col": "metacell_id", "ref_metacell_match_multiplier": ms, "lazy_constraints": True, "min_angle_deg": 15})
    res = same_amd.sliding_window_matching(mc_r, mc_a, commonCT=types, outprefix=str(tmp_path / f"ms{ms}"), optim_params=op,
                                           gurobi_params=gpar, ignore_precomputed_triangulation=False)
    rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("res", res).items()}, g, prefix=f"ms{ms}/res_")
    model = rec.record_model(gp.Model.last)
    rec.assert_same_record(model, {f"m/{k}": g[f"ms{ms}/{k}"] for k in model}, prefix="m/")
    indiv = same_amd.unpack_metacell_matches(res, mc_a.metacell_df, mc_r.metacell_df, aligned_df=a_df, ref_df=r_df, strategy="nearest",
                                             aligned_original_idx_col="Cell_Num", ref_original_idx_col="Cell_Num")
    rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("unp", indiv[["Aligned_cell_id", "Ref_cell_id"]]).items()}, g,
                           prefix=f"ms{ms}/unp_")


@pytest.mark.parametrize("ms", [1, 3])
def test_real_heart_flow_equals_reference(gp, ms, tmp_path, monkeypatch):
    """Real data with built-in degeneracy (examples/heart: ISS spots on a regular 242.5-unit lattice -> co-circular Delaunay
    input and equal distances; percentages from small counts -> exact zeros and exact cost ties), 13 windows with merges."""
    import pandas as pd
    import run_same_record as rec
    import same_amd

    monkeypatch.chdir(tmp_path)
    g = load_golden("real_heart")
    types = ["Smooth muscle cells", "Fibroblast", "Atrial cardiomyocytes", "Cardiomyocytes", "Endothelium", "Epicardium",
             "Schwan progenitors", "Ventricular cardiomyocytes"]

    def frame(prefix):
        cols = [str(c) for c in g[f"{prefix}_columns"]]
        return pd.DataFrame({c: g[f"{prefix}__{c}"] for c in cols})

    a_df, r_df = frame("query"), frame("ref")
    for df in (a_df, r_df):
        df.rename(columns={f"{ct}_percentage": ct for ct in types}, inplace=True)
        df["X"] = df["spot_x"] + 75
        df["Y"] = df["spot_y"] + 75
        df["cell_type"] = df[types].idxmax(axis=1)
    mck = dict(cell_type_col="cell_type", original_idx_col="Cell_Num", x_col="X", y_col="Y", max_metacell_size=ms, r_max=500,
               min_angle_deg=15, use_alpha_shape=False, return_object=True, verbose=False)
    mc_a = same_amd.greedy_triangle_collapse(a_df, **mck)
    mc_r = same_amd.greedy_triangle_collapse(r_df, **mck)
    assert [len(mc_a.metacell_df), len(mc_r.metacell_df)] == g[f"ms{ms}/n_metacells"].tolist()
    gpar = same_amd.init_gurobi_params()
    gpar.update(mip_gap=0.05, lazy_allowed_flip_fraction=0.05, time_limit=7200, init_method="greedy")
    op = same_amd.init_optim_params()
    op.update({"window_size": 4000, "overlap": 100, "min_cells_per_window": 30, "max_matches": 1, "radius": 500, "knn": 8,
               "no_match_penalty": 10000, "penalty_coeff": 100, "dist_ct_coeff": 1, "delaunay_penalty": 10,
               "cell_id_col": "metacell_id", "ref_metacell_match_multiplier": ms, "ignore_same_type_triangles": True,
               "lazy_constraints": True, "min_angle_deg": 15})
    res = same_amd.sliding_window_matching(mc_r, mc_a, commonCT=types, outprefix=str(tmp_path / f"ms{ms}"), optim_params=op,
                                           gurobi_params=gpar, ignore_precomputed_triangulation=False)
    rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("res", res).items()}, g, prefix=f"ms{ms}/res_")
    model = rec.record_model(gp.Model.last)
    rec.assert_same_record(model, {f"m/{k}": g[f"ms{ms}/{k}"] for k in model}, prefix="m/")
    indiv = same_amd.unpack_metacell_matches(res, mc_a.metacell_df, mc_r.metacell_df, aligned_df=a_df, ref_df=r_df, strategy="nearest",
                                             aligned_original_idx_col="Cell_Num", ref_original_idx_col="Cell_Num")
    rec.assert_same_record({k[4:]: v for k, v in rec.record_frame("unp", indiv[["Aligned_cell_id", "Ref_cell_id"]]).items()}, g,
                           prefix=f"ms{ms}/unp_")


def test_run_same_parameter_sweep_equals_reference(gp, tmp_path, monkeypatch):
    """Sixteen seeded parameter combinations (triangle flags x KNN priority x lazy/eager x start methods x max_matches x
    penalties x cut limits): every recorded array of the reference's run_same is reproduced (tests/golden/run_same_sweep.npz)."""
    import run_same_record as rec
    import same_amd
    from same_amd import synth

    monkeypatch.chdir(tmp_path)
    g = load_golden("run_same_sweep")
    for q in range(int(g["n_cfg"][0])):
        n, T, op, gpar = rec.random_run_same_config(q)
        cells = synth.make_cells(n, T, seed=700 + q)
        r_df = synth.to_frame(cells)
        a_df = synth.to_frame(synth.make_jittered(cells, seed=800 + q))
        out_df, var_out = same_amd.run_same(r_df.copy(), a_df.copy(), synth.type_columns(T), outprefix=str(tmp_path / f"c{q}"),
                                            optim_params=same_amd.init_optim_params(**op),
                                                gurobi_params=same_amd.init_gurobi_params(**gpar))
        rec.assert_same_record(rec.record_run(out_df, var_out, gp.Model.last), g, prefix=f"c{q}/")


def _sharded_windows_worker(rank, world, out_dir):
    import os
    import pickle
    import sys
    import time

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import fake_gurobipy
    import numpy as np

    fake_gurobipy.install()
    os.chdir(out_dir)
    import run_same_record as rec
    from same_amd import synth
    from same_amd.dist import sharded_sliding_window_matching

    def exchange(obj):          # a launcher-agnostic host channel: one pickle per rank in a shared directory
        tmp = os.path.join(out_dir, f"part{rank}.tmp")
        with open(tmp, "wb") as f:
            pickle.dump(obj, f)
        os.replace(tmp, os.path.join(out_dir, f"part{rank}.pkl"))
        parts = []
        for r in range(world):
            path = os.path.join(out_dir, f"part{r}.pkl")
            deadline = time.time() + 600
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise TimeoutError(path)
                time.sleep(0.02)
            with open(path, "rb") as f:
                parts.append(pickle.load(f))
        return parts

    cells = synth.make_cells(1500, 3, seed=51)
    r_big = synth.to_frame(cells)
    m_big = synth.to_frame(synth.make_jittered(cells, seed=52))
    m_big = m_big[~((m_big["X"] < 120) & (m_big["Y"] < 170) & (np.arange(len(m_big)) % 4 != 0))].reset_index(drop=True)
    if world == 2:   # the default channel: same_amd.rendezvous.HostGroup (loopback TCP) built from RANK / WORLD_SIZE / SAME_RDV_DIR
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), SAME_RDV_DIR=os.path.join(out_dir, "rdv"))
        channel = {}
    else:            # a launcher's own channel
        channel = dict(exchange=exchange, rank=rank, world=world)
    res = sharded_sliding_window_matching(r_big, m_big, commonCT=synth.type_columns(3), **channel,
                                          outprefix=os.path.join(out_dir, "sw"),
                                          optim_params=dict(radius=20, knn=4, window_size=150, overlap=40, min_cells_per_window=60),
                                          gurobi_params=dict(init_method="greedy", lazy_allowed_flip_fraction=0.0))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **rec.record_frame("res", res))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sliding_windows_equal_reference(tmp_path, world):
    """BASELINE cfg 5 form: the window plan dealt round-robin to `world` ranks (one process each, match tables exchanged over a
    host channel); every rank ends up with exactly the frame the REFERENCE's single-process sliding_window_matching produced.
    World 2 goes through the package's own plain-Python host group, world 3 through a caller-supplied file exchange."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sharded_windows_worker, args=(rank, world, str(tmp_path))) for rank in range(world)]
    [p.start() for p in procs]
    [p.join(900) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    import run_same_record as rec
    g = load_golden("run_same_mock")
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        rec.assert_same_record({k[4:]: got[k] for k in got.files}, g, prefix="sw/res_")
        assert (tmp_path / "sw" / f"rank{rank}" / "matchedDF.csv").exists()


def test_window_rows_do_not_depend_on_the_section_grid():
    """same_window_stage builds a window's row lists from the cells of the section's grid the box covers.  Whatever the grid -- the
    default one, the window grid (every box a union of cells: no test at all), a grid the boxes cut through (exact box test on the
    candidates), cells so small that a box covers more than 64 of them (one mask over the section) -- the rows, pairs and costs
    are the same, and they are np.flatnonzero of the reference's four comparisons (src/same.py:293-295)."""
    from same_amd import synth
    from same_amd import windows as W

    T = 4
    ref = synth.make_cells(60_000, T, seed=50)
    mov = synth.make_jittered(ref, seed=51)
    mov["xy"][7] = (np.nan, 3.0)                       # a row no box holds
    ref["xy"][11] = (np.inf, 3.0)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    cols = synth.type_columns(T)
    rxy, mxy = r_df[["X", "Y"]].to_numpy(), m_df[["X", "Y"]].to_numpy()
    ok_r, ok_m = np.isfinite(rxy).all(axis=1), np.isfinite(mxy).all(axis=1)
    xs, ys, _ = W.window_grid(rxy[ok_r], mxy[ok_m], 600, 150)
    plan = W.window_plan(rxy[ok_r], mxy[ok_m], 600, 150, 10)
    boxes = [w["box"] for w in plan[::5]] + [(101.5, 640.25, 333.0, 1200.0), (-50.0, 90.0, -50.0, 4000.0), (0.0, 1e9, 0.0, 1e9),
                                             (5.0, 5.0, 0.0, 9.0)]
    ref_sec, mov_sec = W.Section.from_frame(r_df, cols), W.Section.from_frame(m_df, cols)
    x0, y0, cell = W.window_cell_grid((xs, ys), 600, 150)
    assert cell == 150.0
    grids = {"default": None, "window grid": (x0, y0, cell), "offset grid": (x0 + 37.25, y0 - 11.5, 211.0), "tiny cells": (x0, y0, 25.0),
             "quarter windows of an odd size": W.window_cell_grid((xs, ys), 601, 150)}
    got = {}
    for name, g in grids.items():
        dref, dmov = W.DeviceSection(ref_sec, "float32"), W.DeviceSection(mov_sec, "float32")
        if g is not None:
            dref.bin(*g)
            dmov.bin(*g)
        st = W.DeviceWindow()
        out = []
        for box in boxes:
            counts = st.stage(dmov, dref, box, 25, 8, 1.0)
            out.append((counts, st.fetch(W._W_ROWS_M), st.fetch(W._W_ROWS_R), st.fetch(W._W_PAIRS), st.fetch(W._W_COSTS),
                        st.fetch(W._W_ALIGNED_ROWS)))
        got[name] = out
        st.close()
        dref.close()
        dmov.close()
    for q, box in enumerate(boxes):
        x0_, x1_, y0_, y1_ = box
        want_m = np.flatnonzero((mxy[:, 0] >= x0_) & (mxy[:, 0] < x1_) & (mxy[:, 1] >= y0_) & (mxy[:, 1] < y1_))
        want_r = np.flatnonzero((rxy[:, 0] >= x0_) & (rxy[:, 0] < x1_) & (rxy[:, 1] >= y0_) & (rxy[:, 1] < y1_))
        base = got["default"][q]
        assert np.array_equal(base[1], want_m) and np.array_equal(base[2], want_r) and base[0][:2] == (len(want_m), len(want_r)), box
        for name in grids:
            for a, b in zip(got[name][q], base):
                assert np.array_equal(a, b), (name, box)
    assert sum(len(o[3]) for o in got["default"]) > 10_000


def test_window_rows_when_points_sit_on_cell_and_box_edges():
    """A row belongs to the grid cell whose edges -- the very doubles x0 + c * cell -- bracket it under the reference's comparisons
    (>= lower, < upper, src/same.py:293-295), so a box on cell edges needs no test.  Lattice points ON the edges, with an edge that is
    not a binary fraction (0.1 * c), origins off the lattice, boxes on and off the edges, degenerate boxes: always np.flatnonzero."""
    from same_amd import windows as W
    from same_amd._lib import SameHipError

    g = np.arange(60, dtype=np.float64)
    for scale in (1.0, 0.1, 7.0 / 3.0):
        pts = np.array([(x * scale, y * scale) for y in g for x in g])                 # 3 600 lattice points, many exactly on cell edges
        sec = W.Section(pts, np.ones((len(pts), 2)), None, None)
        edges = [0.0, 3 * scale, 10 * scale, 20 * scale, 30 * scale, 0.30000000000000004, 59 * scale, 60 * scale, 12.5 * scale, -4.0, 1e9]
        y_spans = ((0.0, 60 * scale), (10 * scale, 30 * scale), (12.5 * scale, 12.5 * scale))
        boxes = [(a, b, c, d) for a in edges[:6] for b in edges[3:] for (c, d) in y_spans]
        boxes += [(5.0, 5.0, 0.0, 9.0), (9.0, 5.0, 0.0, 9.0), (float("nan"), 5.0, 0.0, 9.0), (-1e300, 1e300, -1e300, 1e300)]
        st = W.DeviceWindow()
        for origin, cell in (((0.0, 0.0), 10 * scale), ((0.0, 0.0), scale), ((-3.7, 1.3), 4.9 * scale),
                             ((30 * scale, 30 * scale), 10 * scale)):
            dsec = W.DeviceSection(sec, "float64").bin(origin[0], origin[1], cell)
            for box in boxes:
                x0, x1, y0, y1 = box
                want = np.flatnonzero((pts[:, 0] >= x0) & (pts[:, 0] < x1) & (pts[:, 1] >= y0) & (pts[:, 1] < y1))
                n_m, n_r, _k, _p = st.stage(dsec, dsec, box, scale, 2, 1.0)
                assert n_m == n_r == len(want), (scale, origin, cell, box)
                case = (scale, origin, cell, box)
                assert np.array_equal(st.fetch(W._W_ROWS_M), want) and np.array_equal(st.fetch(W._W_ROWS_R), want), case
            dsec.close()
        st.close()
    # a section without a usable row, and grids the library refuses
    empty = W.DeviceSection(W.Section(np.zeros((0, 2)), np.zeros((0, 1)), None, None), "float64")
    nans = W.DeviceSection(W.Section(np.full((5, 2), np.nan), np.ones((5, 1)), None, None), "float64").bin(0.0, 0.0, 1.0)
    st = W.DeviceWindow()
    assert st.stage(empty, nans, (0.0, 1.0, 0.0, 1.0), 1.0, 1, 1.0) == (0, 0, 0, 0)
    assert st.stage(nans, empty, (-1e9, 1e9, -1e9, 1e9), 1.0, 1, 1.0) == (0, 0, 0, 0)
    some = W.DeviceSection(W.Section(np.array([[0.0, 0.0], [1e6, 1e6]]), np.ones((2, 1)), None, None), "float64")
    # the last: 10^18 cells
    for bad in ((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (float("nan"), 0.0, 1.0), (0.0, float("inf"), 1.0), (0.0, 0.0, 1e-3)):
        with pytest.raises(SameHipError):
            some.bin(*bad)
    assert st.stage(some, some, (0.0, 2e6, 0.0, 2e6), 1.0, 1, 1.0)[:2] == (2, 2)       # a refused grid leaves the old one in place
    for h in (st, empty, nans, some):
        h.close()


def test_window_calls_stay_within_their_launch_budget():
    """What a window costs in runtime calls, counted by the library itself (same_ctx_stat), ENTERED THROUGH THE PRODUCT FUNCTIONS.  Every
    kernel of the two window calls takes up to eight windows per launch, the heads of their buffers are zeroed by one launch, the call's
    simplices go up in one copy and the answers are written into the pinned blocks by one launch: a batch of eight windows is 6-7 + 16
    launches (19 with fp64 costs), no fill, one copy and two waits -- per window 2.75-3.25 launches, 0 fills, 0.125 copies, 0.25 waits,
    the figures of README.md / profiles/README.md / include/same_hip.h (round 5 before that: 18 launches, 3 fills, 4 copies; round 4 two
    waits; round 3 ~80 launches, ~24 fills, ~13 copies and 5-6 waits).  The plan here is 16 windows = two whole batches on frames that
    are resident already (the sections' upload and binning are per JOB, not per window), so the budgets sit right above those figures:
    a slip back to a launch, a fill or a copy per window fails.  With the window merge on the device (merge=True) a batch adds the three
    launches of same_window_collect; what the merge asks for once per pass is counted apart.  iter_prepared_windows (what
    sliding_window_matching hands its run_same body) adds the seven arrays it fetches for the solver: pairs, reference rows, costs,
    triangles, signs, weights."""
    import same_amd
    from same_amd import _lib, synth
    from same_amd import windows as W

    T = 8
    ref = synth.make_cells(125_000, T, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    cols = synth.type_columns(T)
    op = dict(radius=25, knn=8, dist_ct_coeff=1.0, min_angle_deg=15, ignore_same_type_triangles=True, no_match_penalty=100.0,
              hip_cost_dtype="float32",
              window_size=1200, overlap=300, min_cells_per_window=10)
    plan = W.window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
    assert len(plan) == 16
    ctx = _lib.default_context()
    with same_amd.resident_frames(r_df, m_df, ctx=ctx) as res_frames:
        call = lambda **kw: same_amd.sliding_window_incumbent(res_frames, res_frames, commonCT=cols, optim_params=dict(op), workers=1,
                                                              ctx=ctx,
                                                              batch=8, return_stats=True, **kw)
        call()                                                           # buffers, helpers, the prune index, the sections
        frames = next(iter(res_frames._frames.values()))
        spent0, before = dict(frames.merge_runtime_calls), ctx.stats()
        res, stats = call()
        after, spent1 = ctx.stats(), frames.merge_runtime_calls
        once = {k: spent1[k] - spent0[k] for k in after}      # per PASS: the accumulator's begin, the rows laid end to end, the columns
        per = {k: (after[k] - before[k] - once[k]) / len(stats) for k in after}
        print("per window (sliding_window_incumbent):", per, "; once per pass:", once)
        assert len(stats) == len(plan) and sum(s["pairs"] for s in stats) > 100_000 and len(res) > 50_000
        # windows go to the library in batches of 8: ONE wait per call for the whole batch (it was one per window and call); the three
        # launches of same_window_collect per batch are the windows' too
        assert per["launches"] <= 3.5 and per["fills"] == 0 and per["copies"] <= 0.3 and per["waits"] <= 0.3, per
        assert once["launches"] <= 4 and once["fills"] <= 2 and once["copies"] <= 2 and once["waits"] <= 3, once
        call(merge=True)                                                 # the merge's work buffers
        spent0, before = dict(frames.merge_runtime_calls), ctx.stats()
        merged, stats_m = call(merge=True)
        after, spent1 = ctx.stats(), frames.merge_runtime_calls
        once = {k: spent1[k] - spent0[k] for k in after}
        per = {k: (after[k] - before[k] - once[k]) / len(stats_m) for k in after}
        print("per window (merge=True):", per, "; the merge, once per pass:", once)
        assert 50_000 < len(merged) <= len(res) and stats_m == stats
        assert per["launches"] <= 3.5 and per["fills"] == 0 and per["copies"] <= 0.3 and per["waits"] <= 0.3, per
        assert once["launches"] <= 80 and once["copies"] <= 8 and once["waits"] <= 6, once
    before = ctx.stats()
    preps = [p for _w, p in same_amd.iter_prepared_windows(r_df, m_df, cols, plan, optim_params=dict(op)) if not isinstance(p, Exception)]
    after = ctx.stats()
    per = {k: (after[k] - before[k]) / len(preps) for k in after}
    print("per window (iter_prepared_windows):", per)
    # (these two build their sections inside the call: a handful of launches and copies per JOB, spread over 16 windows)
    assert len(preps) == len(stats) and per["launches"] <= 6 and per["fills"] <= 1 and per["copies"] <= 9 and per["waits"] <= 8, per
    # ... and through the reference's own signature, with the incumbent standing in for the solver half of run_same
    from same_amd.incumbent import incumbent_of_prepared

    before = ctx.stats()
    stand_in = lambda prep, _o: (incumbent_of_prepared(prep, cols)[0], {})
    out = same_amd.sliding_window_matching(r_df, m_df, commonCT=cols, optim_params=dict(op), _solve=stand_in)
    after = ctx.stats()
    per = {k: (after[k] - before[k]) / len(stats) for k in after}
    print("per window (sliding_window_matching):", per)
    assert out["window_id"].nunique() == len(stats) and len(out) == len(res)
    assert per["launches"] <= 6 and per["fills"] <= 1 and per["copies"] <= 9 and per["waits"] <= 8, per
    for column in ("Aligned_Cell_Num_Old", "Ref_Cell_Num_Old"):
        assert np.array_equal(out[column].to_numpy(), res[column].to_numpy())
