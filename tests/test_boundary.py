"""CPU-only checks of the drop-in boundary: the C ABI surface, the product/oracle separation,
parameter dictionaries, and host-side logic that needs no GPU."""
import ast
import ctypes
import os
import re

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("same_hip.h", "same_hip_diag.h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(same_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    """libsame_hip.so loads without a GPU and exports exactly what include/same_hip.h (the path) and include/same_hip_diag.h (measurement
    hooks, opt-in controls) declare; the diagnostics stay out of the path's header."""
    from same_amd import _lib

    lib = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in same_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "ctypes prototypes and header drifted apart"
    assert lib.same_abi_version() == _lib.ABI_VERSION == 8
    path_header = open(os.path.join(ROOT, "include", "same_hip.h")).read()
    for hook in ("same_ctx_stat", "same_timer_start", "same_dev_alloc_spread", "same_dense_cost_q32_dev", "same_comm_gather_time"):
        assert hook + "(" not in path_header, hook
    assert b"range" in lib.same_strerror(-34) and b"Qhull" in lib.same_strerror(-11)


def test_no_gpu_is_reported_not_hidden():
    """Without a device the product raises; it never falls back to a CPU path."""
    from same_amd import _lib, ops

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert _lib.load().same_ctx_create(0, ctypes.byref(h)) == -19  # SAME_ENODEV
    with pytest.raises(_lib.SameHipError):
        ops.knn_prune(np.zeros((4, 2)), np.zeros((4, 2)), 1.0, 2)
    import same_amd
    df = pd.DataFrame({"X": [0.0, 1.0], "Y": [0.0, 1.0]})
    with pytest.raises(_lib.SameHipError):
        same_amd.find_knn_within_radius(df, df, 5.0, knn=1, verbose=False)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "same_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n.split(".")[0] == "oracle" or "same_oracle" in n for n in names), f"{f} imports the oracle"
            assert "libsame_oracle" not in open(os.path.join(dirpath, f)).read()
    for f in os.listdir(os.path.join(pkg, "csrc")):  # native side: no include of / link against the oracle
        if f.endswith((".hip", ".h")) or f == "Makefile":
            txt = open(os.path.join(pkg, "csrc", f)).read()
            assert "#include \"same_oracle" not in txt and "lsame_oracle" not in txt and "orc_" not in txt
    # torch is plumbing for bench.py / dist tests only: the package itself must import without it
    import subprocess, sys
    code = "import sys; sys.modules['torch']=None; import same_amd; print('ok')"
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True).stdout.strip() == "ok"


def test_param_dicts_match_reference_defaults():
    """Keys and defaults of src/same.py:106-130 and :215-242."""
    import same_amd

    g = same_amd.init_gurobi_params()
    assert g == {"time_limit": 7200, "mip_gap": 0.05, "mip_focus": 2, "cuts": 2, "heuristics": 0.1, "init_method": None,
                 "init_big_m": 1e9, "init_hungarian_max_n": 5000, "lazy_max_cuts": None,
                 "lazy_allowed_flip_fraction": 0.05, "lazy_max_cuts_per_incumbent": 1000}
    o = same_amd.init_optim_params()
    assert o == {"window_size": 1000, "overlap": 250, "min_cells_per_window": 10, "max_matches": 1,
                 "ref_metacell_match_multiplier": None, "radius": 250, "penalty_coeff": 100, "no_match_penalty": 100,
                 "delaunay_penalty": 5, "dist_ct_coeff": 1, "knn": 8, "cell_id_col": "Cell_Num_Old",
                 "hard_spatial_constraints": False, "ignore_same_type_triangles": True, "ignore_knn_if_matched": False,
                 "lazy_constraints": True, "min_angle_deg": 15}
    assert same_amd.init_optim_params(radius=5, knn=3)["radius"] == 5
    assert same_amd.init_gurobi_params(brand_new=1)["brand_new"] == 1  # overrides may add keys, like dict.update


def test_public_signatures_match_reference():
    import inspect
    import same_amd

    assert list(inspect.signature(same_amd.run_same).parameters) == [
        "ref_df", "aligned_df", "commonCT", "outprefix", "aligned_delaunay", "aligned_delaunay_vertex_col", "optim_params",
        "gurobi_params", "ignore_precomputed_triangulation"]
    sw = list(inspect.signature(same_amd.sliding_window_matching).parameters)
    assert sw[:9] == ["ref", "moving", "commonCT", "outprefix", "moving_delaunay", "moving_delaunay_vertex_col", "optim_params",
                      "gurobi_params", "ignore_precomputed_triangulation"]
    assert list(inspect.signature(same_amd.find_knn_within_radius).parameters)[:4] == ["aligned_df", "ref_df", "radius", "knn"]
    p = inspect.signature(same_amd.find_knn_within_radius).parameters
    assert p["radius"].default == 25 and p["knn"].default == 5
    assert list(inspect.signature(same_amd.verify_spatial_preservation).parameters)[:5] == [
        "aligned_df", "ref_df", "matches_df", "triangle_info", "tolerance"]
    f = inspect.signature(same_amd.filter_triangles_by_radius).parameters
    assert list(f)[:8] == ["points", "triangles", "radius", "aligned_df", "ignore_same_type_triangles",
                           "ensure_min_triangle_per_node", "remove_unconstrained_nodes", "min_angle_deg"]
    assert f["min_angle_deg"].default == 15 and f["ensure_min_triangle_per_node"].default is True
    c = inspect.signature(same_amd.compute_mip_start_pairs).parameters
    assert all(c[k].kind is inspect.Parameter.KEYWORD_ONLY for k in c)
    assert c["init_big_m"].default == 1e9 and c["init_hungarian_max_n"].default == 2000


def test_cos_threshold_is_the_reference_angle_rule(oracle):
    """c >= thr  <=>  degrees(arccos(c)) < min_angle_deg, on the double lattice around the threshold."""
    from same_amd.triangles import cos_threshold

    for deg in (0.5, 5, 15, 30, 45, 60, 89.999, 90, 120, 179):
        en, thr = cos_threshold(deg)
        assert (en, thr) == oracle.cos_threshold(deg)
        c = thr
        for _ in range(50):
            assert np.degrees(np.arccos(c)) < deg
            c = np.nextafter(c, 2.0)
        c = np.nextafter(thr, -2.0)
        for _ in range(50):
            assert not (np.degrees(np.arccos(c)) < deg)
            c = np.nextafter(c, -2.0)
    assert cos_threshold(None) == (0, float("inf"))
    assert cos_threshold(0) == (1, float("inf"))      # nothing is < 0 degrees
    assert cos_threshold(181) == (1, float("-inf"))   # everything fails


def test_triangle_remap_matches_reference_semantics():
    from same_amd.triangles import _as_triangle_array, _remap_triangles_by_vertex_ids

    vid = np.array([10, 20, 30, 40, 20])          # duplicate id 20: the dict keeps the LAST row (4)
    tri = np.array([[10, 20, 30], [30, 40, 99], [40, 20, 10]])
    out = _remap_triangles_by_vertex_ids(tri, vid)
    assert out.tolist() == [[0, 4, 2], [3, 4, 0]]  # triangle with missing id 99 dropped
    assert _remap_triangles_by_vertex_ids(np.zeros((0, 3)), vid).shape == (0, 3)
    assert _remap_triangles_by_vertex_ids(pd.DataFrame(tri), vid).tolist() == out.tolist()
    with pytest.raises(ValueError):
        _as_triangle_array(np.zeros((4, 2)))
    assert _as_triangle_array(None) is None


def test_compaction_matches_oracle(oracle):
    from same_amd.knn import compact_pairs, pairs_from_padded

    rng = np.random.default_rng(0)
    idx = rng.integers(-1, 50, size=(40, 6)).astype(np.int32)
    idx[5] = -1  # a row without neighbours
    idx = np.where(np.sort(idx < 0, axis=1), -1, np.take_along_axis(idx, np.argsort(idx < 0, axis=1, kind="stable"), 1))
    a = pd.DataFrame({"X": rng.random(40), "Y": rng.random(40), "tag": np.arange(40)})
    r = pd.DataFrame({"X": rng.random(50), "Y": rng.random(50), "tag": np.arange(50)})
    p = pairs_from_padded(idx)
    assert np.array_equal(p, oracle.pairs_from_padded(idx))
    na, nr, np_ = compact_pairs(a, r, p)
    oa, orr, op = oracle.compact_pairs(a, r, p)
    assert na.equals(oa) and nr.equals(orr) and np.array_equal(np_, op)
    assert 5 not in na["tag"].to_numpy()
    e = compact_pairs(a, r, np.empty((0, 2), int))
    assert len(e[0]) == 0 and len(e[1]) == 0 and len(e[2]) == 0


def test_row_block_partition():
    from same_amd.dist import row_block

    for n, w in ((100, 8), (7, 8), (0, 4), (25000, 3), (8, 8)):
        blocks = [row_block(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(b[1] == blocks[i + 1][0] for i, b in enumerate(blocks[:-1]))
        assert len({b[2] for b in blocks}) == 1 and all(b[1] - b[0] <= b[2] for b in blocks)


def test_plain_c_program_links_against_the_abi(tmp_path):
    """examples/abi_demo.c (C11, no C++/Python) compiles against include/same_hip.h and links libsame_hip.so;
    without a GPU it exits 2 with the ENODEV message (it is run for real by the GPU suite)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = tmp_path / "abi_demo"
    lib_dir = os.path.join(ROOT, "same_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "examples", "abi_demo.c"),
                           "-o", str(exe), f"-L{lib_dir}", "-lsame_hip", f"-Wl,-rpath,{lib_dir}", "-lm"])
    from same_amd import _lib
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if _lib.device_count() == 0:
        assert r.returncode == 2 and "no usable GPU" in r.stderr
    else:
        assert r.returncode == 0 and "flipped" in r.stdout


def test_knn_argument_checks_mirror_ckdtree():
    """Non-finite coordinates raise what cKDTree raises for the reference; a NaN radius or knn=0 selects nothing.  All of it is
    decided on the host before any device call, so it runs without a GPU."""
    import pandas as pd
    from same_amd.knn import find_knn_within_radius

    a = pd.DataFrame({"X": [0.0, 1.0], "Y": [0.0, 1.0]})
    r = pd.DataFrame({"X": [0.0, 2.0], "Y": [0.5, np.nan]})
    with pytest.raises(ValueError, match="data must be finite"):
        find_knn_within_radius(a, r, 5, 2, verbose=False)
    with pytest.raises(ValueError, match="'x' must be finite"):
        find_knn_within_radius(r, a, 5, 2, verbose=False)
    for radius, knn in ((float("nan"), 3), (5.0, 0)):
        na, nr, pairs = find_knn_within_radius(a, a, radius, knn, verbose=False)
        assert len(na) == 0 and len(nr) == 0 and len(pairs) == 0


def test_no_name_is_read_without_being_bound():
    """A forgotten import in a GPU-only code path cannot be caught by running it on this CPU box; catch it on the syntax tree."""
    import glob
    import subprocess
    import sys

    files = sorted(glob.glob(os.path.join(ROOT, "same_amd", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "undefined_names.py")] + files, capture_output=True, text=True)
    assert p.returncode == 0, p.stdout


def test_stage_markers_are_inert_unless_asked_for(monkeypatch):
    """same_amd._trace: no bookkeeping by default; with tracing on, wall time per stage accumulates (and rocTX ranges are emitted
    when the library is there -- it loads without a GPU)."""
    from same_amd import _trace

    _trace.reset()
    _trace.enable(False)
    with _trace.stage("a"):
        pass
    assert _trace.report() == {}
    _trace.enable(True)
    try:
        for _ in range(3):
            with _trace.stage("a"):
                pass
        with pytest.raises(RuntimeError):
            with _trace.stage("b"):
                raise RuntimeError("x")
        rep = _trace.report()
        assert rep["a"][0] == 3 and rep["b"][0] == 1 and rep["a"][1] >= 0.0
    finally:
        _trace.enable(False)
        _trace.reset()
