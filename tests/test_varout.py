"""var_out on disk: the pickle-free JSON + npz pair round-trips the nested dict run_same returns, and a
reference-written `var_out.npy` is only ever read through the numpy-only unpickler."""
import os
import pickle

import numpy as np
import pandas as pd
import pytest

from conftest import frames_from_golden, load_golden


def assert_same(a, b, path=""):
    """equal values, equal container kinds, equal key order"""
    if isinstance(a, dict):
        assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), path
        for k in a:
            assert_same(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (set, frozenset)):
        assert isinstance(b, set) and set(a) == b, path
    elif isinstance(a, np.ndarray):
        assert isinstance(b, np.ndarray) and a.shape == b.shape and np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), path
    elif isinstance(a, (list, tuple)):
        assert type(b) is (tuple if isinstance(a, tuple) else list) and len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            assert_same(x, y, f"{path}[{i}]")
    elif a is None or isinstance(a, str):
        assert a == b and type(b) is type(a), path
    else:
        assert (a == b) or (a != a and b != b), (path, a, b)
        assert isinstance(b, (bool, int, float)), (path, type(b))          # numbers come back as Python numbers


def _var_out_like(case):
    """A var_out shaped like run_same's (src/same.py:1404-1453), built on the CPU from the reference fixtures."""
    from oracle import same_oracle as orc
    from same_amd import triangles

    g = load_golden(case)
    a_df, r_df, _ = frames_from_golden(g)
    na = a_df.iloc[g["kept_aligned"]].reset_index(drop=True)
    nr = r_df.iloc[g["kept_ref"]].reset_index(drop=True)
    tris, ch = g["tri_plain"], g["greedy_chosen"]
    smap = triangles.build_simplex_map(len(na), tris)
    info = triangles.precompute_triangle_info(na, tris, smap)
    viol = orc.verify_spatial_preservation(na, nr, pd.DataFrame({"aligned_idx": ch[:, 0], "ref_idx": ch[:, 1]}), info)
    n = len(tris)
    return {
        "x": list(g["x_vals"]), "no_match_vars": [0.0] * len(na), "penalty_vars": [np.float64(1.5)] * len(nr), "area_penalty_vars": [],
        "violations": viol,
        "violation_penalty_comparison": {"points_both": [], "points_only_violations": list(viol["points_with_violations"]),
                                         "points_only_penalties": []},
        "triangle_data": {"triangles": tris, "triangle_info": info, "aligned_simplex_map": smap,
                          "areas_before": {t: g["area_before"][t] for t in range(n)},
                          "areas_after": {t: (None if np.isnan(g["area_after"][t]) else g["area_after"][t]) for t in range(n)},
                          "flipped_triangles": [int(t) for t in g["area_flipped"]],
                          "matched_vertices": {t: [bool(b) for b in g["area_matched3"][t]] for t in range(n)}},
        "lazy_constraints": True, "lazy_cuts_added": 98,
    }


@pytest.mark.parametrize("case", ["cfg2_small", "synthetic_example"])
def test_var_out_round_trip(tmp_path, case):
    from same_amd import varout

    vo = _var_out_like(case)
    varout.save(str(tmp_path), vo)
    assert sorted(os.listdir(tmp_path)) == ["var_out.json", "var_out.npz"]
    back = varout.load(str(tmp_path))
    assert_same(vo, back)
    # the bulk went to the npz, not into the JSON
    assert os.path.getsize(tmp_path / "var_out.json") < 20_000
    # triangle_info keeps its insertion order (verify_spatial_preservation iterates it, src/violationhelper.py:53)
    assert list(back["triangle_data"]["triangle_info"].keys()) == list(vo["triangle_data"]["triangle_info"].keys())
    assert list(back["triangle_data"]["triangle_info"].keys()) != sorted(back["triangle_data"]["triangle_info"].keys())


def test_var_out_codec_edge_cases(tmp_path):
    from same_amd import varout

    vo = {
        "empty": {"d": {}, "l": [], "t": (), "s": set(), "a": np.zeros((0, 3), np.int32)},
        "floats": [float("nan"), float("inf"), -float("inf"), 0.1, -0.0],
        "long_floats": [float("nan")] + [0.5 * i for i in range(40)],
        "ints": list(range(30)), "bools": [True, False] * 10, "mixed": [1, "a", None, 2.5, [1, 2], (3, 4), {"k": {5, 6}}],
        "tuple_keys": {(1, 2): "a", (3, 4): [1.0, 2.0]},
        "int_map_mixed": {1: "x", 2: 3},                                  # not homogeneous -> generic pairs
        "one_record": [{"a": 1, "b": {"c": 2.0}}], "two_records": [{"a": 1, "b": {"c": 2.0}}, {"a": 3, "b": {"c": float("nan")}}],
        "records_with_vectors": [{"v": (1, 2, 3), "w": [0.5, 1.5], "n": np.int32(7)}, {"v": (4, 5, 6), "w": [2.5, 3.5], "n": np.int32(8)}],
        "ragged_sets": {10: {3, 1, 2}, 11: set(), 12: {99}},
        "ragged_lists": {0: [1.5], 5: [2.5, 3.5], 2: []},
        "scalar_map_nulls": {3: None, 1: 2.0, 2: None},
        "bool_vectors": {0: [True, False, True], 1: [False, False, False]},
        "tri_list": [(i, i + 1, i + 2) for i in range(20)], "tri_list_of_lists": [[i, i + 1, i + 2] for i in range(20)],
        "nested": {"a": {"b": {"c": [np.float32(1.5), np.int64(2), np.bool_(True)]}}},
        "arr2d": np.arange(12, dtype=np.float64).reshape(3, 4), "u8": np.array([1, 2, 3], np.uint8),
        "text": "naïve ✓", "none": None,
    }
    varout.save(str(tmp_path), vo)
    assert_same(vo, varout.load(str(tmp_path)))
    import json

    json.loads(open(tmp_path / "var_out.json").read(), parse_constant=lambda c: pytest.fail(f"non-standard JSON constant {c}"))
    with pytest.raises(TypeError):
        varout.save(str(tmp_path), {"f": lambda: 0})
    with pytest.raises(ValueError):
        varout.save(str(tmp_path), {"__set__": 1})


class _Boom:
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_legacy_var_out_npy_goes_through_the_numpy_only_unpickler(tmp_path):
    from same_amd import varout
    from same_amd.merge import load_matching_results

    vo = {"x": [np.float64(1.0), np.float64(0.0)], "violations": {"points_with_violations": [np.int64(3)], "s": {1, 2}},
          "triangle_data": {"triangles": np.arange(6).reshape(2, 3)}}
    np.save(tmp_path / "var_out.npy", vo, allow_pickle=True)          # what the reference writes (src/same.py:1455-1462)
    back = varout.load_legacy_npy(str(tmp_path / "var_out.npy"))
    assert back["x"] == [1.0, 0.0] and back["violations"]["s"] == {1, 2}
    assert np.array_equal(back["triangle_data"]["triangles"], vo["triangle_data"]["triangles"])
    for name in ("aligned_df", "ref_df", "matches_df"):
        pd.DataFrame({"X": [1.0]}).to_csv(tmp_path / f"{name}.csv", index=False)
    lv, la, lr, lm = load_matching_results(str(tmp_path))             # a reference-written directory
    assert lv["x"] == [1.0, 0.0] and len(la) == len(lr) == len(lm) == 1
    # a file that names anything else is refused before it can run
    np.save(tmp_path / "evil.npy", {"x": _Boom()}, allow_pickle=True)
    with pytest.raises(pickle.UnpicklingError):
        varout.load_legacy_npy(str(tmp_path / "evil.npy"))
    # the pickle-free pair wins when both are present
    varout.save(str(tmp_path), {"x": [5.0]})
    assert load_matching_results(str(tmp_path))[0] == {"x": [5.0]}


def test_product_never_loads_pickles_blindly():
    """Every np.load call in the package passes allow_pickle=False (checked on the syntax tree, not the text)."""
    import ast

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "same_amd")
    seen = 0
    for f in sorted(os.listdir(root)):
        if not f.endswith(".py"):
            continue
        for node in ast.walk(ast.parse(open(os.path.join(root, f)).read())):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "load" \
                    and isinstance(node.func.value, ast.Name) and node.func.value.id in ("np", "numpy"):
                seen += 1
                kw = {k.arg: k.value for k in node.keywords}
                assert isinstance(kw.get("allow_pickle"), ast.Constant) and kw["allow_pickle"].value is False, (f, node.lineno)
    assert seen >= 1
