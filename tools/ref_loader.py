"""Load the reference's hot-path modules for fixture generation (build container only).

The reference package cannot be imported whole (`src/__init__.py` pulls in gurobipy,
which is proprietary and absent).  This loader follows SURVEY.md §8(c): empty stub
modules for the absent third-party imports (none of them is touched by the hot-path
functions), then each reference file is loaded by path under a synthetic package so
its relative imports resolve.  Nothing under /root/reference is copied or written
(`sys.dont_write_bytecode`), and this module refuses to work when the reference is
not mounted -- it never travels to the GPU box.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("SAME_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")
_PKG = "_same_reference"


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference(with_run_same=False):
    """Return a namespace with the reference modules that hold hot-path functions.  with_run_same=True also loads
    src/same.py (run_same, sliding_window_matching); the caller must have put a solver double into sys.modules['gurobipy']."""
    if not os.path.isdir(REF_SRC):
        raise RuntimeError(f"reference not mounted at {REF_ROOT}; fixtures can only be generated in the build container")
    sys.dont_write_bytecode = True

    class _Dummy:  # placeholder for gurobipy names imported at module scope
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Dummy()

    _stub("gurobipy", Model=_Dummy, GRB=_Dummy(), quicksum=lambda *a, **k: None)
    _stub("scanpy")
    _stub("alphashape", alphashape=None)
    shp = _stub("shapely")
    geo = _stub("shapely.geometry", Point=_Dummy, Polygon=_Dummy, MultiPolygon=_Dummy, GeometryCollection=_Dummy)
    shp.geometry = geo
    _stub("shapely.ops", unary_union=lambda *a, **k: None)

    pkg = types.ModuleType(_PKG)
    pkg.__path__ = [REF_SRC]
    sys.modules[_PKG] = pkg
    ns = types.SimpleNamespace()
    names = ("utils", "helpers", "violationhelper", "init_helpers", "knn_utils", "eval_utils", "metacell_utils")
    if with_run_same:
        names += ("triangle_utils", "same")
    for name in names:
        spec = importlib.util.spec_from_file_location(f"{_PKG}.{name}", os.path.join(REF_SRC, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{_PKG}.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(ns, name, mod)
    return ns


if __name__ == "__main__":
    ref = load_reference()
    print("loaded:", [m for m in vars(ref)])
