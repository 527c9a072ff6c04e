"""`var_out` on disk without pickle.

The reference dumps the nested `var_out` dict of run_same with `np.save(..., allow_pickle=True)` into `var_out.npy`
(src/same.py:1455-1462) and reads it back with `np.load(..., allow_pickle=True)` (src/helpers.py:667-689): loading such a
file executes whatever the pickle says.  Here the same dict is written as two plain files beside the CSVs:

  var_out.json   the structure: dict / list / tuple / set nesting, strings, small values; every bulk numeric member is a
                 reference into the .npz
  var_out.npz    the numbers: one array per numeric sequence, and columnar blocks for the three shapes that make up the
                 bulk of var_out -- lists of same-shaped records (the x/y-order violation lists), int-keyed maps of
                 same-shaped records (triangle_info), int-keyed maps of scalars / short vectors / index sets
                 (areas_before/after, matched_vertices, aligned_simplex_map)

`load` rebuilds an equal dict (same keys in the same order, lists / tuples / sets / arrays as they were, numbers as
Python numbers).  A legacy `var_out.npy` is still readable, but only through an unpickler that admits nothing except
numpy array / dtype / scalar reconstruction and `set` -- never plain `allow_pickle=True`.
"""
import json
import math
import os
import pickle

import numpy as np

FORMAT_VERSION = 1
INLINE_MAX = 8          # numeric sequences up to this length stay in the JSON
_NUM = (int, float, bool, np.integer, np.floating, np.bool_)


def _is_num(v):
    return isinstance(v, _NUM)


def _plain(v):
    """numpy scalar -> Python scalar; non-finite floats get a tag (strict JSON has no NaN)."""
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    f = float(v)
    if math.isfinite(f):
        return f
    return {"__float__": "nan" if f != f else ("inf" if f > 0 else "-inf")}


def _seq_kind(o):
    return "tuple" if isinstance(o, tuple) else ("ndarray" if isinstance(o, np.ndarray) else "list")


def _numeric_array(seq):
    """1-D/ND numeric array of a homogeneous numeric sequence, or None."""
    if isinstance(seq, np.ndarray):
        return seq if seq.dtype.kind in "iufb" else None
    if len(seq) == 0 or not all(_is_num(v) for v in seq):
        return None
    if all(isinstance(v, (bool, np.bool_)) for v in seq):
        return np.array(seq, dtype=bool)
    if all(isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_)) for v in seq):
        return np.array(seq, dtype=np.int64)
    return np.array(seq, dtype=np.float64)


class _Encoder:
    def __init__(self):
        self.arrays = {}

    def put(self, arr):
        key = f"a{len(self.arrays)}"
        self.arrays[key] = np.ascontiguousarray(arr)
        return key

    # ---- record schemas: nested str-keyed dicts whose leaves are numbers or fixed-length numeric sequences -------------
    def _schema(self, rec):
        if not isinstance(rec, dict) or not all(isinstance(k, str) for k in rec):
            return None
        out = {}
        for k, v in rec.items():
            if _is_num(v):
                out[k] = "n"
            elif isinstance(v, dict):
                sub = self._schema(v)
                if sub is None:
                    return None
                out[k] = sub
            elif isinstance(v, (list, tuple, np.ndarray)) and len(v) and _numeric_array(v) is not None and np.ndim(v) == 1:
                out[k] = [_seq_kind(v), len(v)]
            else:
                return None
        return out

    def _columns(self, records, schema):
        """schema with every leaf replaced by the npz key of its column over all records"""
        out = {}
        for k, leaf in schema.items():
            vals = [r[k] for r in records]
            if isinstance(leaf, dict):
                out[k] = self._columns(vals, leaf)
            elif leaf == "n":
                out[k] = {"__col__": self.put(_numeric_array(vals))}
            else:
                out[k] = {"__col__": self.put(np.array([np.asarray(v) for v in vals])), "seq": leaf[0]}
        return out

    def _records(self, records):
        if len(records) < 2:
            return None
        schema = self._schema(records[0])
        if schema is None or any(self._schema(r) != schema for r in records[1:]):
            return None
        return self._columns(records, schema)

    # ---- int-keyed maps -------------------------------------------------------------------------------------------
    def _int_map(self, d):
        keys = list(d.keys())
        if len(keys) < 2 or not all(isinstance(k, (int, np.integer)) and not isinstance(k, (bool, np.bool_)) for k in keys):
            return None
        vals = list(d.values())
        head = {"__map__": None, "keys": self.put(np.array(keys, dtype=np.int64))}
        if all(v is None or _is_num(v) for v in vals):
            null = np.array([v is None for v in vals])
            arr = _numeric_array([0 if v is None else v for v in vals])
            head.update({"__map__": "scalar", "values": self.put(arr)})
            if null.any():
                head["null"] = self.put(null)
            return head
        if all(isinstance(v, dict) for v in vals):
            cols = self._records(vals)
            if cols is not None:
                head.update({"__map__": "records", "columns": cols})
                return head
            return None
        if all(isinstance(v, (list, tuple, np.ndarray, set, frozenset)) for v in vals):
            kinds = {"set" if isinstance(v, (set, frozenset)) else _seq_kind(v) for v in vals}
            if len(kinds) != 1:
                return None
            kind = kinds.pop()
            rows = [sorted(v) if kind == "set" else list(v) for v in vals]
            flat = [x for r in rows for x in r]
            if not all(_is_num(x) for x in flat):
                return None
            arr = _numeric_array(flat) if flat else np.zeros(0, np.int64)
            off = np.zeros(len(rows) + 1, np.int64)
            np.cumsum([len(r) for r in rows], out=off[1:])
            head.update({"__map__": "ragged", "kind": kind, "offsets": self.put(off), "values": self.put(arr)})
            return head
        return None

    def _rows(self, seq):
        """a long list of equal-length numeric rows (triangles given as a list of triples) -> one 2-D array"""
        if len(seq) <= INLINE_MAX or not all(isinstance(r, (list, tuple, np.ndarray)) for r in seq):
            return None
        kinds, lens = {_seq_kind(r) for r in seq}, {len(r) for r in seq}
        if len(kinds) != 1 or len(lens) != 1 or 0 in lens:
            return None
        flat = _numeric_array([x for r in seq for x in (r.tolist() if isinstance(r, np.ndarray) else r)]) \
            if all(np.ndim(r) == 1 for r in seq) else None
        if flat is None:
            return None
        return {"__rows__": kinds.pop(), "outer": _seq_kind(seq), "array": self.put(flat.reshape(len(seq), -1))}

    # ---- anything ---------------------------------------------------------------------------------------------------
    def enc(self, o):
        if o is None or isinstance(o, str):
            return o
        if _is_num(o):
            return _plain(o)
        if isinstance(o, dict):
            if all(isinstance(k, str) for k in o):
                if any(k.startswith("__") and k.endswith("__") for k in o):
                    raise ValueError("dict keys of the form __name__ are reserved by the var_out format")
                return {k: self.enc(v) for k, v in o.items()}
            m = self._int_map(o)
            if m is not None:
                return m
            return {"__map__": "pairs", "items": [[self.enc(k), self.enc(v)] for k, v in o.items()]}
        if isinstance(o, (set, frozenset)):
            try:
                members = sorted(o)
            except TypeError:
                members = list(o)
            return {"__set__": self.enc(members)}
        if isinstance(o, (list, tuple, np.ndarray)):
            kind = _seq_kind(o)
            arr = _numeric_array(o)
            if arr is not None and (arr.ndim > 1 or len(arr) > INLINE_MAX or kind == "ndarray"):
                return {"__seq__": kind, "array": self.put(arr)}
            if kind == "list" or kind == "tuple":
                rows = self._rows(o)
                if rows is not None:
                    return rows
            if kind == "list":
                recs = self._records(o) if all(isinstance(r, dict) for r in o) else None
                if recs is not None:
                    return {"__records__": recs, "n": len(o)}
                return [self.enc(v) for v in o]
            return {"__seq__": kind, "items": [self.enc(v) for v in o]}
        raise TypeError(f"var_out member of type {type(o).__name__} has no pickle-free encoding")


class _Decoder:
    def __init__(self, arrays):
        self.arrays = arrays

    def _col_rows(self, cols, n):
        """columns -> list of n record dicts"""
        built = {}
        for k, leaf in cols.items():
            if "__col__" in leaf:
                a = self.arrays[leaf["__col__"]]
                rows = a.tolist()
                if "seq" in leaf:
                    rows = [tuple(r) if leaf["seq"] == "tuple" else (np.array(r, dtype=a.dtype) if leaf["seq"] == "ndarray" else r)
                            for r in rows]
                built[k] = rows
            else:
                built[k] = self._col_rows(leaf, n)
        return [{k: built[k][i] for k in cols} for i in range(n)]

    def dec(self, o):
        if isinstance(o, list):
            return [self.dec(v) for v in o]
        if not isinstance(o, dict):
            return o
        if "__float__" in o:
            return float(o["__float__"])
        if "__set__" in o:
            return set(self.dec(o["__set__"]))
        if "__seq__" in o:
            kind = o["__seq__"]
            if "array" in o:
                a = self.arrays[o["array"]]
                return a if kind == "ndarray" else (tuple(a.tolist()) if kind == "tuple" else a.tolist())
            items = [self.dec(v) for v in o["items"]]
            return tuple(items) if kind == "tuple" else np.array(items)
        if "__records__" in o:
            return self._col_rows(o["__records__"], int(o["n"]))
        if "__rows__" in o:
            a = self.arrays[o["array"]]
            inner = {"list": list, "tuple": tuple, "ndarray": lambda r: np.array(r, dtype=a.dtype)}[o["__rows__"]]
            rows = [inner(r) for r in a.tolist()]
            return tuple(rows) if o["outer"] == "tuple" else rows
        if "__map__" in o:
            how = o["__map__"]
            if how == "pairs":
                return {self._key(self.dec(k)): self.dec(v) for k, v in o["items"]}
            keys = self.arrays[o["keys"]].tolist()
            if how == "scalar":
                vals = self.arrays[o["values"]].tolist()
                if "null" in o:
                    vals = [None if z else v for v, z in zip(vals, self.arrays[o["null"]].tolist())]
                return dict(zip(keys, vals))
            if how == "records":
                return dict(zip(keys, self._col_rows(o["columns"], len(keys))))
            if how == "ragged":
                off, flat = self.arrays[o["offsets"]].tolist(), self.arrays[o["values"]]
                make = {"set": set, "tuple": tuple, "list": list, "ndarray": lambda r: np.array(r, dtype=flat.dtype)}[o["kind"]]
                flat = flat.tolist()
                return {k: make(flat[off[i]: off[i + 1]]) for i, k in enumerate(keys)}
            raise ValueError(f"unknown __map__ kind {how!r}")
        return {k: self.dec(v) for k, v in o.items()}

    @staticmethod
    def _key(k):
        return tuple(k) if isinstance(k, list) else k


def save(outprefix, var_out):
    """Write var_out.json + var_out.npz into the directory `outprefix`."""
    enc = _Encoder()
    doc = {"format": "same_amd.var_out", "version": FORMAT_VERSION, "var_out": enc.enc(var_out)}
    tmp = os.path.join(outprefix, "var_out.json.tmp")
    with open(tmp, "w") as f:
        json.dump(doc, f, allow_nan=False)
    tmp_npz = os.path.join(outprefix, "var_out.npz.tmp")
    with open(tmp_npz, "wb") as f:                               # a file object: savez would append ".npz" to a temp NAME
        np.savez_compressed(f, **enc.arrays)
    # both halves are complete before either appears; the arrays go first and the structure last, so a reader that finds
    # a new var_out.json finds its arrays, and a crash in between leaves the OLD json beside arrays it does not index into
    # wrongly only for the instant between the two renames (each rename is atomic)
    os.replace(tmp_npz, os.path.join(outprefix, "var_out.npz"))
    os.replace(tmp, os.path.join(outprefix, "var_out.json"))


def load(outprefix):
    with open(os.path.join(outprefix, "var_out.json")) as f:
        doc = json.load(f)
    if doc.get("format") != "same_amd.var_out" or int(doc.get("version", -1)) > FORMAT_VERSION:
        raise ValueError(f"{outprefix}/var_out.json is not a var_out file this version reads")
    with np.load(os.path.join(outprefix, "var_out.npz"), allow_pickle=False) as z:
        arrays = {k: z[k] for k in z.files}
    return _Decoder(arrays).dec(doc["var_out"])


# ---- legacy var_out.npy (a pickled dict inside an .npy) ----------------------------------------------------------------
class _NumpyOnlyUnpickler(pickle.Unpickler):
    """Admits what a var_out pickle consists of -- numpy array / dtype / scalar reconstruction and `set` -- and nothing
    else: no other callable can be named by the file."""

    ADMITTED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("builtins", "set"), ("builtins", "frozenset")}

    def find_class(self, module, name):
        if (module, name) in self.ADMITTED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"var_out.npy names {module}.{name}, which a var_out file has no business calling")


def load_legacy_npy(path):
    """The reference's `var_out.npy` (src/same.py:1455-1462) through the numpy-only unpickler."""
    with open(path, "rb") as f:
        version = np.lib.format.read_magic(f)
        if version == (1, 0):
            np.lib.format.read_array_header_1_0(f)
        elif version == (2, 0):
            np.lib.format.read_array_header_2_0(f)
        else:
            raise ValueError(f"unsupported .npy version {version}")
        obj = _NumpyOnlyUnpickler(f).load()
    return obj.item() if isinstance(obj, np.ndarray) and obj.shape == () else obj
