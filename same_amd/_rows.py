"""Lists that remember the array they were made from.

The reference hands Python lists across its internal seams -- the kept triangles as a list of 3-vectors
(src/helpers.py:395), the pair costs as a list of float64 (src/same.py:1180-1189), weights and signs as lists
(:1128-1146) -- and the drop-in functions here return the same.  Inside the package those values come from arrays and go
back into arrays; converting 20 000 little row arrays (or 100 000 boxed floats) back with np.asarray costs more than
the kernels that produced them.  RowList / ValueList are plain lists for every outside purpose, plus the source array
for consumers in the package (rows_array / values_array fall back to np.asarray when the list was edited)."""
import numpy as np


class RowList(list):
    """list of the rows of a 2-D array"""
    __slots__ = ("array",)

    def __init__(self, array):
        super().__init__(array)
        self.array = array


class ValueList(list):
    """list of the items of a 1-D array"""
    __slots__ = ("array",)

    def __init__(self, array):
        super().__init__(array)
        self.array = array


def _same(x, y):
    x, y = np.asarray(x), np.asarray(y)
    return x.shape == y.shape and bool(np.all((x == y) | ((x != x) & (y != y))))


def _source(seq):
    """the remembered array, unless the list has been edited since (length, first and last item are checked)"""
    arr = getattr(seq, "array", None)
    if arr is None or len(arr) != len(seq):
        return None
    if len(seq) and not (_same(seq[0], arr[0]) and _same(seq[-1], arr[-1])):
        return None
    return arr


def rows_array(seq, width=3, dtype=None):
    """(n, width) array of a list of rows (or of an array): the remembered array when there is one."""
    arr = _source(seq) if isinstance(seq, RowList) else None
    if arr is None:
        arr = np.asarray(seq)
        arr = arr.reshape(-1, width) if arr.size else np.zeros((0, width), dtype=int)
    return arr if dtype is None else np.asarray(arr, dtype=dtype)


def values_array(seq, dtype=None):
    arr = _source(seq) if isinstance(seq, ValueList) else None
    if arr is None:
        arr = np.asarray(seq) if dtype is None else np.asarray(seq, dtype=dtype)
    return arr if dtype is None else np.asarray(arr, dtype=dtype)
