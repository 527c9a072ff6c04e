"""Pair costs (the objective coefficients, src/same.py:1180-1189) and the dense cost builder."""
import numpy as np

from . import ops
from ._rows import ValueList


def _xy(df):
    return np.ascontiguousarray(df[["X", "Y"]].to_numpy(dtype=np.float64))


def pair_costs(aligned_df, ref_df, valid_pairs, commonCT, dist_ct_coeff, ctx=None, dtype=np.float64, _as_array=False):
    """-> list of np.float64, one per pair, in pair order (what run_same calls `c`).
    dtype=float32 (BASELINE config 5, `optim_params['hip_cost_dtype']='float32'`): operands and arithmetic in float;
    the values are handed on as float64 (every float is one), so the solver-facing type does not change."""
    cols = list(commonCT)
    A = aligned_df[cols].to_numpy(dtype=np.float64)
    R = ref_df[cols].to_numpy(dtype=np.float64)
    pairs = np.asarray(valid_pairs, dtype=np.int64).reshape(-1, 2)
    c = ops.pair_cost(A, R, _xy(aligned_df), _xy(ref_df), pairs, dist_ct_coeff, dtype=dtype, ctx=ctx).astype(np.float64, copy=False)
    return c if _as_array else ValueList(c)      # _as_array: package-internal, skips boxing 10^5..10^6 floats


def dense_cost_matrix(aligned_df, ref_df, commonCT, dist_ct_coeff, row_begin=0, row_end=None, dtype=np.float64, ctx=None):
    """The same cost for every (aligned i, ref j): rows [row_begin,row_end) x all refs."""
    cols = list(commonCT)
    return ops.dense_cost(aligned_df[cols].to_numpy(dtype=np.float64), ref_df[cols].to_numpy(dtype=np.float64),
                          _xy(aligned_df), _xy(ref_df), dist_ct_coeff, row_begin, row_end, dtype, ctx=ctx)
