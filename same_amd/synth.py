"""Seeded synthetic inputs for the pre-MIP path.

The reference's own generator (src/synthetic_datagen.py) draws a fixed 4-quadrant figure
with a global seed; benchmarks need arbitrary sizes, so this is the build's generator:
XY ~ U[0, side)^2 at density ``rho`` cells per unit area, type rows ~ Dirichlet(alpha)*100
(the reference's probability columns are on a 0-100 scale: src/synthetic_datagen.py:170),
``cell_type`` = argmax column name, ``size`` = 1.
"""
import numpy as np


def make_cells(n, n_types, seed, rho=0.01, alpha=0.3, side=None):
    """Return dict(xy (n,2) f64, types (n,T) f64, cell_type (n,) int32, size (n,) f64)."""
    rng = np.random.default_rng(seed)
    side = float(np.sqrt(n / rho)) if side is None else float(side)
    xy = rng.uniform(0.0, side, size=(n, 2))
    types = rng.dirichlet(np.full(n_types, alpha), size=n) * 100.0
    return {"xy": np.ascontiguousarray(xy), "types": np.ascontiguousarray(types),
            "cell_type": np.argmax(types, axis=1).astype(np.int32), "size": np.ones(n), "side": side}


def make_jittered(ref, seed, sigma=2.0, drop=0.05):
    """Moving section = ref + N(0, sigma^2) jitter with a fraction of rows dropped (sweep inputs)."""
    rng = np.random.default_rng(seed)
    n = len(ref["xy"])
    keep = rng.random(n) >= drop
    xy = ref["xy"][keep] + rng.normal(0.0, sigma, size=(int(keep.sum()), 2))
    return {"xy": np.ascontiguousarray(xy), "types": np.ascontiguousarray(ref["types"][keep]),
            "cell_type": ref["cell_type"][keep].copy(), "size": np.ones(int(keep.sum())), "side": ref["side"]}


def type_columns(n_types):
    return [f"c{t + 1}" for t in range(n_types)]


def to_frame(cells):
    """pandas frame with the columns run_same reads: X, Y, cell_type (str), c1..cT, size."""
    import pandas as pd

    cols = type_columns(cells["types"].shape[1])
    df = pd.DataFrame(cells["types"], columns=cols)
    df.insert(0, "Y", cells["xy"][:, 1])
    df.insert(0, "X", cells["xy"][:, 0])
    df["cell_type"] = [cols[t] for t in cells["cell_type"]]
    df["size"] = cells["size"]
    df["Cell_Num_Old"] = np.arange(len(df))
    return df
