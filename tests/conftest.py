import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))


def frames_from_golden(g):
    """Rebuild the (aligned, ref) input frames a fixture was generated from."""
    import pandas as pd

    cols = [str(c) for c in g["commonCT"]]
    out = []
    for p in ("in_aligned", "in_ref"):
        df = pd.DataFrame(g[f"{p}_types"], columns=cols)
        df.insert(0, "Y", g[f"{p}_xy"][:, 1])
        df.insert(0, "X", g[f"{p}_xy"][:, 0])
        df["cell_type"] = g[f"{p}_cell_type"].astype(object)
        df["size"] = g[f"{p}_size"]
        df["__row"] = np.arange(len(df))
        df["Cell_Num_Old"] = np.arange(len(df))
        out.append(df)
    return out[0], out[1], cols


@pytest.fixture(scope="session")
def oracle():
    from oracle import same_oracle

    same_oracle.lib()
    return same_oracle


FULL_CASES = ["synthetic_example", "cfg1_500", "cfg2_small"]
