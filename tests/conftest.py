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


@pytest.fixture(scope="session", autouse=True)
def _progress_heartbeat():
    """A run harness may treat several silent minutes as a hang.  pytest prints nothing while one test runs (a cold GPU box can
    spend minutes paging in librccl / the HIP runtime inside a single test), so a daemon thread notes the time in
    gpurun_out/pytest_heartbeat.log every 30 s for as long as the session lasts.  Best effort: any error disables it."""
    import threading
    import time

    stop = threading.Event()
    path = os.path.join(ROOT, "gpurun_out", "pytest_heartbeat.log")

    def beat():
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            while not stop.wait(30.0):
                with open(path, "a") as f:
                    f.write(f"{time.strftime('%H:%M:%S')} pytest session alive\n")
        except OSError:
            pass

    t = threading.Thread(target=beat, name="pytest-heartbeat", daemon=True)
    t.start()
    yield
    stop.set()
