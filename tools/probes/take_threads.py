"""Is a table's row gather faster column-parallel?  (merge._take_rows, incumbent._TableBuilder.table)  Run on the GPU box's host."""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

rng = np.random.default_rng(0)
n = 1_000_000
cols = [rng.random(n) for _ in range(18)] + [np.arange(n), np.arange(n) * 3, rng.random(n) < 0.5, rng.random(n) < 0.5]
rows = np.sort(rng.choice(n, 950_000, replace=False))
for workers in (1, 2, 4, 8):
    pool = ThreadPoolExecutor(workers)
    best = 1e9
    for _ in range(8):
        t = time.perf_counter()
        out = list(pool.map(lambda c: c.take(rows), cols)) if workers > 1 else [c.take(rows) for c in cols]
        best = min(best, time.perf_counter() - t)
    print(f"{workers} thread(s): {best * 1e3:.1f} ms for {len(cols)} columns x {len(rows)} rows", flush=True)
