"""Timing of same_batched_assign (f4) at metacell-flow shapes: 1e6 matches with member lists of 1..3 (MS=3) and 1..9 (MS=9)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from same_amd import ops

for hi in (4, 10):
    rng = np.random.default_rng(hi)
    n = 1_000_000
    na, nr = rng.integers(1, hi, n), rng.integers(1, hi, n)
    a_off, r_off = np.concatenate(([0], np.cumsum(na))), np.concatenate(([0], np.cumsum(nr)))
    axy, rxy = rng.uniform(0, 100, (a_off[-1], 2)), rng.uniform(0, 100, (r_off[-1], 2))
    ops.batched_assign(a_off[:1001], r_off[:1001], axy[:a_off[1000]], rxy[:r_off[1000]])   # warm up
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); out = ops.batched_assign(a_off, r_off, axy, rxy); best = min(best, time.perf_counter() - t)
    print(f"members 1..{hi - 1}: {n} problems, {a_off[-1]} aligned members: {best * 1e3:.1f} ms end to end through the host-buffer entry point "
          f"(H2D of offsets/coordinates + kernel + D2H) = {n / best / 1e6:.1f} M problems/s", flush=True)
    if hi == 4:
        from scipy.optimize import linear_sum_assignment
        from scipy.spatial.distance import cdist
        m = 20000
        t = time.perf_counter()
        for p in range(m):
            d = cdist(axy[a_off[p]:a_off[p + 1]], rxy[r_off[p]:r_off[p + 1]])
            if d.shape[0] > d.shape[1]:
                d = np.tile(d, (1, int(np.ceil(d.shape[0] / d.shape[1]))))
            linear_sum_assignment(d)
        dt = time.perf_counter() - t
        print(f"scipy cdist + linear_sum_assignment loop (the reference's inner work, without its pandas lookups): {m / dt / 1e6:.3f} M problems/s", flush=True)
