"""Soak: same_batched_assign against the oracle's restatement of scipy's solver on many random / tied small problems."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import same_oracle as orc
from same_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot = 0
t0 = time.time()
for rnd in range(rounds):
    rng = np.random.default_rng(1000 + rnd)
    n = 40000
    hi_a, hi_r = int(rng.choice([4, 8, 20, 45])), int(rng.choice([4, 8, 20, 45]))
    na, nr = rng.integers(0, hi_a, n), rng.integers(1, hi_r, n)
    a_off, r_off = np.concatenate(([0], np.cumsum(na))), np.concatenate(([0], np.cumsum(nr)))
    mode = rnd % 4
    if mode == 0:
        axy, rxy = rng.uniform(0, 100, (a_off[-1], 2)), rng.uniform(0, 100, (r_off[-1], 2))
    elif mode == 1:
        axy, rxy = rng.integers(0, 5, (a_off[-1], 2)).astype(float), rng.integers(0, 5, (r_off[-1], 2)).astype(float)
    elif mode == 2:
        axy = rng.integers(0, 3, (a_off[-1], 2)) * 0.1 + rng.choice([0, 1e-9], (a_off[-1], 2))
        rxy = rng.integers(0, 3, (r_off[-1], 2)) * 0.1
    else:   # collinear members, duplicate points
        axy = np.column_stack((rng.integers(0, 6, a_off[-1]).astype(float), np.zeros(a_off[-1])))
        rxy = np.column_stack((rng.integers(0, 6, r_off[-1]).astype(float), np.zeros(r_off[-1])))
    got, want = ops.batched_assign(a_off, r_off, axy, rxy), orc.batched_assign(a_off, r_off, axy, rxy)
    assert np.array_equal(got, want), (rnd, mode)
    tot += n
    print(f"round {rnd}: {tot} problems ok ({time.time() - t0:.0f} s)", flush=True)
