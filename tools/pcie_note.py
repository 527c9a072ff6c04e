"""PCIe-inclusive rates of the host-buffer entry points (DESIGN.md section 7's note): the same calls with operands on the host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import ops, synth

ref = synth.make_cells(10000, 20, seed=0); mov = synth.make_cells(10000, 20, seed=1, side=ref["side"])


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


t = best(lambda: ops.dense_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], 1.0))
print(f"same_dense_cost_f64 (host buffers), 10k x 10k, T=20: {t * 1e3:.1f} ms incl. H2D + 0.8 GB D2H = {1e8 / t:.2e} cell-pairs/s, {0.8 / t:.1f} GB/s of output over PCIe")
idx, _, _ = ops.knn_prune(mov["xy"], ref["xy"], 25.0, 32)
rr, cc = np.nonzero(idx >= 0); pairs = np.column_stack((rr, idx[rr, cc])).astype(np.int32)
t = best(lambda: ops.knn_prune(mov["xy"], ref["xy"], 25.0, 32))
print(f"same_knn_prune (host buffers), 10k x 10k, k=32: {t * 1e3:.2f} ms = {1e8 / t:.2e} dense-equivalent cell-pairs/s")
t = best(lambda: ops.pair_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], pairs, 1.0))
print(f"same_pair_cost_f64 (host buffers), {len(pairs)} pairs, T=20: {t * 1e3:.2f} ms = {len(pairs) / t:.2e} pairs/s")
