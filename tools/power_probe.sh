#!/bin/bash
# Dev probe: board power / clocks (rocm-smi) while the dense kernel loops, vs idle.
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -E "GPU\[0\]|Power|sclk|mclk|fclk" | head -12
echo "--- under load (T=20 dense loop)"
python - <<'PY' &
import sys, ctypes, numpy as np
sys.path.insert(0, ".")
from same_amd import _lib, synth
T = int(__import__("os").environ.get("PROBE_T", "20"))
n = 100000
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
ref = synth.make_cells(n, max(T,1), seed=0); mov = synth.make_cells(n, max(T,1), seed=1, side=ref["side"])
dA, dR = ctx.to_device(mov["types"][:, :T].copy()), ctx.to_device(ref["types"][:, :T].copy())
dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
dD = ctx.alloc(n * n * 8)
import time
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(20):
        ctx.check(L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, dD.ptr, n), "d")
    ctx.sync()
PY
sleep 4
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" | head -4; sleep 0.5; done
wait
