"""Dev probe: where inside an 80 GB buffer is the store rate lost?  Two separately allocated 80 GB buffers; the T=0 dense
kernel (pure store) over each whole buffer and over 20 row slices of 4 GB each."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from same_amd import _lib, synth

n = 100000
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
ref = synth.make_cells(n, 1, seed=0); mov = synth.make_cells(n, 1, seed=1, side=ref["side"])
dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
dz = ctx.alloc(64)

def t(call, reps=4):
    out = []
    for _ in range(reps):
        ctx.check(L.same_timer_start(H), "t"); ctx.check(call(), "k")
        v = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(v)), "t"); out.append(v.value)
    return float(np.mean(out[1:]))

bufs = [ctx.alloc(n * n * 8) for _ in range(2)]
S = 5000
for b in bufs:
    whole = t(lambda: L.same_dense_cost_f64_dev(H, dz.ptr, dz.ptr, 0, dax.ptr, drx.ptr, n, 0, n, 1.0, b.ptr, n))
    sl = [t(lambda: L.same_dense_cost_f64_dev(H, dz.ptr, dz.ptr, 0, dax.ptr, drx.ptr, n, r0, r0 + S, 1.0, b.ptr + r0 * n * 8, n))
          for r0 in range(0, n, S)]
    print(f"@{b.ptr:#x} whole {whole:6.2f} ms; 4 GB slices (ms): " + " ".join(f"{x:.3f}" for x in sl) + f"  sum {sum(sl):.2f}", flush=True)
