"""Dev probe: is the streaming-store rate a property of the box or of where the 80 GB buffer landed?  Times the T=0 dense kernel
(pure 80 GB store) into three separately allocated 80 GB buffers, then frees and re-allocates them and does it again."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from same_amd import _lib, synth

n = 100000
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
ref = synth.make_cells(n, 1, seed=0); mov = synth.make_cells(n, 1, seed=1, side=ref["side"])
dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
dz = ctx.alloc(64)

def t(call):
    out = []
    for _ in range(4):
        ctx.check(L.same_timer_start(H), "t"); ctx.check(call(), "k")
        v = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(v)), "t"); out.append(v.value)
    return float(np.mean(out[1:]))

for gen in range(3):
    bufs = [ctx.alloc(n * n * 8) for _ in range(3)]
    for rnd in range(2):
        line = []
        for b in bufs:
            k = t(lambda: L.same_dense_cost_f64_dev(H, dz.ptr, dz.ptr, 0, dax.ptr, drx.ptr, n, 0, n, 1.0, b.ptr, n))
            m = t(lambda: L.same_dev_memset(H, b.ptr, 0, n * n * 8))
            line.append(f"@{b.ptr:#x} kernel {k:6.2f} memset {m:6.2f}")
        print(f"gen {gen} round {rnd}: " + " | ".join(line), flush=True)
    for b in bufs:
        b.free()
