"""Dev probe: product T=0 dense kernel and the stand-alone pattern kernel on the SAME buffers (three 80 GB allocations)."""
import ctypes, os, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from same_amd import _lib, synth

n = 100000
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
P = ctypes.CDLL(os.path.join(here, "libpattern.so"))
P.pattern_time.restype = ctypes.c_float
P.pattern_time.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint, ctypes.c_int]
ref = synth.make_cells(n, 1, seed=0); mov = synth.make_cells(n, 1, seed=1, side=ref["side"])
dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
dz = ctx.alloc(64)

def t(call, reps=4):
    out = []
    for _ in range(reps):
        ctx.check(L.same_timer_start(H), "t"); ctx.check(call(), "k")
        v = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(v)), "t"); out.append(v.value)
    return float(np.mean(out[1:]))

bufs = [ctx.alloc(n * n * 8) for _ in range(3)]
for b in bufs:
    k = t(lambda: L.same_dense_cost_f64_dev(H, dz.ptr, dz.ptr, 0, dax.ptr, drx.ptr, n, 0, n, 1.0, b.ptr, n))
    ctx.sync()
    pats = "  ".join(f"R={R}/rpb={rpb} {P.pattern_time(b.ptr, n * 8, n, n * 8, rpb, R, 4):.2f}" for R in (1, 8, 64, 256) for rpb in (64, 256))
    k2 = t(lambda: L.same_dense_cost_f64_dev(H, dz.ptr, dz.ptr, 0, dax.ptr, drx.ptr, n, 0, n, 1.0, b.ptr, n))
    print(f"@{b.ptr:#x} product T=0 {k:.2f} / {k2:.2f} ms | pattern: {pats}", flush=True)
