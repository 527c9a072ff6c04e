"""Dev probe: do two kernels that write the same 80 GB really differ, or does the box drift?  Alternates the fp64 dense kernel
at T=0, the fixed-point kernel at T=0 and hipMemsetAsync in one process, several rounds."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from same_amd import _lib, synth

n = 100000
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
ref = synth.make_cells(n, 1, seed=0); mov = synth.make_cells(n, 1, seed=1, side=ref["side"])
dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
dz = ctx.alloc(64)
dD = ctx.alloc(n * n * 8)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if T:
    from same_amd import ops
    r = synth.make_cells(n, T, seed=0); m = synth.make_cells(n, T, seed=1, side=ref["side"])
    dA, dR = ctx.to_device(m["types"]), ctx.to_device(r["types"])
    off, l2 = ops.quantize_types(m["types"], r["types"])
    dAq, dRq = ctx.alloc(n * T * 4), ctx.alloc(n * T * 4)
    ctx.check(L.same_quantize_u32_dev(H, dA.ptr, n * T, off, 2.0 ** l2, dAq.ptr), "q"); ctx.check(L.same_quantize_u32_dev(H, dR.ptr, n * T, off, 2.0 ** l2, dRq.ptr), "q")
else:
    dA = dR = dAq = dRq = dz; l2 = 0

def t(call):
    out = []
    for _ in range(5):
        ctx.check(L.same_timer_start(H), "t"); ctx.check(call(), "k")
        v = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(v)), "t"); out.append(v.value)
    return float(np.mean(out[1:]))

for rnd in range(6):
    a = t(lambda: L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, dD.ptr, n))
    b = t(lambda: L.same_dense_cost_q32_dev(H, dAq.ptr, dRq.ptr, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, 2.0 ** -l2, 1e-6 if T else 0.0, dD.ptr, n))
    c = t(lambda: L.same_dev_memset(H, dD.ptr, 0, n * n * 8))
    print(f"round {rnd}: fp64 kernel T={T} {a:6.2f} ms   q32 kernel T={T} {b:6.2f} ms   memset {c:6.2f} ms", flush=True)
