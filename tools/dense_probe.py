"""Dev probe: dense cost kernel time vs T (store-bound at T=0, VALU-bound as T grows)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
Ts = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4, 8, 12, 16, 20, 24]
q32 = len(sys.argv) > 3 and sys.argv[3] == "q32"      # the opt-in fixed-point build (fp64 output)
dtype = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
ctx = _lib.Context(0)
L, H = ctx.lib, ctx.handle
es = np.dtype(dtype).itemsize
ld = int(os.environ.get("PROBE_LD", n))
dD = ctx.alloc_spread(n * ld * es)   # over the HBM regions (SAME_SPREAD=0: plain hipMalloc)
print("output buffer:", dD.spread_info, flush=True)
for T in Ts:
    ref = synth.make_cells(n, max(T, 1), seed=0); mov = synth.make_cells(n, max(T, 1), seed=1, side=ref["side"])
    A = mov["types"][:, :T].astype(dtype); R = ref["types"][:, :T].astype(dtype)
    dA, dR = ctx.to_device(A), ctx.to_device(R)
    dax, drx = ctx.to_device(mov["xy"].astype(dtype)), ctx.to_device(ref["xy"].astype(dtype))
    fn = L.same_dense_cost_f64_dev if dtype == np.float64 else L.same_dense_cost_f32_dev
    if q32:
        from same_amd import ops
        off, l2 = ops.quantize_types(A, R) if T else (0.0, 0)
        dAq, dRq = ctx.alloc(max(A.size, 1) * 4), ctx.alloc(max(R.size, 1) * 4)
        ctx.check(L.same_quantize_u32_dev(H, dA.ptr, A.size, off, 2.0 ** l2, dAq.ptr), "q")
        ctx.check(L.same_quantize_u32_dev(H, dR.ptr, R.size, off, 2.0 ** l2, dRq.ptr), "q")
    ms_all = []
    for it in range(6):
        ctx.check(L.same_timer_start(H), "t")
        if q32:
            ctx.check(L.same_dense_cost_q32_dev(H, dAq.ptr, dRq.ptr, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, 2.0 ** -l2,
                                                float(os.environ.get("PROBE_REL_TOL", "1e-6")), dD.ptr, ld), "dense q32")
        else:
            ctx.check(fn(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, dD.ptr, ld), "dense")
        ms = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(ms)), "t"); ms_all.append(ms.value)
    best = min(ms_all[1:]); mean = float(np.mean(ms_all[1:]))
    byts = es * n * n + es * (T + 2) * 2 * n
    print(f"ld={ld} T={T:3d} {'q32->f64' if q32 else np.dtype(dtype).name} best {best:8.3f} ms mean {mean:8.3f} ms  {byts/best/1e6:8.1f} GB/s  valu/pair~{2*T+5}", flush=True)
