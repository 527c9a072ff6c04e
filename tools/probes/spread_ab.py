"""Dev probe: T=20 fp64 dense build into a plain and into a spread buffer, alternating, several rounds (power/thermal drift
would otherwise decide the comparison).  Usage: python tools/probes/spread_ab.py [T] [rounds]"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from same_amd import _lib, synth

n = 100000
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
ref = synth.make_cells(n, max(T, 1), seed=0); mov = synth.make_cells(n, max(T, 1), seed=1, side=ref["side"])
dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
dA, dR = ctx.to_device(mov["types"]), ctx.to_device(ref["types"])

def t(buf, TT, reps=20):
    out = []
    for _ in range(reps):
        ctx.check(L.same_timer_start(H), "t")
        ctx.check(L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, TT, dax.ptr, drx.ptr, n, 0, n, 1.0, buf.ptr, n), "k")
        v = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(v)), "t"); out.append(v.value)
    return float(np.mean(out[2:]))

plain = ctx.alloc(n * n * 8)
sp = ctx.alloc_spread(n * n * 8)
print("spread info:", json.dumps(sp.spread_info), flush=True)
for r in range(rounds):
    a0, b0 = t(plain, 0, 6), t(sp, 0, 6)
    a, b = t(plain, T), t(sp, T)
    a2, b2 = t(plain, T), t(sp, T)
    print(f"round {r}: T=0 plain {a0:6.2f} spread {b0:6.2f} | T={T} plain {a:6.2f} spread {b:6.2f} | again plain {a2:6.2f} spread {b2:6.2f} ms", flush=True)
