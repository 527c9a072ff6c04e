"""Dev probe: does the row pitch decide the streaming-store rate?  T=0 dense kernel (pure store of 100000 x 100000 doubles)
into two separately allocated buffers at several row pitches (ld), in one process."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from same_amd import _lib, synth

n = 100000
lds = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [100000, 100002, 100032, 100096, 100352, 100864, 102400, 131072]
tag = ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('SAME_DENSE_'))
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

bufs = [ctx.alloc(n * max(lds) * 8) for _ in range(2)]
for b in bufs:
    res = []
    for ld in lds:
        ms = t(lambda: L.same_dense_cost_f64_dev(H, dz.ptr, dz.ptr, 0, dax.ptr, drx.ptr, n, 0, n, 1.0, b.ptr, ld))
        res.append(f"ld={ld}: {ms:6.2f}")
    mm = t(lambda: L.same_dev_memset(H, b.ptr, 0, n * n * 8))
    print(f"{tag} @{b.ptr:#x}  " + "  ".join(res) + f"  memset80GB {mm:6.2f}", flush=True)
