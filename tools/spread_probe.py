"""Store rate of the dense build into a plain hipMalloc buffer and into a spread buffer (same_dev_alloc_spread), plus the
integrity checks that matter for memory mapped through the virtual-memory API: values written through the spread range
read back exactly, two live spread buffers do not alias, and a buffer allocated after a free is clean of the old mapping.
Usage: python tools/spread_probe.py [n] [T]"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
ref = synth.make_cells(n, max(T, 1), seed=0); mov = synth.make_cells(n, max(T, 1), seed=1, side=ref["side"])
dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
dA, dR = ctx.to_device(mov["types"]), ctx.to_device(ref["types"])

def t(call, reps=6):
    out = []
    for _ in range(reps):
        ctx.check(L.same_timer_start(H), "t"); ctx.check(call(), "k")
        v = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(v)), "t"); out.append(v.value)
    return float(np.mean(out[1:]))

def build(buf, TT):
    return lambda: L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, TT, dax.ptr, drx.ptr, n, 0, n, 1.0, buf.ptr, n)

def rows(buf, r0, k):
    return buf.download((k, n), np.float64, offset_bytes=r0 * n * 8)

plain = ctx.alloc(n * n * 8)
p0, pT, pm = t(build(plain, 0)), t(build(plain, T)), t(lambda: L.same_dev_memset(H, plain.ptr, 0, n * n * 8))
probe_rows = [0, 1, n // 3, n // 2 + 7, n - 2]
ctx.check(build(plain, T)(), "k"); ctx.sync()
want = {r: rows(plain, r, 2) for r in probe_rows}          # T-type costs from the plain buffer
print(f"plain  hipMalloc @{plain.ptr:#x}: T=0 {p0:6.2f} ms  T={T} {pT:6.2f} ms  memset {pm:6.2f} ms", flush=True)
plain.free()

sp = ctx.alloc_spread(n * n * 8)
print("spread info:", json.dumps(sp.spread_info), flush=True)
s0, sT, sm = t(build(sp, 0)), t(build(sp, T)), t(lambda: L.same_dev_memset(H, sp.ptr, 0, n * n * 8))
print(f"spread         @{sp.ptr:#x}: T=0 {s0:6.2f} ms  T={T} {sT:6.2f} ms  memset {sm:6.2f} ms", flush=True)
ctx.check(build(sp, T)(), "k"); ctx.sync()
ok = all(np.array_equal(rows(sp, r, 2), want[r]) for r in probe_rows)
print("spread buffer reads back the plain buffer's values:", "ok" if ok else "MISMATCH", flush=True)

# a second live spread buffer must not alias the first (sized to what is left on the card)
m = 30000
sp2 = ctx.alloc_spread(m * m * 8)
print("second spread buffer:", json.dumps(sp2.spread_info), flush=True)
ctx.check(L.same_dev_memset(H, sp2.ptr, 0x5A, m * m * 8), "memset"); ctx.sync()
ok1 = all(np.array_equal(rows(sp, r, 2), want[r]) for r in probe_rows)
b = sp2.download((1 << 20,), np.uint8, offset_bytes=(m * m * 8) // 2)
print("first buffer untouched by writes to the second:", "ok" if ok1 else "MISMATCH", "| second holds its own bytes:", "ok" if (b == 0x5A).all() else "MISMATCH", flush=True)
sp.free(); sp2.free()

# after the frees: a new spread buffer and a new plain buffer, both written and read back
sp3 = ctx.alloc_spread(n * n * 8)
print("after free, new spread buffer:", json.dumps(sp3.spread_info), flush=True)
s3 = t(build(sp3, 0))
ctx.check(build(sp3, T)(), "k"); ctx.sync()
ok3 = all(np.array_equal(rows(sp3, r, 2), want[r]) for r in probe_rows)
pl2 = ctx.alloc(8 << 30)
ctx.check(L.same_dev_memset(H, pl2.ptr, 0x33, 8 << 30), "memset"); ctx.sync()
ok4 = all(np.array_equal(rows(sp3, r, 2), want[r]) for r in probe_rows) and (pl2.download((1 << 20,), np.uint8, offset_bytes=4 << 30) == 0x33).all()
print(f"new spread buffer: T=0 {s3:6.2f} ms; values {'ok' if ok3 else 'MISMATCH'}; after a plain allocation and memset beside it: {'ok' if ok4 else 'MISMATCH'}", flush=True)
