"""Board power / shader clock / energy per launch of the dense builds, one after another on one box:
fp64 column-resident kernel at T = 0, 8, 16, 20, fp32 at T = 20, the fixed-point control at T = 20.
Each variant loops alone for ~2.5 s while same_amd.telemetry samples sysfs (power1_input, freq1_input).
Usage: python tools/energy_table.py [n]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import _lib, ops, synth
from same_amd.telemetry import GpuTelemetry

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = _lib.Context(0)
L, H = ctx.lib, ctx.handle
tel = GpuTelemetry(ctx.pci_bus_id())
ld = (n + 1) & ~1
dD = ctx.alloc_spread(n * ld * 8)   # over the HBM regions (SAME_SPREAD=0: plain hipMalloc)
idle = None
if tel.available():
    tel.start(); time.sleep(1.0); idle = tel.stop()
    print(f"idle: {idle['power']['mean']:.0f} W")
print("| build | T | ms / launch | GB/s | frac of 8 TB/s | W (steady) | MHz (steady) | J / launch | J above idle |")
print("|---|---|---|---|---|---|---|---|---|")
for kind, T in (("f64", 0), ("f64", 8), ("f64", 16), ("f64", 20), ("f32", 20), ("q32->f64", 20), ("f64", 20)):
    ref = synth.make_cells(n, max(T, 1), seed=0); mov = synth.make_cells(n, max(T, 1), seed=1, side=ref["side"])
    dt = np.float32 if kind == "f32" else np.float64
    A, R = mov["types"][:, :T].astype(dt), ref["types"][:, :T].astype(dt)
    dA, dR = ctx.to_device(A), ctx.to_device(R)
    dax, drx = ctx.to_device(mov["xy"].astype(dt)), ctx.to_device(ref["xy"].astype(dt))
    if kind == "q32->f64":
        off, l2 = ops.quantize_types(A, R)
        dAq, dRq = ctx.alloc(A.size * 4), ctx.alloc(R.size * 4)
        ctx.check(L.same_quantize_u32_dev(H, dA.ptr, A.size, off, 2.0 ** l2, dAq.ptr), "q")
        ctx.check(L.same_quantize_u32_dev(H, dR.ptr, R.size, off, 2.0 ** l2, dRq.ptr), "q")
        call = lambda: L.same_dense_cost_q32_dev(H, dAq.ptr, dRq.ptr, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, 2.0 ** -l2, 1e-6, dD.ptr, ld)
        es = 8
    elif kind == "f32":
        call = lambda: L.same_dense_cost_f32_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, dD.ptr, (n + 3) & ~3)
        es = 4
    else:
        call = lambda: L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, dD.ptr, ld)
        es = 8
    for _ in range(3):
        ctx.check(call(), "warm")
    ctx.sync()
    ms = []
    if tel.available():
        tel.start()
    t_end = time.perf_counter() + 2.5
    while time.perf_counter() < t_end:
        ctx.check(L.same_timer_start(H), "t"); ctx.check(call(), "dense")
        v = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(v)), "t"); ms.append(v.value)
    t = tel.stop() if tel.available() else None
    m = float(np.mean(ms[len(ms) // 4:]))
    by = es * float(n) * n + es * (T + 2) * 2.0 * n
    if t and t.get("power_steady"):
        w, mhz = t["power_steady"]["mean"], t["sclk_steady"]["mean"]
        j = w * m * 1e-3
        ji = (w - idle["power"]["mean"]) * m * 1e-3 if idle else float("nan")
        print(f"| {kind} | {T} | {m:.2f} | {by / m / 1e6:.0f} | {by / m / 1e6 / 8000:.3f} | {w:.0f} | {mhz:.0f} | {j:.1f} | {ji:.1f} |", flush=True)
    else:
        print(f"| {kind} | {T} | {m:.2f} | {by / m / 1e6:.0f} | {by / m / 1e6 / 8000:.3f} | n/a | n/a | n/a | n/a |", flush=True)
