"""Probe: one dense build (dtype, T) looped alone for ~2.5 s at whatever occupancy SAME_DENSE_LDS_PAD leaves it, with board
power / shader clock sampled from sysfs -- is the kernel bound by VALU issue (time grows as waves per SIMD shrink) or by
the board power cap (the clock rises as the VALU idles, time stays)?
Usage: SAME_DENSE_LDS_PAD=<bytes> python tools/probes/dense_occupancy.py f32|f64 T [n]
HISTORICAL: csrc/cost.hip had the SAME_DENSE_LDS_PAD switch only up to commit 2b8b3b4 (where profiles/r03_dense_occupancy.log was
taken); against a later library every setting measures the same kernel, so the probe refuses to run there."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if b"SAME_DENSE_LDS_PAD" not in open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "same_amd",
                                                  "libsame_hip.so"), "rb").read():
    sys.exit("this libsame_hip.so has no SAME_DENSE_LDS_PAD switch (removed after commit 2b8b3b4): the sweep would label one kernel four ways")
from same_amd import _lib, synth
from same_amd.telemetry import GpuTelemetry

kind, T = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
dt = np.float32 if kind == "f32" else np.float64
es = np.dtype(dt).itemsize
ctx = _lib.Context(0)
L, H = ctx.lib, ctx.handle
tel = GpuTelemetry(ctx.pci_bus_id())
ld = (n + 3) & ~3
dD = ctx.alloc_spread(n * ld * es)
ref = synth.make_cells(n, max(T, 1), seed=0); mov = synth.make_cells(n, max(T, 1), seed=1, side=ref["side"])
dA, dR = ctx.to_device(mov["types"][:, :T].astype(dt)), ctx.to_device(ref["types"][:, :T].astype(dt))
dax, drx = ctx.to_device(mov["xy"].astype(dt)), ctx.to_device(ref["xy"].astype(dt))
fn = L.same_dense_cost_f32_dev if kind == "f32" else L.same_dense_cost_f64_dev
call = lambda: fn(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, dD.ptr, ld)
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
t = tel.stop() if tel.available() else {}
m = float(np.mean(ms[len(ms) // 4:]))
w = (t.get("power_steady") or {}).get("mean"); mhz = (t.get("sclk_steady") or {}).get("mean")
pad = int(os.environ.get("SAME_DENSE_LDS_PAD", "0"))
floor = (2 * T + 4) * float(n) * n / (1024 * (32 if kind == "f32" else 16) * (mhz or 2400) * 1e6) * 1e3
print(f"{kind} T={T} lds_pad={pad:6d} (<= {160 * 1024 // pad if pad else 8} blocks/CU by LDS): {m:7.3f} ms  {w or float('nan'):6.0f} W  {mhz or float('nan'):5.0f} MHz  "
      f"VALU floor at that clock {floor:6.2f} ms -> busy {floor / m:.2f}", flush=True)
