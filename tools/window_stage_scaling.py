"""a13 as a one-pass bin: what one window costs in same_window_stage as the SECTION grows.  The same interior 1200 x 1200 box
(~14 400 cells of each section at the bench's density) is staged against sections of 1M, 4M and 16M cells binned on the window grid
(every box a union of cells) and, for contrast, on cells so small that the box covers more than 64 of them (the mask over the whole
section that round 3 used for every window).  Usage: python tools/window_stage_scaling.py [cells ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import _lib, synth  # noqa: E402
from same_amd import windows as W  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 4_000_000, 16_000_000]
T, reps = 8, 200
ctx = _lib.default_context()
print(f"{'cells':>10} {'grid':>14} {'rows in box':>12} {'pairs':>8} {'us per stage call':>18} {'launches':>9} {'fills':>6} {'copies':>7} {'waits':>6}")
for n in sizes:
    ref = synth.make_cells(n, T, seed=0)
    mov = synth.make_cells(n, T, seed=1)
    rs, ms = W.Section(ref["xy"], ref["types"], None, None), W.Section(mov["xy"], mov["types"], None, None)
    box = (3000.0, 4200.0, 3000.0, 4200.0)
    for name, cell in (("window grid", 300.0), ("25-unit cells", 25.0)):
        dref, dmov = W.DeviceSection(rs, "float32", ctx).bin(0.0, 0.0, cell), W.DeviceSection(ms, "float32", ctx).bin(0.0, 0.0, cell)
        st = W.DeviceWindow(ctx)
        for _ in range(5):
            counts = st.stage(dmov, dref, box, 25, 8, 1.0)
        c0, t0 = ctx.stats(), time.perf_counter()
        for _ in range(reps):
            st.stage(dmov, dref, box, 25, 8, 1.0)
        dt, c1 = time.perf_counter() - t0, ctx.stats()
        per = {k: (c1[k] - c0[k]) / reps for k in c1}
        print(f"{n:>10} {name:>14} {counts[0]:>12} {counts[3]:>8} {dt / reps * 1e6:>18.1f} {per['launches']:>9.1f} {per['fills']:>6.1f} {per['copies']:>7.1f} {per['waits']:>6.1f}", flush=True)
        st.close()
        dref.close()
        dmov.close()
