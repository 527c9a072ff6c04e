import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, "/root/repo")
from same_amd import _lib, synth
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
for n in (3000, 6000, 12000, 25000, 50000, 100000):
    ref = synth.make_cells(n, 2, seed=0); mov = synth.make_cells(n, 2, seed=1, side=ref["side"])
    dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
    didx, dcnt = ctx.alloc(n * 32 * 4), ctx.alloc(n * 4)
    res = {}
    for mode in ("brute", "grid"):
        os.environ["SAME_KNN_MODE"] = mode
        for _ in range(3):
            ctx.check(L.same_knn_prune_dev(H, dax.ptr, drx.ptr, n, 0, n, 25.0, 32, didx.ptr, None, dcnt.ptr), "k")
        ctx.sync()
        t = time.perf_counter()
        for _ in range(20):
            ctx.check(L.same_knn_prune_dev(H, dax.ptr, drx.ptr, n, 0, n, 25.0, 32, didx.ptr, None, dcnt.ptr), "k")
        ctx.sync()
        res[mode] = (time.perf_counter() - t) / 20 * 1e3
    print(f"n={n:7d} brute {res['brute']:.3f} ms  grid {res['grid']:.3f} ms", flush=True)
