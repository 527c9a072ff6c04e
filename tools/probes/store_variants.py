"""Dev probe: time the dense cost kernel (100k x 100k fp64) for each store cache-policy variant built by store_variants.sh."""
import ctypes, glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if len(sys.argv) > 1:   # child: one library, one process (a process can hold only one copy of the library's kernels cleanly)
    sys.path.insert(0, ROOT)
    import numpy as np
    from same_amd import _lib, synth
    _lib.LIB_PATH = sys.argv[1]
    n = 100000
    ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
    dD = ctx.alloc(n * n * 8)
    for T in (0, 20):
        ref = synth.make_cells(n, max(T, 1), seed=0); mov = synth.make_cells(n, max(T, 1), seed=1, side=ref["side"])
        dA, dR = ctx.to_device(np.ascontiguousarray(mov["types"][:, :T])), ctx.to_device(np.ascontiguousarray(ref["types"][:, :T]))
        dax, drx = ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])
        ms_all = []
        for it in range(7):
            ctx.check(L.same_timer_start(H), "t")
            ctx.check(L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n, 0, n, 1.0, dD.ptr, n), "dense")
            ms = ctypes.c_float(0); ctx.check(L.same_timer_stop(H, ctypes.byref(ms)), "t"); ms_all.append(ms.value)
        print(f"{os.path.basename(sys.argv[1]):32s} T={T:2d} best {min(ms_all[1:]):7.3f} ms mean {np.mean(ms_all[2:]):7.3f} ms", flush=True)
else:
    for so in sorted(glob.glob(os.path.join(HERE, "build", "libsame_hip_*.so"))):
        subprocess.run([sys.executable, __file__, so], check=True)
