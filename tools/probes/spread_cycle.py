"""Dev probe: allocate / write / check / free spread buffers of changing sizes many times in one process (address ranges are
never reused, chunks go back to the driver every time).  Usage: python tools/probes/spread_cycle.py [cycles]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from same_amd import _lib

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _lib.Context(0); L, H = ctx.lib, ctx.handle
hbm_free = lambda: ctx.mem_free() / 2**30
rng = np.random.default_rng(0)
seen = set()
keep = None
for c in range(cycles):
    gib = int(rng.integers(6, 90))
    t = time.perf_counter()
    b = ctx.alloc_spread(gib << 30)
    dt = time.perf_counter() - t
    assert b.spread_info["spread"], b.spread_info
    assert b.ptr not in seen, "an address range was handed out twice"
    seen.add(b.ptr)
    val = 0x10 + c
    ctx.check(L.same_dev_memset(H, b.ptr, val, gib << 30), "memset"); ctx.sync()
    for off in (0, (gib << 29) + 12345, (gib << 30) - (1 << 20)):
        assert (b.download((1 << 20,), np.uint8, offset_bytes=off & ~0xFFF) == val).all(), (c, off)
    if keep is not None:   # the buffer kept from the previous cycle still holds its own bytes
        kb, kval, kgib = keep
        assert (kb.download((1 << 20,), np.uint8, offset_bytes=(kgib << 29)) == kval).all(), c
        kb.free()
    print(f"cycle {c}: free after alloc {hbm_free():.1f} GiB; {gib} GiB {b.spread_info['per_region']} straddling {b.spread_info['straddling']} examined {b.spread_info['examined']} in {dt:.2f} s @{b.ptr:#x}", flush=True)
    keep = (b, val, gib)
keep[0].free()
print(f"all freed: {hbm_free():.1f} GiB free")
plain = ctx.alloc(64 << 30)
ctx.check(L.same_dev_memset(H, plain.ptr, 0x77, 64 << 30), "memset"); ctx.sync()
assert (plain.download((1 << 20,), np.uint8, offset_bytes=32 << 30) == 0x77).all()
print("ok: plain 64 GiB allocation after all frees works")
