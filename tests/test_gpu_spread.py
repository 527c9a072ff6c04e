"""same_dev_alloc_spread (csrc/spread.hip): a buffer mapped from 1 GiB chunks over the card's HBM regions must behave exactly
like a same_dev_alloc buffer -- same values from the same kernels, no aliasing between live buffers, clean after a free."""
import ctypes

import numpy as np
import pytest

from same_amd import _lib, synth

pytestmark = pytest.mark.gpu

N = 32768            # 32768 x 32768 doubles = 8 GiB exactly: eight chunks


@pytest.fixture(scope="module")
def ctx():
    return _lib.Context(0)


def _inputs(ctx, T):
    ref = synth.make_cells(N, T, seed=3)
    mov = synth.make_cells(N, T, seed=4, side=ref["side"])
    return [ctx.to_device(mov["types"]), ctx.to_device(ref["types"]), ctx.to_device(mov["xy"]), ctx.to_device(ref["xy"])]


def _build(ctx, d, T, buf):
    ctx.check(ctx.lib.same_dense_cost_f64_dev(ctx.handle, d[0].ptr, d[1].ptr, T, d[2].ptr, d[3].ptr, N, 0, N, 1.0, buf.ptr, N), "dense")
    ctx.sync()


def _bands(buf):
    return [buf.download((32, N), np.float64, offset_bytes=r0 * N * 8) for r0 in (0, 8191, 16384, 24000, N - 32)]


def test_small_request_is_a_plain_allocation(ctx):
    b = ctx.alloc_spread(1 << 20)
    assert b.spread_info["spread"] is False and b.ptr
    b.upload(np.arange(1 << 17, dtype=np.float64))
    assert np.array_equal(b.download((1 << 17,), np.float64), np.arange(1 << 17, dtype=np.float64))
    b.free()


def test_spread_buffer_holds_what_a_plain_one_holds(ctx):
    T = 5
    d = _inputs(ctx, T)
    plain = ctx.alloc(N * N * 8)
    _build(ctx, d, T, plain)
    want = _bands(plain)
    plain.free()
    sp = ctx.alloc_spread(N * N * 8)
    si = sp.spread_info
    assert si["spread"] is True and si["chunks_gib"] == 8
    assert sum(si["per_region"]) + si["straddling"] == 8 and si["examined"] >= 8
    # no region above half (within one chunk) -- unless the card had nothing else within the look-ahead
    assert max(si["per_region"]) <= 5 or si["examined"] >= 8 + 128, si
    _build(ctx, d, T, sp)
    got = _bands(sp)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    # a second live spread buffer does not alias the first; a buffer taken after a free starts from its own memory
    sp2 = ctx.alloc_spread(6 << 30)
    ctx.check(ctx.lib.same_dev_memset(ctx.handle, sp2.ptr, 0x5A, 6 << 30), "memset")
    ctx.sync()
    assert all(np.array_equal(g, w) for g, w in zip(_bands(sp), want))
    assert (sp2.download((1 << 20,), np.uint8, offset_bytes=3 << 30) == 0x5A).all()
    old = sp.ptr
    sp.free()
    sp3 = ctx.alloc_spread(N * N * 8)
    assert sp3.spread_info["spread"] is True and sp3.ptr != old      # an address is never mapped twice (spread.hip header)
    _build(ctx, d, T, sp3)
    assert all(np.array_equal(g, w) for g, w in zip(_bands(sp3), want))
    assert (sp2.download((1 << 20,), np.uint8, offset_bytes=3 << 30) == 0x5A).all()
    sp2.free()
    sp3.free()
    for b in d:
        b.free()


def test_memory_returns_and_addresses_are_not_reused(ctx):
    """Six allocate / write / check / free cycles: the card's free memory is back after every free (surplus chunks of the
    labelling included), no address is handed out twice, and a buffer kept across a cycle keeps its bytes."""
    ctx.sync()
    base = ctx.mem_free()
    seen, kept = set(), None
    for c, gib in enumerate((8, 11, 6, 16, 9, 7)):
        b = ctx.alloc_spread(gib << 30)
        assert b.spread_info["spread"] is True and b.ptr not in seen
        seen.add(b.ptr)
        ctx.check(ctx.lib.same_dev_memset(ctx.handle, b.ptr, 0x20 + c, gib << 30), "memset")
        ctx.sync()
        for off in (0, gib << 29, (gib << 30) - (1 << 20)):
            assert (b.download((1 << 20,), np.uint8, offset_bytes=off) == 0x20 + c).all()
        held = (gib << 30) + (kept[2] << 30 if kept else 0)
        assert abs((base - ctx.mem_free()) - held) < (1 << 30), (c, base - ctx.mem_free(), held)   # nothing but the live buffers is charged
        if kept:
            assert (kept[0].download((1 << 20,), np.uint8, offset_bytes=kept[2] << 29) == kept[1]).all()
            kept[0].free()
        kept = (b, 0x20 + c, gib)
    kept[0].free()
    assert abs(base - ctx.mem_free()) < (1 << 30)


def test_opt_out_by_environment(ctx, monkeypatch):
    monkeypatch.setenv("SAME_SPREAD", "0")
    b = ctx.alloc_spread(6 << 30)
    assert b.spread_info["spread"] is False
    b.free()


def test_too_large_a_request_fails_cleanly(ctx):
    p = ctypes.c_void_p()
    rc = ctx.lib.same_dev_alloc_spread(ctx.handle, 1 << 46, ctypes.byref(p), None)
    assert rc in (-12, -5) and not p.value
    ok = ctx.alloc(1 << 20)        # the context is still usable
    ok.free()
