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


def test_finished_range_is_verified_and_reported(ctx):
    """The finished mapping is checked, not trusted: one store over the whole range is timed against the same-region level and
    sampled neighbouring chunks are timed against each other as their labels say (spread.hip)."""
    b = ctx.alloc_spread(12 << 30)
    si = b.spread_info
    assert si["spread"] is True and si["final_store_gbps"] > 0 and si["pairs_checked"] >= 1
    assert si["pairs_as_labelled"] <= si["pairs_checked"]
    # the verdict is what the two measurements say (1.12 = spread.hip's LEVEL_RATIO; the reported rates are rounded to 1 GB/s) ...
    fast = si["final_store_gbps"] >= 1.12 * si["same_region_level_gbps"] + 1
    slow = si["final_store_gbps"] <= 1.12 * si["same_region_level_gbps"] - 1
    assert not (fast and si["pairs_as_labelled"] == si["pairs_checked"]) or si["verified"] is True, si
    assert not slow or si["verified"] is False, si
    # ... and a balanced choice of chunks stores faster than one region does (it normally verifies; a box where it does not is
    # reported by the flag, which is the point of having it, not failed here)
    if max(si["per_region"]) <= 7:
        assert si["final_store_gbps"] > si["same_region_level_gbps"], si
    assert isinstance(si["stopped_at_time_bound"], bool)
    b.free()


def test_time_bound_stops_the_search_not_the_allocation(ctx, monkeypatch):
    """SAME_SPREAD_MAX_SECONDS: past the bound the search for better-balanced chunks stops and what there is gets mapped (or,
    past twice the bound while still taking the buffer's own chunks, the plain allocation is used) -- either way a usable buffer."""
    monkeypatch.setenv("SAME_SPREAD_MAX_SECONDS", "0.05")
    b = ctx.alloc_spread(8 << 30)
    si = b.spread_info
    # what the bound changes is countable: the search for better-balanced chunks ends after a handful of extra chunks (taking and
    # labelling one is ~10 ms: 14 examined on this pool) instead of walking the 128-chunk look-ahead -- by the bound
    # (`stopped_at_time_bound`) or because the choice was balanced before it.  The call's own time is mostly giving chunks back to the
    # card and is the box's business (2-4 s on this pool): only a hang would fail it.
    assert b.ptr and (si["spread"] is False or (8 <= si["examined"] < 8 + 64 and si["seconds"] < 30.0)), si
    ctx.check(ctx.lib.same_dev_memset(ctx.handle, b.ptr, 0x11, 8 << 30), "memset")
    ctx.sync()
    assert (b.download((1 << 20,), np.uint8, offset_bytes=5 << 30) == 0x11).all()
    b.free()


def test_labelling_leaves_an_open_timer_alone(ctx):
    """The labelling stores are timed with events of their own: a same_timer_start .. same_timer_stop pair open around a spread
    allocation measures what the caller enqueued, not a labelling store (callers do allocate output blocks inside regions they time)."""
    ms = ctypes.c_float(0)
    ctx.check(ctx.lib.same_timer_start(ctx.handle), "timer")
    b = ctx.alloc_spread(8 << 30)
    assert b.spread_info["spread"] is True
    ctx.check(ctx.lib.same_timer_stop(ctx.handle, ctypes.byref(ms)), "timer")
    assert ms.value >= b.spread_info["seconds"] * 1e3 * 0.5      # the whole allocation lies inside the pair (a labelling store is ~1 ms)
    b.free()


def test_rccl_collectives_through_fresh_buffers_after_a_spread_free(ctx):
    """spread alloc -> free -> size-1 RCCL communicator -> all-gather + all-reduce through freshly allocated buffers, payload
    checked: the addresses a spread buffer used are never mapped again by this library, and whatever RCCL or hipMalloc place
    there afterwards must reach their own memory (the stale-mapping behaviour described in spread.hip's header)."""
    from same_amd.dist import RcclGroup

    c2 = _lib.Context(0)                         # own context: the communicator lives and dies with it
    sp = c2.alloc_spread(8 << 30)
    assert sp.spread_info["spread"] is True
    c2.check(c2.lib.same_dev_memset(c2.handle, sp.ptr, 0x77, 8 << 30), "memset")
    c2.sync()
    sp.free()
    comm = RcclGroup(c2, 1, 0, lambda b: b)
    try:
        info = comm.info()
        assert info["nranks"] == 1 and info["rank"] == 0 and info["device"] == 0 and info["version"] > 20000
        n = 1 << 22
        rng = np.random.default_rng(5)
        payload = rng.integers(0, 255, n, dtype=np.uint8)
        send, recv = c2.to_device(payload), c2.alloc(n)
        big = [c2.alloc(1 << 30) for _ in range(4)]        # fresh plain blocks, likely on recycled physical memory
        for i, blk in enumerate(big):
            c2.check(c2.lib.same_dev_memset(c2.handle, blk.ptr, 0x30 + i, 1 << 30), "memset")
        comm.allgather_dev(send, recv, n)
        comm.wait()
        c2.sync()
        assert np.array_equal(recv.download((n,), np.uint8), payload)
        ms, nb = comm.gather_time()
        assert nb == n and ms >= 0.0
        counters = c2.to_device(np.array([3, 9, 27], np.uint64))
        comm.allreduce_dev(counters, 3, _lib.DT_U64, _lib.OP_SUM)
        c2.sync()
        assert counters.download((3,), np.uint64).tolist() == [3, 9, 27]
        for i, blk in enumerate(big):
            assert (blk.download((1 << 16,), np.uint8, offset_bytes=(1 << 29) + i) == 0x30 + i).all()
        # and a second spread buffer beside the live communicator
        sp2 = c2.alloc_spread(6 << 30)
        c2.check(c2.lib.same_dev_memset(c2.handle, sp2.ptr, 0x42, 6 << 30), "memset")
        comm.allgather_dev_async(send, recv, n)
        comm.wait()
        c2.sync()
        assert np.array_equal(recv.download((n,), np.uint8), payload)
        assert (sp2.download((1 << 20,), np.uint8, offset_bytes=3 << 30) == 0x42).all()
        sp2.free()
    finally:
        comm.close()
    c2.close()
