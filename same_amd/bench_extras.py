"""bench.py's measurements AFTER the timed region (rank 0, N = 1): store ceilings of the cost block, the pruned path and the sweeps
alone on their stream, the realistic-matching pass, the operating point (board power, held clock) with the dense kernel looped, and the
T sweep of the dense kernel on the block re-taken spread over the HBM regions.  None of it is in `value`; it fills `roofline`'s
ceilings / telemetry / sweep fields.  Kept out of bench.py so that the driver's contract (step, timing, line) reads on its own."""
import ctypes
import os
import time

import numpy as np


def measure_extras(args, env, prob, headline_buffer):
    """-> extras dict (ceilings, pruned_path, triangle_maps_and_sweeps, realistic_matching, telemetry, sweep).  May re-take the cost
    block spread over the HBM regions (prob.respread()): read prob.dD afterwards."""
    from scipy.spatial import Delaunay

    from . import _lib, synth
    from .bench_common import HBM_PEAK_GBS, note
    from .bench_problem import dense_kernel_label
    from .telemetry import GpuTelemetry

    group, ctx, tctx = env.group, env.ctx, env.tctx
    L, H, chk = ctx.lib, ctx.handle, ctx.check
    n_ref, T, k, radius, rows, n_mov, Tr, ld = prob.n_ref, prob.T, prob.k, prob.radius, prob.rows, prob.n_mov, prob.Tr, prob.ld
    dD, dA, dR, dax, drx = prob.dD, prob.dA, prob.dR, prob.dax, prob.drx
    mov, ref, use_q32 = prob.mov, prob.ref, prob.use_q32
    extras = {}
    note(group, "ceilings, telemetry window, T sweep (after the timed region)")

    def timed_ms(call, what, reps=5):
        out = []
        for _ in range(reps + 1):
            chk(L.same_timer_start(H), "timer")
            chk(call(), what)
            ms = ctypes.c_float(0)
            chk(L.same_timer_stop(H, ctypes.byref(ms)), "timer")
            out.append(ms.value)
        return float(np.mean(out[1:])) * 1e-3   # first launch of a new shape is a warm-up

    t_store_only = timed_ms(lambda: L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, 0, dax.ptr, drx.ptr, n_ref, 0, rows, 1.0, dD.ptr, ld),
                            "dense T=0")
    t_memset = timed_ms(lambda: L.same_dev_memset(H, dD.ptr, 0, rows * ld * 8), "memset")
    # device copy of half the block onto the other half: reads N bytes and writes N bytes, 2N bytes of HBM traffic
    half = (rows * ld * 8 // 2) & ~0xFFF
    t_copy = timed_ms(lambda: L.same_d2d(H, dD.ptr + half, dD.ptr, half), "d2d copy") if half > 0 else None
    extras["ceilings"] = {"same_kernel_T0_store_only_GBs": 8.0 * n_ref * rows / t_store_only / 1e9,
                          "hipMemsetAsync_GBs": 8.0 * ld * rows / t_memset / 1e9,
                          "device_copy_GBs": (2.0 * half / t_copy / 1e9) if t_copy else None,
                          "device_copy_means": f"hipMemcpyAsync device-to-device of {half / 1e9:.1f} GB inside the cost block; read + "
                                               f"written bytes over its time",
                          "measured": "after the timed loop, warm chip, mean of 5 launches each"}
    # the same two stores into a plain hipMalloc buffer of this process, when the card has room for a second block
    if dD.spread_info and dD.spread_info["spread"]:
        try:
            plain = ctx.alloc(rows * ld * 8)
        except _lib.SameHipError:
            plain = None
        if plain is not None:
            t_p = timed_ms(lambda: L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, 0, dax.ptr, drx.ptr, n_ref, 0, rows, 1.0, plain.ptr, ld),
                           "dense T=0 (plain)")
            t_pm = timed_ms(lambda: L.same_dev_memset(H, plain.ptr, 0, rows * ld * 8), "memset (plain)")
            t_pT = None if use_q32 else timed_ms(lambda: L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n_ref, 0, rows,
                                                                      1.0, plain.ptr, ld), "dense (plain)")
            extras["ceilings"]["plain_hipMalloc_buffer"] = {
                "same_kernel_T0_store_only_GBs": 8.0 * n_ref * rows / t_p / 1e9, "hipMemsetAsync_GBs": 8.0 * ld * rows / t_pm / 1e9,
                "bench_kernel_ms": None if t_pT is None else t_pT * 1e3,
                "what": "one hipMalloc of the same size in this process: its rate depends on which HBM regions the driver drew it from"}
            plain.free()
    # the pruned path on its own (SURVEY 8d "fused KNN + cost, no dense store": reported as cell-pairs/s): indexed prune + costs of the
    # padded candidate lists for all rows of the block, HIP events on the tail stream
    def tail_ms(call, reps=5):
        out = []
        for _ in range(reps + 1):
            chk(L.same_timer_start(env.TH), "timer")
            call()
            ms = ctypes.c_float(0)
            chk(L.same_timer_stop(env.TH, ctypes.byref(ms)), "timer")
            out.append(ms.value)
        return float(np.mean(out[1:])) * 1e-3

    ctx.sync()
    t_pc = tail_ms(prob.prune_and_costs)
    t_sw = tail_ms(lambda: (prob.tri_maps(), prob.local_sweeps()))
    pc_bytes = 8.0 * (T + 2) * (n_ref + rows) + 16.0 * k * rows
    extras["pruned_path"] = {"ms": t_pc * 1e3, "cell_pairs_per_s": float(n_ref) * rows / t_pc, "algorithmic_bytes": pc_bytes,
                             "GBs": pc_bytes / t_pc / 1e9,
                             "what": f"same_knn_prune_indexed_dev (r={radius:g}, k={k}, caller-held grid index) + same_padded_cost_f64_dev "
                                     f"for {rows} aligned rows "
                                     f"against {n_ref} refs, alone on its stream: dense-equivalent pairs covered per second without "
                                     f"materialising the matrix "
                                     "(latency / gather-bound, no roofline fraction claimed)"}
    # SURVEY 8d / DESIGN 5 per-unit figures
    sw_bytes = Tr * (74 + (12 + 3 * 40 + 1) + (12 + 12 + 96 + 16 + 4) + (12 + 48 + 12 + 1 + 16) + (12 + 48 + 24 + 1 + 8)) + n_mov
    extras["triangle_maps_and_sweeps"] = {"ms": t_sw * 1e3, "triangles": Tr, "triangles_per_s": Tr / t_sw, "touched_bytes": sw_bytes,
                                          "GBs": sw_bytes / t_sw / 1e9,
                                          "what": "classify + weights / signs + XY-order sweep + area flips + orientation sweep incl. its "
                                                  "read-back, alone on its stream"}
    # SURVEY 8d's second input variant: moving = refs + N(0, 2^2) jitter with 5 % of the rows dropped, matched by the greedy MIP
    # start (src/init_helpers.py:104-133) -- a realistic matching, so the sweeps see few, local flips instead of the dense
    # disorder of two independent sections.  Through the host-buffer entry points (PCIe included), one pass, untimed region.
    from same_amd import ops as _ops

    jm = synth.make_jittered(ref, seed=1)
    jt = np.ascontiguousarray(Delaunay(jm["xy"]).simplices, dtype=np.int32)
    j0 = time.perf_counter()
    jidx, _, _ = _ops.knn_prune(jm["xy"], ref["xy"], radius, k, want_d2=False, ctx=tctx)
    jr, jc = np.nonzero(jidx >= 0)
    jpairs = np.column_stack((jr, jidx[jr, jc])).astype(np.int32)
    jcost = _ops.pair_cost(jm["types"], ref["types"], jm["xy"], ref["xy"], jpairs, 1.0, ctx=tctx)
    jwants = _ops.pair_rowmin(jpairs, jcost, len(jm["xy"]), ctx=tctx) < 100.0
    jpor, jrounds = _ops.greedy_match(jpairs, jcost, len(jm["xy"]), n_ref, jwants, ctx=tctx)
    j1 = time.perf_counter()
    jmatch = np.full(len(jm["xy"]), -1, np.int32)
    jai = np.flatnonzero(jpor >= 0)
    jmatch[jai] = jpairs[jpor[jai], 1]
    jsign, _ = _ops.tri_sign_weight(jm["xy"], jm["size"], jt, ctx=tctx)
    jsw = _ops.BoundSweep(jt, jsign, ref["xy"], len(jm["xy"]), ctx=tctx)
    j2 = time.perf_counter()
    jchecked, jviol = jsw.sweep_match(jmatch)
    j3 = time.perf_counter()
    jsw.close()
    _je, _jtf, jpf, jcounts = _ops.xyorder_sweep(jm["xy"], ref["xy"], jt, jmatch, ctx=tctx)
    _jb, _ja, _jm3, jflip = _ops.area_flip(jm["xy"], ref["xy"], jt, jmatch, ctx=tctx)
    j4 = time.perf_counter()
    extras["realistic_matching"] = {
        "what": f"moving = refs + N(0, 2^2) jitter, 5 % of rows dropped ({len(jm['xy'])} aligned cells, {len(jt)} triangles); prune "
                f"(r={radius:g}, k={k}) + "
                "fp64 pair costs + greedy MIP start on the device, then the three sweeps under that matching; host-buffer entry points, "
                "one pass",
        "pairs": int(len(jpairs)), "matched_rows": int(len(jai)), "greedy_rounds": int(jrounds), "prune_costs_start_ms": (j1 - j0) * 1e3,
        "orientation_sweep_ms": (j3 - j2) * 1e3, "orientation_checked": int(jchecked), "orientation_flipped": int(len(jviol)),
        "xyorder_and_area_ms": (j4 - j3) * 1e3, "xy_comparisons": int(jcounts[0]), "xy_violations": int(jcounts[1]),
        "points_with_violations": int(np.count_nonzero(jpf)), "area_flips": int(np.count_nonzero(jflip))}
    # operating point: loop the dense kernel alone for a few seconds while a side thread reads board power and shader clock
    tel = GpuTelemetry(ctx.pci_bus_id())
    if tel.available():
        loop_ms = []
        tel.start()
        # six seconds at the metric's size: long enough for steady-state means, and for an outside sampler with a 5 s period
        # (the driver's gpu_busy probe) to see the card busy at least once; a second for the small test workloads
        t_end = time.perf_counter() + float(os.environ.get("SAME_BENCH_TELEMETRY_S", "6.0" if float(n_ref) * rows >= 1e9 else "1.0"))
        while time.perf_counter() < t_end:
            prob.dense_all(timed=loop_ms)
            prob.dense_time(loop_ms)
        tele = tel.stop()
        tele["dense_ms_during_window"] = float(np.mean([m for m, _ in loop_ms]))
        tele["what"] = (f"dense kernel (T={T}, fp64) looped alone for the window ({tele.get('window_s', 0):.1f} s); sysfs read every "
                        f"{tel.period * 1e3:.0f} ms by a side thread")
    else:
        tele = {"available": False, "reason": f"no readable power/clock nodes under {tel.dev_dir}"}
    extras["telemetry"] = tele
    # the same measurement at the type counts of the reference's real datasets (examples/*/run_same.sh: T = 3, 5, 8).  Those shapes are
    # bound by the store stream, which is where placement over the HBM regions pays: the block is re-taken spread for the sweep
    # (--spread off keeps the plain one), after everything above was measured on the block the timed loop used
    if args.spread != "off":
        dD = prob.respread()
    sweep_buffer = "spread over the HBM regions" if (dD.spread_info and dD.spread_info["spread"]) else "plain hipMalloc"
    if not headline_buffer.get("spread") and dD.spread_info and dD.spread_info["spread"]:
        t_sp = timed_ms(lambda: L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, 0, dax.ptr, drx.ptr, n_ref, 0, rows, 1.0, dD.ptr, ld),
                        "dense T=0 (spread)")
        extras["ceilings"]["spread_buffer"] = {"same_kernel_T0_store_only_GBs": 8.0 * n_ref * rows / t_sp / 1e9, "info": dD.spread_info,
                                               "what": "the block re-taken through same_dev_alloc_spread for the T sweep below (after the "
                                                       "timed region)"}
    sweep_rows = []
    for dt_name, T_s in (("f64", 3), ("f64", 5), ("f64", 8), ("f64", 16), ("f64", 20), ("f32", 20)):
        npdt = np.float64 if dt_name == "f64" else np.float32
        es = np.dtype(npdt).itemsize
        r_s, m_s = synth.make_cells(n_ref, T_s, seed=0), synth.make_cells(rows, T_s, seed=1, side=ref["side"])
        bufs = [ctx.to_device(m_s["types"].astype(npdt)), ctx.to_device(r_s["types"].astype(npdt)),
                ctx.to_device(m_s["xy"].astype(npdt)), ctx.to_device(r_s["xy"].astype(npdt))]
        fn = L.same_dense_cost_f64_dev if dt_name == "f64" else L.same_dense_cost_f32_dev
        ld_s = ld if dt_name == "f64" else (n_ref + 3) & ~3
        t_s = timed_ms(lambda: fn(H, bufs[0].ptr, bufs[1].ptr, T_s, bufs[2].ptr, bufs[3].ptr, n_ref, 0, rows, 1.0, dD.ptr, ld_s),
                       "dense sweep")
        by = es * float(n_ref) * rows + es * (T_s + 2) * (n_ref + rows)
        sweep_rows.append({"dtype": dt_name, "T": T_s, "kernel": dense_kernel_label(dt_name, T_s), "ms": t_s * 1e3,
                           "GBs": by / t_s / 1e9, "frac": by / t_s / 1e9 / HBM_PEAK_GBS, "output_buffer": sweep_buffer})
        for b in bufs:
            b.free()
    # control: the opt-in fixed-point build at the bench's own T -- the same 80 GB of stores with the 2T fp64 adds replaced
    # by T integer v_sad_u32 (exact sums on a 2^-s grid; NOT the reference's arithmetic, never the reported kernel)
    if T <= 32 and not use_q32:
        from same_amd import ops

        off, l2 = ops.quantize_types(mov["types"], ref["types"])
        cAq, cRq = ctx.alloc(mov["types"].size * 4), ctx.alloc(ref["types"].size * 4)
        chk(L.same_quantize_u32_dev(H, dA.ptr, mov["types"].size, off, 2.0 ** l2, cAq.ptr), "quantize")
        chk(L.same_quantize_u32_dev(H, dR.ptr, ref["types"].size, off, 2.0 ** l2, cRq.ptr), "quantize")
        t_q = timed_ms(lambda: L.same_dense_cost_q32_dev(H, cAq.ptr, cRq.ptr, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n_ref, 0, rows, 1.0,
                                                         2.0 ** -l2, 1e-6, dD.ptr, ld), "dense q32")
        by = 8.0 * n_ref * rows + (4.0 * T + 16.0) * (n_ref + rows)
        # its outputs against the bit-exact kernel's, on 16 sampled rows of this very run
        worst = 0.0
        probe_rows = np.random.default_rng(1).choice(rows, 16, replace=False)
        q_rows = {int(i): dD.download((n_ref,), np.float64, offset_bytes=int(i) * ld * 8) for i in probe_rows}
        chk(L.same_dense_cost_f64_dev(H, dA.ptr, dR.ptr, T, dax.ptr, drx.ptr, n_ref, 0, rows, 1.0, dD.ptr, ld), "dense")
        for i, qv in q_rows.items():
            ev = dD.download((n_ref,), np.float64, offset_bytes=i * ld * 8)
            worst = max(worst, float(np.max(np.abs(qv - ev) / ev)))
        sweep_rows.append({"dtype": "q32->f64", "T": T, "kernel": f"dense_cost_q32_kernel<{T}>", "ms": t_q * 1e3, "GBs": by / t_q / 1e9,
                           "frac": by / t_q / 1e9 / HBM_PEAK_GBS, "opt_in": True, "max_rel_diff_vs_exact_on_16_rows": worst,
                           "note": f"fixed-point control, NOT the reference's arithmetic and not the kernel this line reports: type sums "
                                   f"exact on a 2^-{l2} grid (|error| <= {T * 2.0 ** -l2:.2e} absolute), sums too small for the grid "
                                   "recomputed in fp64, so every output is within 1e-6 relative of the bit-exact build by construction"})
        cAq.free()
        cRq.free()
    extras["sweep"] = sweep_rows
    return extras
