"""BASELINE cfg 5 as a bench step (bench.py --workload cfg5, and the `cfg5` record every bench line embeds): the reference's window
loop src/same.py:507-593 with whole windows dealt to the ranks.  Runs on the caller's host group, context and communicator -- no
process of its own -- so the same code measures one rank or eight."""
import os
import time

from .bench_common import HBM_PEAK_GBS, Env, baseline_metric, comm_report, note

RECORD_KEYS = ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "dtype", "windows_per_s", "aligned_cells_per_s", "aligned_cells_per_window",
               "windows_per_s_triangulations_given", "per_rank",
               "host_glue_share", "python_share", "qhull_wait_share", "serial_tail_s_per_step",
               "threads_per_rank", "runtime_calls_per_window", "qhull", "table_allgather", "merged_matches", "parity_spot_check", "rccl")


def record(line):
    """The sub-record a bench line of another workload carries as `cfg5`: the cfg 5 line's own numbers, without its prose."""
    rec = {k_: line.get(k_) for k_ in RECORD_KEYS if k_ in line}
    rec["workload"], rec["pipeline"] = line["config"]["workload"], line["config"]["pipeline"]
    rec["what"] = ("BASELINE cfg 5 (whole sliding windows dealt to the ranks, fp32 costs, all sweeps, tables exchanged once and merged) measured in "
                   "this job, on its ranks, contexts and communicator, after this line's own timed region; windows_per_s is the whole job's")
    return rec


def run(args, group, ctx, comm, transport, steps, warmup, cpu_baseline=None):
    """BASELINE cfg 5: a section of --cfg5-cells cells tiled into overlapping windows (src/same.py:481-488), the windows dealt
    round-robin (heaviest first) to the ranks, every window through the whole pre-MIP path with fp32 costs and all three sweeps,
    with both sections resident on the device (same_amd.windows.iter_device_windows over csrc/window.hip: the host triangulates and receives
    the match; --cfg5-pipeline columns is the host-buffer form it is tested against),
    every rank's central-trimmed match table exchanged in ONE device all-gather (dist.allgather_table) and merged
    (src/helpers.py:692-815, de-duplication on the GPU).  There is no solver on the GPU box: the incumbent whose violations are swept is the greedy MIP start
    (src/init_helpers.py:109-133), which is what the reference hands Gurobi as its first incumbent.
    `ctx` is the context the communicator lives on (the exchange runs on its stream); worker threads get contexts of their own.
    One step = the whole plan once (every rank its share) + the exchange + the merge (on rank 0, which owns the result).
    cpu_baseline: None, or the caller's function (state dict) -> (cpu_baseline record, parity text) run on rank 0 at world 1.
    -> the line as a dict on rank 0, None elsewhere; nothing is closed here.  value = dense-equivalent cell pairs
    (sum over windows of aligned x ref cells in the window) per second; `windows_per_s` per rank and the share of the step
    spent outside libsame_hip calls (`host_glue_share`) come from the stage markers of same_amd/_trace.py."""
    import numpy as np
    import pandas as pd

    import same_amd
    from same_amd import _lib, _trace, ops, synth
    from same_amd.merge import merge_window_matches_unique_ref
    from same_amd.dist import allgather_table, last_table_gather
    from same_amd import windows as W
    from same_amd.windows import DeviceSection, Section, assign_windows, iter_device_windows, iter_window_arrays, window_plan

    _trace.enable(True)
    _lib.instrument()
    n, T = int(args.cfg5_cells), 8
    ref = synth.make_cells(n, T, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
    cols = synth.type_columns(T)
    op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32")
    plan = window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
    mine = assign_windows(plan, group.world)[group.rank]
    my_plan = [plan[q] for q in mine]
    note(group, f"cfg5: {n} cells, {len(plan)} windows of ~{int(np.mean([w['n_mov'] for w in plan]))} aligned cells; this rank runs {len(my_plan)}")

    TABLE_COLUMNS = (("Aligned_Cell_Num_Old", np.int64), ("Ref_Cell_Num_Old", np.int64), ("X", np.float64), ("Y", np.float64),
                     ("filtered_violation", bool), ("window_id", np.int64))
    ref_sec, mov_sec = Section.from_frame(r_df, cols), Section.from_frame(m_df, cols)
    ref_ids, mov_ids = r_df["Cell_Num_Old"].to_numpy(), m_df["Cell_Num_Old"].to_numpy()
    on_device = args.cfg5_pipeline == "device"
    dref, dmov = (DeviceSection(ref_sec, np.float32, ctx), DeviceSection(mov_sec, np.float32, ctx)) if on_device else (None, None)
    if on_device:       # the rows of both sections binned ONCE on the window grid: every window box is then a union of cells
        xs, ys, _ = W.window_grid(ref["xy"], mov["xy"], 1200, 300)
        cell_grid = W.window_cell_grid((xs, ys), 1200, 300)
        dref.bin(*cell_grid)
        dmov.bin(*cell_grid)
    path_kw = dict(radius=25, knn=8, dist_ct_coeff=1.0, min_angle_deg=15, ignore_same_type_triangles=True)

    def device_table(dw):
        """the window's central match table from what iter_device_windows leaves on the host (section rows, XY, match, flags)"""
        w = dw.window
        x, y = dw.axy[:, 0], dw.axy[:, 1]
        tx0, tx1, ty0, ty1 = w["trim"]                                  # central region (src/same.py:566-581), matched cells only
        c = np.flatnonzero((dw.match_row >= 0) & (x >= tx0) & (x < tx1) & (y >= ty0) & (y < ty1))
        tab = {"Aligned_Cell_Num_Old": mov_ids[dw.rows_m[c]], "Ref_Cell_Num_Old": ref_ids[dw.match_row[c]], "X": x[c], "Y": y[c],
               "filtered_violation": dw.point_flag[c].astype(bool), "window_id": np.full(len(c), w["window_id"], np.int64)}
        st = dw.stats
        return tab, {"pairs": dw.counts[3], "triangles": dw.n_triangles, "checked": st["checked"], "flipped": st["flipped"],
                     "xy_violations": st["xy_violations"], "area_flips": st["area_flips"]}

    def run_window(wa, wctx):
        """greedy incumbent -> orientation sweep (lazy-constraint body), XY-order sweep, area flips -> the window's central match table"""
        w, pairs = wa.window, wa.pairs.astype(np.int32)
        # greedy MIP start (src/init_helpers.py:104-133) in its flat device form: per-row minimum, rows that beat their
        # no-match penalty, the scan's matching -> one pair index per aligned row
        wants = ops.pair_rowmin(pairs, wa.costs, wa.n_aligned, ctx=wctx) < 100.0 * wa.size.astype(float)
        pair_of_row, _rounds = ops.greedy_match(pairs, wa.costs, wa.n_aligned, wa.n_ref, wants, ctx=wctx)
        ai = np.flatnonzero(pair_of_row >= 0)
        ri = pairs[pair_of_row[ai], 1].astype(np.int64)
        match = np.full(wa.n_aligned, -1, np.int32)
        match[ai] = ri
        sw = ops.BoundSweep(wa.triangles, wa.signs, wa.rxy, wa.n_aligned, ctx=wctx)      # the lazy-constraint body (src/same.py:645-669)
        checked, viol = sw.sweep_match(match)
        sw.close()
        # XY-order sweep (src/violationhelper.py:53-117) and signed-area flips (src/same.py:1362-1402) in their flat device forms
        _edge, _tflag, pflag, counts = ops.xyorder_sweep(wa.axy, wa.rxy, wa.triangles, match, ctx=wctx)
        _before, _after, _m3, flipped = ops.area_flip(wa.axy, wa.rxy, wa.triangles, match, ctx=wctx)
        x, y = wa.axy[ai, 0], wa.axy[ai, 1]
        tx0, tx1, ty0, ty1 = w["trim"]                                  # central region (src/same.py:566-581)
        c = np.flatnonzero((x >= tx0) & (x < tx1) & (y >= ty0) & (y < ty1))
        tab = {"Aligned_Cell_Num_Old": mov_ids[wa.rows_m[ai[c]]], "Ref_Cell_Num_Old": ref_ids[wa.rows_r[ri[c]]], "X": x[c], "Y": y[c],
               "filtered_violation": pflag[ai[c]].astype(bool), "window_id": np.full(len(c), w["window_id"], np.int64)}
        return tab, {"pairs": len(pairs), "triangles": len(wa.triangles), "checked": int(checked), "flipped": len(viol),
                     "xy_violations": int(counts[1]), "area_flips": int(np.count_nonzero(flipped))}

    # The windows of a pass are independent and the host work per window (numpy index work, ~7 ms) dwarfs its kernels (~0.3 ms), so
    # the rank walks its windows with --cfg5-threads workers, each with a context (= stream) of its own; numpy and the library calls
    # release the interpreter lock.  Results are put back into plan order, so the tables do not depend on the thread count.
    from same_amd import qhull_pool as _share

    cpu_share = _share.cpu_budget() / _share.cpu_sharers()[0]          # this rank's part of the CPUs it may use
    default_threads = (2 if cpu_share >= 8 else 1) if on_device else 4  # a second Python thread only pays where there are CPUs to feed it
    n_workers = max(1, int(args.cfg5_threads if args.cfg5_threads is not None else default_threads))
    worker_ctx = [ctx] + [_lib.Context(ctx.device) for _ in range(n_workers - 1)]

    tri_cache = [None]        # set for the diagnostic pass after the timed loop (triangulations remembered: Qhull out of the picture)

    def walk(windows, wctx, out):
        if on_device:
            for dw in iter_device_windows(ref_sec, mov_sec, dref, dmov, windows, no_match_penalty=100.0, ctx=wctx, triangulator=tri_cache[0], **path_kw):
                if dw.error is None:
                    with _trace.stage("table (bench step)"):
                        out.append((dw.window["window_id"], *device_table(dw)))
            return
        for wa in iter_window_arrays(ref_sec, mov_sec, windows, cost_dtype=np.float32, ctx=wctx, **path_kw):
            if wa.error is not None:              # a window whose prune leaves no pairs (src/same.py:1003)
                continue
            with _trace.stage("incumbent + sweeps + table (bench step)"):
                out.append((wa.window["window_id"], *run_window(wa, wctx)))

    def one_pass(windows):
        import threading

        outs = [[] for _ in range(n_workers)]
        if n_workers == 1:
            walk(windows, ctx, outs[0])
        else:
            errors = []

            def guarded(q):
                try:
                    walk(windows[q::n_workers], worker_ctx[q], outs[q])
                except BaseException as e:   # noqa: BLE001 -- re-raised in the main thread below
                    errors.append(e)

            threads = [threading.Thread(target=guarded, args=(q,)) for q in range(n_workers)]
            [t.start() for t in threads]
            [t.join() for t in threads]
            if errors:
                raise errors[0]
        pos = {w["window_id"]: q for q, w in enumerate(windows)}
        done = sorted((r for part in outs for r in part), key=lambda r: pos[r[0]])
        return [r[1] for r in done], [r[2] for r in done]

    def all_ranks(fn, *a):
        """fn(*a) on this rank; if ANY rank raised, every rank raises (so that no rank walks into a collective the others never reach)."""
        err, out = None, None
        try:
            out = fn(*a)
        except Exception as e:  # noqa: BLE001 -- re-raised below, on every rank
            err = e
        if group.world > 1 and group.min(0.0 if err is not None else 1.0) < 1.0:
            raise RuntimeError(f"cfg5: a rank failed in its window pass ({type(err).__name__ if err else 'another rank'}: {err})")
        if err is not None:
            raise err
        return out

    pass_seconds = [0.0]      # wall time of the window passes inside the timed loop (the threads' capacity is this x threads + the serial rest)

    def step():
        t_pass = time.perf_counter()
        tabs, stats = all_ranks(one_pass, my_plan)
        pass_seconds[0] += time.perf_counter() - t_pass
        mine_tab = {c: (np.concatenate([t[c] for t in tabs]) if tabs else np.zeros(0, dt)) for c, dt in TABLE_COLUMNS}
        mine_tab["filtered_violation"] = mine_tab["filtered_violation"].astype(np.uint8)
        with _trace.stage("table exchange (all-gather)"):
            every = allgather_table(ctx, comm, group, mine_tab)          # the ONE exchange: one table per rank, a device all-gather
            if comm is not None:
                ms, nbytes = last_table_gather()
                exchange_ms.append(ms)
                exchange_bytes[0] = nbytes
        merged = None
        if group.rank == 0:                                              # the merged table is the job's result: rank 0 holds it
            with _trace.stage("merge (device de-duplication + host matching)"):
                frames = [pd.DataFrame(dict(t, filtered_violation=t["filtered_violation"].astype(bool))) for t in every if len(t["X"])]
                merged = merge_window_matches_unique_ref(frames)
        return merged, stats

    from same_amd import qhull_pool as _qp

    exchange_ms, exchange_bytes = [], [0]

    # warm-up: scratch slots, Qhull helpers, first-launch costs -- and, on the device path, every window state a worker keeps
    # in flight gets its buffers (they stay with the context afterwards: the timed passes allocate nothing)
    n_warm = (_qp.lookahead() + 1) * n_workers if on_device else 2
    for _ in range(warmup):
        all_ranks(one_pass, my_plan[: max(1, min(n_warm, len(my_plan)))])
    group.barrier()
    _trace.reset()
    calls0 = [c.stats() for c in worker_ctx]
    t0 = time.perf_counter()
    merged = stats = None
    for _ in range(steps):
        merged, stats = step()
    group.barrier()
    wall_here = time.perf_counter() - t0
    calls1 = [c.stats() for c in worker_ctx]
    n_done = max(1, len(stats) * steps)
    calls_per_window = {k_: sum(b[k_] - a[k_] for a, b in zip(calls0, calls1)) / n_done for k_ in calls1[0]}
    dt = group.max(wall_here)
    rep = _trace.report()
    in_lib = sum(sec for name, (_c, sec) in rep.items() if name.startswith("lib:"))
    stages = {name: {"calls": c, "seconds": sec} for name, (c, sec) in sorted(rep.items()) if not name.startswith("lib:")}
    lib_top = sorted(((name[4:], sec) for name, (_c, sec) in rep.items() if name.startswith("lib:")), key=lambda e: -e[1])[:8]
    qhull_wait = sum(sec for name, (_c, sec) in rep.items() if name.startswith("triangulate"))
    # DIAGNOSTIC, outside the timed region and never part of `value`: the same pass with every window's triangulation remembered from a
    # first pass -- what the library calls + the Python glue cost once Qhull is out of the picture, i.e. the rate a host with enough
    # CPU per rank could approach (on this box the timed pass is bound by its 16 CPUs' worth of Qhull)
    no_qhull = None
    if on_device:
        tri_cache[0] = W.TriangulationCache()
        all_ranks(one_pass, my_plan)                       # fills the cache
        group.barrier()
        tq = time.perf_counter()
        for _ in range(2):
            all_ranks(one_pass, my_plan)
        no_qhull = len(my_plan) * 2 / max(time.perf_counter() - tq, 1e-9)
        tri_cache[0] = None
    # what the threads could have used: every worker for the window passes, one thread for the exchange + merge behind them
    thread_seconds = max(pass_seconds[0] * n_workers + (wall_here - pass_seconds[0]), 1e-9)
    mine_rec = {"rank": group.rank, "windows": len(my_plan), "seconds": wall_here, "windows_per_s": len(my_plan) * steps / wall_here,
                "in_library_s": in_lib, "host_glue_share": 1.0 - in_lib / thread_seconds, "threads": n_workers,
                "qhull_wait_s": qhull_wait, "python_share": max(0.0, 1.0 - (in_lib + qhull_wait) / thread_seconds),
                "qhull_wait_share": qhull_wait / thread_seconds, "serial_tail_s_per_step": (wall_here - pass_seconds[0]) / steps,
                "windows_per_s_triangulations_given": no_qhull,
                "runtime_calls_per_window": calls_per_window, "table_allgather_ms": (sum(exchange_ms) / len(exchange_ms)) if exchange_ms else None,
                "qhull_helpers": _qp.pool().n, "qhull_domains": len(_qp.pool().domains), "local_world": _qp.local_world()[0],
                "cells": int(sum(w["n_mov"] for w in my_plan)), "pairs": int(sum(s["pairs"] for s in stats)),
                "triangles": int(sum(s["triangles"] for s in stats))}
    every = group.allgather_object(mine_rec)
    rccl = comm_report(Env(args, group, ctx, ctx, comm, transport), np) if comm is not None else None
    # N=1: the caller's oracle leg (bench.py: four windows through the CPU oracle, as the CPU baseline and as the parity check of what the
    # GPU produced for them -- the package itself never imports the oracle)
    cpu, parity = None, "not checked in this run (the oracle only runs in the cpu_baseline leg: N=1 without --no-cpu-baseline)"
    if group.rank == 0 and group.world == 1 and cpu_baseline is not None:
        cpu, parity = cpu_baseline(dict(plan=plan, my_plan=my_plan, r_df=r_df, m_df=m_df, cols=cols, op=op, ref_sec=ref_sec, mov_sec=mov_sec,
                                        dref=dref, dmov=dmov, path_kw=path_kw, ctx=ctx, on_device=on_device))
    out = None
    if group.rank == 0:
        total_pairs = float(sum(w["n_mov"] * w["n_ref"] for w in plan))
        es, k = 4, 8
        P, Tr = sum(r["pairs"] for r in every), sum(r["triangles"] for r in every)
        touched = P * (2 * es * (T + 2) + 8 + es) + Tr * (74 + 12 + 3 * 40 + 1) + 16 * k * sum(r["cells"] for r in every)   # SURVEY 8d per-unit figures
        lib_s = max(r["in_library_s"] for r in every) / steps
        out = {"metric": baseline_metric(), "value": total_pairs * steps / dt, "unit": "cell-pairs/s", "n_gpus": group.world,
               "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"cfg5: {n}-cell section, {len(plan)} sliding windows (window 1200, overlap 300, ~{int(np.mean([w['n_mov'] for w in plan]))} "
                                      f"aligned cells each), T={T}, r=25 / k={k} prune, fp32 pair costs, Delaunay filter / weights / signs, greedy incumbent, "
                                      "orientation + XY-order + area-flip sweeps per window, window tables exchanged once and merged",
                          "pipeline": ("device: both sections resident in HBM and binned on the window grid, two library calls per window (csrc/window.hip); the "
                                       "host triangulates (Qhull helpers) and receives the match" if on_device
                                       else "columns: subsetting, compaction and gathers on the host, every kernel through host buffers"),
                          "parallelism": f"whole windows round-robin (heaviest first) x{group.world}; no collective inside a window; one all-gather of the "
                                         "ranks' match tables per pass" + (f": {transport}" if comm is not None else "")},
               "windows_per_s": len(plan) * steps / dt,
               "aligned_cells_per_s": float(sum(w["n_mov"] for w in plan)) * steps / dt,
               "aligned_cells_per_window": float(np.mean([w["n_mov"] for w in plan])),
               "per_rank": {"windows": [r["windows"] for r in every], "windows_per_s": [r["windows_per_s"] for r in every],
                            "host_glue_share": [r["host_glue_share"] for r in every], "python_share": [r["python_share"] for r in every],
                            "qhull_wait_s_per_step": [r["qhull_wait_s"] / steps for r in every],
                            "in_library_s_per_step": [r["in_library_s"] / steps for r in every],
                            "table_allgather_ms": [r["table_allgather_ms"] for r in every], "qhull_helpers": [r["qhull_helpers"] for r in every],
                            "windows_per_s_triangulations_given": [r["windows_per_s_triangulations_given"] for r in every],
                            "qhull_l3_domains": [r["qhull_domains"] for r in every]},
               "host_glue_share": mine_rec["host_glue_share"],
               "host_glue_share_means": "1 - (wall time inside libsame_hip calls, summed over the worker threads) / (wall time of the window passes x threads + "
                                        "wall time of the exchange and merge behind them, which one thread runs), "
                                        "rank 0: Python / numpy / scipy glue, waiting for the Qhull helpers and the table exchange included",
               "python_share": mine_rec["python_share"],
               "qhull_wait_share": mine_rec["qhull_wait_share"],
               "qhull_wait_share_means": "the worker threads' waits for the Qhull helpers (hand-over when all are busy + collecting answers) over the same "
                                         "capacity: host_glue_share = qhull_wait_share + python_share",
               "serial_tail_s_per_step": mine_rec["serial_tail_s_per_step"],
               "python_share_means": "host_glue_share without the worker threads' waits for the Qhull helpers: what Python / numpy itself takes of the "
                                     "threads' time (the merge and the table exchange included)",
               "threads_per_rank": n_workers,
               "windows_per_s_triangulations_given": None if no_qhull is None else sum(r["windows_per_s_triangulations_given"] or 0.0 for r in every),
               "windows_per_s_triangulations_given_means": "DIAGNOSTIC, not a throughput: the rate of two extra passes (windows only: no exchange, no merge) "
                                                           "in which every window's Delaunay simplices are remembered from an earlier pass, summed over "
                                                           "the ranks -- what the library calls and the Python glue allow once Qhull is out of the picture",
               "runtime_calls_per_window": mine_rec["runtime_calls_per_window"],
               "runtime_calls_per_window_means": "kernel launches / hipMemsetAsync fills / hipMemcpyAsync copies / stream waits the library issued per window on rank 0, "
                                                 "counted by the library itself (same_ctx_stat) over the timed passes; the merge's de-duplication included",
               "table_allgather": None if comm is None else {
                   "ms": mine_rec["table_allgather_ms"], "bytes_per_rank": int(exchange_bytes[0]), "bytes_total": int(exchange_bytes[0]) * group.world,
                   "timed_with": ("HIP events on the stream the all-gather ran on (same_comm_gather_time), rank 0" if not comm.synchronous
                                  else "host wall time of the host-transport exchange (no RCCL communicator), rank 0")},
               "qhull": {"helpers": _qp.pool().n, "helpers_all_ranks": sum(r["qhull_helpers"] for r in every), "ranks_on_this_host": every[0]["local_world"],
                         "l3_domains_used": len(_qp.pool().domains), "cpu_budget": _qp.cpu_budget(),
                         "waiting_s_per_step_rank0": sum(v["seconds"] for k_, v in stages.items() if k_.startswith("triangulate")) / steps,
                         "what": "helper processes that run scipy.spatial.Delaunay for the windows ahead (a6 stays on the host); waiting = the worker "
                                 "threads' time in the hand-over (all helpers busy) and in collecting an answer, summed over the threads"},
               "stages_rank0": stages, "library_calls_rank0_top": [{"entry_point": nme, "seconds": sec} for nme, sec in lib_top],
               "merged_matches": int(len(merged)),
               "roofline": {"bound": "hbm", "kernel": "window pipeline: many small gather / latency-bound kernels (the padded / pair cost kernel is the largest)",
                            "achieved": touched / lib_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": touched / lib_s / 1e9 / HBM_PEAK_GBS,
                            "traffic": None, "algorithmic_bytes_per_step": touched,
                            "note": "touched bytes per step (SURVEY 8d per-unit figures: pairs x (2 s (T+2) + 8 + s), triangles x (74 + 133), 16 k per aligned "
                                    "cell) over the slowest rank's time inside libsame_hip per step; this configuration is bound by launch latency and host "
                                    "glue, not by HBM -- see host_glue_share"},
               "cpu_baseline": cpu, "parity_spot_check": parity}
        if rccl is not None:
            out["rccl"] = rccl
    group.barrier()
    for sec in (dref, dmov):
        if sec is not None:
            sec.close()
    for c in worker_ctx[1:]:
        c.close()
    return out if group.rank == 0 else None
