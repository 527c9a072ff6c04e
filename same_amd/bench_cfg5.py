"""BASELINE cfg 5 as a bench step (bench.py --workload cfg5, and the `cfg5` record every bench line embeds): the reference's window
loop src/same.py:507-593 with whole windows dealt to the ranks.  Runs on the caller's host group, context and communicator -- no
process of its own -- so the same code measures one rank or eight."""
import os
import time

from .bench_common import HBM_PEAK_GBS, Env, baseline_metric, comm_report, note

RECORD_KEYS = ("value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "dtype", "windows_per_s", "aligned_cells_per_s",
               "aligned_cells_per_window", "windows_per_s_triangulations_given", "windows_per_s_triangulations_given_merged",
               "windows_per_s_triangulations_given_best_pass", "windows_per_s_triangulations_given_merged_best_pass", "native_delaunay",
               "per_rank",
               "host_glue_share", "python_share", "qhull_wait_share", "own_triangulator_thread_s_per_step", "serial_tail_s_per_step",
               "serial_tail_without_wait_s_per_step",
               "table_gather_s_per_step",
               "after_windows_s_per_step", "unsharded_s_per_step", "seam_wait_s_per_step", "merge_stages_s_per_step_rank0",
               "amdahl_bound_at_8_ranks", "amdahl",
               "seam_exchange", "deal", "delaunay", "threads_per_rank", "runtime_calls_per_window", "runtime_calls_per_pass_merge", "qhull",
               "merged_matches", "parity_spot_check", "rccl",
               "product_function", "api_path_windows_per_s", "api_path", "window_calls_only_windows_per_s")


def record(line):
    """The sub-record a bench line of another workload carries as `cfg5`: the cfg 5 line's own numbers, without its prose."""
    rec = {k_: line.get(k_) for k_ in RECORD_KEYS if k_ in line}
    rec["workload"], rec["pipeline"] = line["config"]["workload"], line["config"]["pipeline"]
    rec["what"] = ("BASELINE cfg 5 (whole sliding windows dealt to the ranks in runs of the plan, fp32 costs, all sweeps, the window merge "
                   "per rank "
                   "with one exchange of the seam rows) measured in "
                   "this job, on its ranks, contexts and communicator, after this line's own timed region; windows_per_s is the whole "
                   "job's")
    return rec


def run(args, group, ctx, comm, transport, steps, warmup, cpu_baseline=None):
    """BASELINE cfg 5: a section of --cfg5-cells cells tiled into overlapping windows (src/same.py:481-488), the windows dealt
    round-robin (heaviest first) to the ranks, every window through the whole pre-MIP path with fp32 costs and all three sweeps --
    by the PRODUCT function `same_amd.sliding_window_incumbent` (the reference's sliding_window_matching signature with the greedy MIP
    start, src/init_helpers.py:109-133, as each window's solution: there is no solver on the GPU box, and that start is what the
    reference hands Gurobi as its first incumbent).  --cfg5-pipeline device: both frames resident on the device (`resident_frames`,
    uploaded and binned before the timed region, as ④ of the bench contract has inputs resident), two library calls per window
    (csrc/window.hip), the host triangulates; --cfg5-pipeline frames: the function's general route on host frames, every kernel through
    host buffers (the pipeline of rounds 1-3).  Every rank's central-trimmed match table is exchanged in ONE device all-gather
    (dist.allgather_table) and merged (src/helpers.py:692-815, de-duplication on the GPU).
    `ctx` is the context the communicator lives on (the exchange runs on its stream); worker threads get contexts of their own.
    One step = the whole plan once (every rank its share) + the exchange + the merge (on rank 0, which owns the result).
    cpu_baseline: None, or the caller's function (state dict) -> (cpu_baseline record, parity text) run on rank 0 at world 1.
    -> the line as a dict on rank 0, None elsewhere; nothing is closed here.  value = dense-equivalent cell pairs
    (sum over windows of aligned x ref cells in the window) per second; `windows_per_s` per rank and the share of the step
    spent outside libsame_hip calls (`host_glue_share`) come from the stage markers of same_amd/_trace.py.  After the timed region
    rank 0 also times the reference's OWN signature (`api_path`): sliding_window_matching with a stand-in for the solver half."""
    import numpy as np
    import pandas as pd

    import same_amd
    from same_amd import _lib, _trace, synth
    from same_amd.incumbent import incumbent_of_prepared
    from same_amd.dist import MergeChannel
    from same_amd import windows as W
    from same_amd.windows import deal_windows, window_plan

    _trace.enable(True)
    _lib.instrument()
    n, T = int(args.cfg5_cells), 8
    ref = synth.make_cells(n, T, seed=0)
    mov = synth.make_jittered(ref, seed=1)
    r_df, m_df = synth.to_frame(ref), synth.to_frame(mov)
    r_df["Cell_Num_Old"], m_df["Cell_Num_Old"] = np.arange(len(r_df)), np.arange(len(m_df))
    cols = synth.type_columns(T)
    op = dict(radius=25, knn=8, no_match_penalty=100, hip_cost_dtype="float32", window_size=1200, overlap=300, min_cells_per_window=10)
    timed_native = getattr(args, "cfg5_delaunay", "qhull") == "native"
    if timed_native:
        op["hip_delaunay"] = "native"            # the timed step itself on the opt-in route (same_amd/delaunay.py)
    plan = window_plan(ref["xy"], mov["xy"], 1200, 300, 10)
    deal = getattr(args, "cfg5_deal", None) or "block"
    owner = deal_windows(plan, group.world, deal)
    my_plan = [plan[q] for q in np.flatnonzero(owner == group.rank)]
    note(group, f"cfg5: {n} cells, {len(plan)} windows of ~{int(np.mean([w['n_mov'] for w in plan]))} aligned cells; this rank runs "
                f"{len(my_plan)} ({deal} deal)")

    on_device = args.cfg5_pipeline == "device"
    # device pipeline: the two frames go to the device ONCE, before the timed region (inputs resident, as the bench contract asks), and
    # every step's call names them; frames pipeline: the frames themselves
    resident = same_amd.resident_frames(r_df, m_df, ctx=ctx) if on_device else None
    frame_args = (resident, resident) if on_device else (r_df, m_df)
    shard = (group.rank, group.world, deal) if group.world > 1 else None
    # carries the seam rows of the window merge (one small all-gather)
    channel = MergeChannel(group, ctx, comm) if group.world > 1 else None

    from same_amd import qhull_pool as _share

    cpu_share = _share.cpu_budget() / _share.cpu_sharers()[0]          # this rank's part of the CPUs it may use
    default_threads = (2 if cpu_share >= 8 else 1) if on_device else 1  # a second Python thread only pays where there are CPUs to feed it
    n_workers = max(1, int(args.cfg5_threads if args.cfg5_threads is not None else default_threads))
    tri_cache = [None]        # set for the diagnostic pass after the timed loop (triangulations remembered: Qhull out of the picture)
    # the product function makes (and closes) the other workers' contexts itself; their calls are summed by _lib.instrument
    worker_ctx = [ctx]

    def one_pass(merge=True):
        """this rank's share of the plan through the product function, window merge included
        -> (its part of the merged table, per-window stats)"""
        kw = dict(workers=n_workers, triangulator=tri_cache[0]) if on_device else dict(_route="general", _pipeline="frames")
        return same_amd.sliding_window_incumbent(*frame_args, commonCT=cols, optim_params=dict(op), return_stats=True, ctx=ctx,
                                                 _shard=shard,
                                                 merge=merge, _merge_channel=channel if merge else None, **kw)

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

    def step():
        # ONE call of the product function per rank: its windows, the merge of what only this rank can see, the exchange of the seam rows,
        # the common seam step, the columns of the rows that stay.  The merged table is the job's result and stays dealt over the ranks
        # (every part in the order of the aligned ids; dist.sharded_merged_window_incumbent(gather=...) joins them where a caller wants one
        # frame)
        return all_ranks(one_pass)

    from same_amd import qhull_pool as _qp

    # warm-up: scratch slots, Qhull helpers, first-launch costs -- and, on the device path, every window state a worker keeps
    # in flight gets its buffers (they stay with the context afterwards: the timed passes allocate nothing)
    if warmup >= 1:
        # one whole pass: the sections go up and are binned, the plan and the type sets are remembered, every window state a worker keeps
        # in flight gets its buffers (they stay with the contexts), the accumulators their arrays -- and a second one while the first one's
        # table is still alive, as a caller's loop has it: the table's columns live in page-locked blocks (windows.PINNED_BLOCKS), two of
        # which alternate in steady state, and pinning 140 MB costs ~35 ms once
        first = all_ranks(one_pass)
        second = all_ranks(one_pass)
        del first, second
    group.barrier()
    _trace.reset()
    calls0 = [c.stats() for c in worker_ctx]
    merge_calls = lambda: dict(next(iter(resident._frames.values())).__dict__.get("merge_runtime_calls", {})) if on_device else {}
    merge_calls0 = merge_calls()
    seam0 = (channel.sent_rows, channel.gather_ms) if channel is not None else (0, 0.0)
    t0 = time.perf_counter()
    merged = stats = None
    for _ in range(steps):
        merged, stats = step()
    group.barrier()
    wall_here = time.perf_counter() - t0
    calls1 = [c.stats() for c in worker_ctx]
    seam1 = (channel.sent_rows, channel.gather_ms) if channel is not None else (0, 0.0)
    # the calls counted are those of `ctx`: worker 0's windows (the first of n_workers contiguous runs of this rank's share) + the merge's
    # de-duplication
    n_done = max(1, (len(stats) // n_workers) * steps)
    # ... without what the merge itself asked for, which is per PASS, not per window (resolve + finish: a sort's worth of launches)
    merge_calls1 = merge_calls()
    per_pass = {k_: (merge_calls1.get(k_, 0) - merge_calls0.get(k_, 0)) / steps for k_ in calls1[0]}
    calls_per_window = {k_: (sum(b[k_] - a[k_] for a, b in zip(calls0, calls1)) - per_pass[k_] * steps) / n_done for k_ in calls1[0]}
    dt = group.max(wall_here)
    rep = _trace.report()
    # same_delaunay2d is host work on the triangulator's own threads (opt-in route): not part of the worker threads' time in the library
    own_tri_s = sum(sec for name, (_c, sec) in rep.items() if name == "lib:same_delaunay2d")
    in_lib = sum(sec for name, (_c, sec) in rep.items() if name.startswith("lib:")) - own_tri_s
    stages = {name: {"calls": c, "seconds": sec} for name, (c, sec) in sorted(rep.items()) if not name.startswith("lib:")}
    lib_top = sorted(((name[4:], sec) for name, (_c, sec) in rep.items() if name.startswith("lib:") and name != "lib:same_delaunay2d"),
                     key=lambda e: -e[1])[:8]
    qhull_wait = sum(sec for name, (_c, sec) in rep.items() if name.startswith("triangulate"))
    # DIAGNOSTIC, outside the timed region and never part of `value`: the same pass with every window's triangulation remembered from a
    # first pass -- what the library calls + the Python glue cost once Qhull is out of the picture, i.e. the rate a host with enough
    # CPU per rank could approach (on this box the timed pass is bound by its 16 CPUs' worth of Qhull)
    no_qhull = no_qhull_merged = no_qhull_best = no_qhull_merged_best = calls_only = None
    if on_device:
        tri_cache[0] = W.TriangulationCache()
        all_ranks(one_pass, False)                         # fills the cache

        def passes_per_s(merge_too, n_passes=3):
            """-> (windows/s over n_passes passes, windows/s of the fastest of them): the host's CPUs are shared, one slow pass is common"""
            group.barrier()
            times = []
            for _ in range(n_passes):
                tq = time.perf_counter()
                all_ranks(one_pass, merge_too)
                times.append(time.perf_counter() - tq)
            return len(my_plan) * n_passes / max(sum(times), 1e-9), len(my_plan) / max(min(times), 1e-9)

        no_qhull, no_qhull_best = passes_per_s(False)          # the product function's table of every window, as in round 5 (no merge)
        no_qhull_merged, no_qhull_merged_best = passes_per_s(True)     # ... and with the window merge, as the timed step runs it
        # ... and the window calls alone (stage + filter_finish in batches, the per-window Python of iter_device_windows; no table): one
        # thread
        frames_obj = next(iter(resident._frames.values()))

        def calls_only_pass(n_threads):
            import threading

            ctxs = frames_obj.worker_contexts(n_threads)
            cut = [len(my_plan) * q // n_threads for q in range(n_threads + 1)]

            def walk(q):
                for _dw in frames_obj.windows(my_plan[cut[q]:cut[q + 1]], triangulator=tri_cache[0], ctx=ctxs[q]):
                    pass

            tq_ = time.perf_counter()
            for _ in range(2):
                threads = [threading.Thread(target=walk, args=(q,)) for q in range(1, n_threads)]
                [t.start() for t in threads]
                walk(0)
                [t.join() for t in threads]
            return len(my_plan) * 2 / max(time.perf_counter() - tq_, 1e-9)

        calls_only = {"one_thread": calls_only_pass(1), "worker_threads": n_workers, "with_the_worker_threads": calls_only_pass(n_workers)}
        tri_cache[0] = None
    # OPT-IN, outside the timed region and never part of `value`: the timed step itself with optim_params["hip_delaunay"] = "native" --
    # the windows triangulated by libsame_hip's own triangulator on threads of this process wherever its answer is provably Qhull's set
    # of triangles, windows that count an order tie finished again with scipy's (same_amd/delaunay.py).  Its table must BE the timed step's
    native = None
    if on_device and not timed_native and not getattr(args, "no_extras", False):
        from same_amd import delaunay as _del

        tr = _del.shared()

        def native_pass():
            return same_amd.sliding_window_incumbent(*frame_args, commonCT=cols, optim_params=dict(op, hip_delaunay="native"),
                                                     return_stats=True, ctx=ctx, _shard=shard, merge=True, _merge_channel=channel,
                                                     workers=n_workers)

        first = all_ranks(native_pass)
        identical = bool(first[0].equals(merged)) and first[1] == stats
        del first
        before, passes = (tr.submitted, tr.asked_qhull), 3
        group.barrier()
        tq = time.perf_counter()
        for _ in range(passes):
            all_ranks(native_pass)
        group.barrier()
        native = {"seconds": time.perf_counter() - tq, "passes": passes, "windows": len(my_plan), "identical": identical,
                  "triangulated": tr.submitted - before[0], "sent_back_to_qhull": tr.asked_qhull - before[1], "threads": tr.threads}
    # after a rank's last window: the merge (keys, de-duplication, matching; the seam rows' exchange and the common seam step) and the
    # columns of the rows that stay -- one thread, while the workers' threads are done
    seconds_of = lambda prefix: sum(sec for name, (_c, sec) in rep.items() if name.startswith(prefix))
    merge_s, table_s = seconds_of("merge:"), seconds_of("table (columns")
    exchange_s, seam_step_s = seconds_of("merge: seam rows exchanged"), seconds_of("merge: seam step")
    unsharded_s = exchange_s + seam_step_s
    walk_s = max(wall_here - merge_s - table_s, 1e-9)
    # what the threads could have used: every worker for the window passes, one thread for what comes after them
    thread_seconds = max(walk_s * n_workers + merge_s + table_s, 1e-9)
    mine_rec = {"rank": group.rank, "windows": len(my_plan), "seconds": wall_here, "windows_per_s": len(my_plan) * steps / wall_here,
                "in_library_s": in_lib, "host_glue_share": 1.0 - in_lib / thread_seconds, "threads": n_workers,
                "qhull_wait_s": qhull_wait, "own_triangulator_thread_s_per_step": own_tri_s / steps,
                "python_share": max(0.0, 1.0 - (in_lib + qhull_wait) / thread_seconds),
                "qhull_wait_share": qhull_wait / thread_seconds, "serial_tail_s_per_step": merge_s / steps,
                "table_gather_s_per_step": table_s / steps, "unsharded_s_per_step": unsharded_s / steps,
                "seam_exchange_s_per_step": exchange_s / steps, "seam_step_s_per_step": seam_step_s / steps,
                "merge_stages_s_per_step": {name: sec / steps for name, (_c, sec) in sorted(rep.items()) if name.startswith("merge:")},
                "windows_per_s_triangulations_given": no_qhull, "windows_per_s_triangulations_given_merged": no_qhull_merged,
                "windows_per_s_triangulations_given_best_pass": no_qhull_best,
                "windows_per_s_triangulations_given_merged_best_pass": no_qhull_merged_best,
                "window_calls_only_windows_per_s": calls_only, "native_delaunay": native, "merged_rows": int(len(merged)),
                "seam_rows_sent_per_step": None if channel is None else (seam1[0] - seam0[0]) / steps,
                "seam_gather_ms": None if channel is None else (seam1[1] - seam0[1]) / steps,
                "runtime_calls_per_window": calls_per_window, "runtime_calls_per_pass_merge": per_pass,
                "qhull_helpers": _qp.pool().n, "qhull_domains": len(_qp.pool().domains), "local_world": _qp.local_world()[0],
                "cells": int(sum(w["n_mov"] for w in my_plan)), "pairs": int(sum(s["pairs"] for s in stats)),
                "triangles": int(sum(s["triangles"] for s in stats))}
    every = group.allgather_object(mine_rec)
    rccl = comm_report(Env(args, group, ctx, ctx, comm, transport), np) if comm is not None else None
    # N=1: the caller's oracle leg (bench.py: four windows through the CPU oracle, as the CPU baseline and as the parity check of what the
    # GPU produced for them -- the package itself never imports the oracle)
    cpu, parity = None, "not checked in this run (the oracle only runs in the cpu_baseline leg: N=1 without --no-cpu-baseline)"
    if group.rank == 0 and group.world == 1 and cpu_baseline is not None:
        cpu, parity = cpu_baseline(dict(plan=plan, my_plan=my_plan, r_df=r_df, m_df=m_df, cols=cols, op=op, ctx=ctx, on_device=on_device))
    # THE REFERENCE'S OWN SIGNATURE, timed (rank 0, after the timed region, never part of `value`): sliding_window_matching on this job's
    # frames (a) with `incumbent_of_prepared` standing in for the solver half of run_same, (b) on a few windows with a do-nothing
    # `gurobipy` in place (bench_solver_double): what a licensed run pays per window in Python before the solver does anything.
    api_path = amdahl = None
    if group.rank == 0 and not getattr(args, "no_extras", False):
        api_path = _api_path_record(same_amd, incumbent_of_prepared, frame_args, r_df, m_df, cols, op, plan, on_device, _trace)
        amdahl = _amdahl_at_8_ranks(same_amd, frame_args, cols, op, deal, every, steps, ctx, on_device, n_workers, _trace,
                                    next(iter(resident._frames.values())) if on_device else None)
    out = None
    if group.rank == 0:
        out = _line(args, group, comm, transport, plan, deal, every, mine_rec, stages, lib_top, steps, warmup, dt, n, T, on_device,
                    n_workers,
                    cpu, parity, api_path, amdahl, _qp)
        if rccl is not None:
            out["rccl"] = rccl
    group.barrier()
    if resident is not None:
        resident.close()
    return out if group.rank == 0 else None


class _OneRankOfEight:
    """dist.MergeChannel's interface for ONE rank of a deal that is not running: what the rank would send is kept, what comes back is its
    own rows alone (the common step on all eight ranks' rows is timed apart)."""

    def __init__(self, rank, world):
        self.rank, self.world, self.sent, self.sent_rows, self.gather_ms = rank, world, None, 0, 0.0

    def tables(self, table):
        self.sent = table
        return [table]

    def max(self, v):
        return float(v)


def _amdahl_at_8_ranks(same_amd, frame_args, cols, op, deal, every, steps, ctx, on_device, n_workers, _trace, frames_obj):
    """DIAGNOSTIC (rank 0, after the timed region): what bounds this configuration at 8 ranks, from this job's own stage times.

    A step is this much work: the windows, the columns of the final table, the merge of what a rank can see alone -- all of it dealt
    with the windows -- and what is NOT dealt: the exchange of the seam rows and the common seam step.  The dealt part is measured
    by this run (every rank's stage times).  The rest is measured here for the 8-rank deal of THIS job: each of the eight shares is run
    through the product function once, alone (its windows, its accumulator, its seams -- the merge stages of the slowest share are what a
    rank of eight would spend on its own rows), and the common step is then run once on the seam rows all eight would have sent."""
    import numpy as np

    from . import incumbent
    from . import merge as M

    ranks8 = 8
    kw = dict(workers=n_workers) if on_device else dict(_route="general", _pipeline="frames")
    local_s, sent, windows_s = [], [], []
    for q in range(ranks8):
        channel = _OneRankOfEight(q, ranks8)
        _trace.reset()
        t0 = time.perf_counter()
        same_amd.sliding_window_incumbent(*frame_args, commonCT=cols, optim_params=dict(op), ctx=ctx, merge=True, _shard=(q, ranks8, deal),
                                          _merge_channel=channel, **kw)
        windows_s.append(time.perf_counter() - t0)
        rep = _trace.report()
        local_s.append(sum(sec for name, (_c, sec) in rep.items() if name.startswith("merge:") and "seam rows exchanged" not in name))
        sent.append(channel.sent if channel.sent is not None else {"row": np.zeros(0, np.int64)})
    t0 = time.perf_counter()
    if on_device:
        incumbent._seam_step_on_device(frames_obj, ctx, sent, 0)
    else:
        M.part_after_seam_step(None, np.zeros(0, np.int64), sent, 0, M._device_dedup(ctx))
    common_s = time.perf_counter() - t0
    seam_rows = [len(t["row"]) for t in sent]
    # the dealt work of one step, summed over this run's ranks (at one rank: the step itself), without what the run spent on seams
    dealt_s = sum(r_["seconds"] / steps - r_["unsharded_s_per_step"] for r_ in every)
    merge_alone_s = sum(r_["serial_tail_s_per_step"] - r_["unsharded_s_per_step"] for r_ in every)
    # an all-gather ends when the LAST rank arrives: the rank that arrives last sees the exchange itself, the others also their wait for it
    # (that wait is the deal's imbalance, part of the dealt work's max over ranks -- not something every rank repeats)
    measured_gather = [r_["seam_exchange_s_per_step"] for r_ in every if len(every) > 1]
    exchange_s = min(measured_gather) if measured_gather else 1.0e-3
    per_rank_8 = (dealt_s - merge_alone_s) / ranks8 + max(local_s)
    not_dealt_8 = exchange_s + common_s
    return {"value": dealt_s / (per_rank_8 + not_dealt_8),
            "dealt_s_per_step": dealt_s, "of_which_merge_s": merge_alone_s,
            "at_8_ranks": {"merge_of_own_rows_s_max_over_ranks": max(local_s), "merge_of_own_rows_s_by_rank": local_s,
                           "seam_rows_by_rank": seam_rows, "seam_rows_share": sum(seam_rows) / max(1,
                                                                      sum(r_["merged_rows"] for r_ in every)),
                           "common_seam_step_s": common_s, "seam_exchange_s": exchange_s,
                           "seam_exchange_s_is": ("measured by this run: the all-gather as the last rank to arrive saw it"
                                                  if measured_gather
                                                  else "assumed (one rank: nothing to exchange; 2 ranks over the host transport measure "
                                                       "1.4-1.8 ms)"),
                           "one_share_alone_s_by_rank": windows_s},
            "means": "speed-up bound at 8 ranks = dealt / ((dealt - merge) / 8 + slowest share's merge of its own rows + seam exchange + "
                     "common "
                     "seam step): the windows, the table and the merge of a rank's own rows are dealt with the windows (Qhull's share "
                     "scales "
                     "with the CPUs each rank has); the seam exchange and the common seam step are what every rank repeats whole.  Each "
                     "of the eight shares was run through the product function alone for its merge stages, the common step once on all "
                     "eight shares' seam rows"}


def _native_record(every):
    recs = [r.get("native_delaunay") for r in every]
    if any(r is None for r in recs):
        return None
    return {"windows_per_s": sum(r["windows"] * r["passes"] for r in recs) / max(r["seconds"] for r in recs),
            "table_identical_to_the_timed_step": all(r["identical"] for r in recs),
            "windows_triangulated": sum(r["triangulated"] for r in recs), "sent_back_to_qhull": sum(r["sent_back_to_qhull"] for r in recs),
            "threads_per_rank": recs[0]["threads"], "passes": recs[0]["passes"],
            "what": "OPT-IN (optim_params['hip_delaunay'] = 'native'), measured after the timed region and never part of `value`: the "
                    "timed step's own call with the windows triangulated by libsame_hip's triangulator (same_delaunay2d, host threads of "
                    "this "
                    "process, no helper processes) wherever its answer is beyond doubt Qhull's set of triangles; a window in which the "
                    "device counts an order tie -- a place where the reference's numbers hang on Qhull's ORDER of triangles or corners -- "
                    "is finished again with scipy's simplices (`sent_back_to_qhull`).  `table_identical_to_the_timed_step`: this rank's "
                    "part of the merged table and every window's counters equal the timed (scipy) step's, checked in this run"}


def _line(args, group, comm, transport, plan, deal, every, mine_rec, stages, lib_top, steps, warmup, dt, n, T, on_device, n_workers,
          cpu, parity, api_path, amdahl, _qp):
    import numpy as np

    total_pairs = float(sum(w["n_mov"] * w["n_ref"] for w in plan))
    es, k = 4, 8
    P, Tr = sum(r["pairs"] for r in every), sum(r["triangles"] for r in every)
    # SURVEY 8d per-unit figures
    touched = P * (2 * es * (T + 2) + 8 + es) + Tr * (74 + 12 + 3 * 40 + 1) + 16 * k * sum(r["cells"] for r in every)
    lib_s = max(r["in_library_s"] for r in every) / steps
    mean_cells = int(np.mean([w["n_mov"] for w in plan]))
    by_rank = lambda key: [r.get(key) for r in every]
    given = [r["windows_per_s_triangulations_given"] for r in every]
    given_merged = [r["windows_per_s_triangulations_given_merged"] for r in every]
    return {
        "metric": baseline_metric(), "value": total_pairs * steps / dt, "unit": "cell-pairs/s", "n_gpus": group.world,
        "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"cfg5: {n}-cell section, {len(plan)} sliding windows (window 1200, overlap 300, ~{mean_cells} aligned cells "
                        f"each), "
                        f"T={T}, r=25 / k={k} prune, fp32 pair costs, Delaunay filter / weights / signs, greedy incumbent, orientation + "
                        "XY-order + area-flip sweeps per window, window merge (one row per aligned and per reference cell)",
            "pipeline": ("device: same_amd.sliding_window_incumbent(merge=True) on resident frames -- both sections in HBM, binned on the "
                         "window grid, two library calls per window (csrc/window*.hip); the host triangulates ("
                         + ("libsame_hip's own triangulator on threads of the process, windows with an order tie and sets it will not "
                            "answer for go to scipy: optim_params['hip_delaunay'] = 'native'" if getattr(args, "cfg5_delaunay",
                                                                      "qhull") == "native"
                            else "scipy.spatial.Delaunay in helper processes, the reference's call")
                         + ") and receives the match; the merge reads the rows' keys, only the rows it keeps get their columns" if on_device
                         else "frames: same_amd.sliding_window_incumbent(merge=True), general route on host frames -- every window's "
                              "frames "
                              "cut on the host, every kernel through host buffers, a DataFrame per window"),
            "parallelism": f"whole windows, {deal} deal x{group.world}; no collective inside a window; every rank merges what only it can "
                           "see, one all-gather of the seam rows per pass" + (f": {transport}" if comm is not None else "")},
        "deal": deal, "delaunay": getattr(args, "cfg5_delaunay", "qhull"),
        "windows_per_s": len(plan) * steps / dt,
        "aligned_cells_per_s": float(sum(w["n_mov"] for w in plan)) * steps / dt,
        "aligned_cells_per_window": float(np.mean([w["n_mov"] for w in plan])),
        "per_rank": {"windows": by_rank("windows"), "windows_per_s": by_rank("windows_per_s"),
                     "host_glue_share": by_rank("host_glue_share"),
                     "python_share": by_rank("python_share"), "qhull_wait_s_per_step": [r["qhull_wait_s"] / steps for r in every],
                     "in_library_s_per_step": [r["in_library_s"] / steps for r in every],
                     "serial_tail_s_per_step": by_rank("serial_tail_s_per_step"),
                     "table_gather_s_per_step": by_rank("table_gather_s_per_step"),
                     "unsharded_s_per_step": by_rank("unsharded_s_per_step"),
                     "seam_exchange_s_per_step": by_rank("seam_exchange_s_per_step"),
                     "seam_step_s_per_step": by_rank("seam_step_s_per_step"), "seam_rows_sent_per_step": by_rank("seam_rows_sent_per_step"),
                     "seam_gather_ms": by_rank("seam_gather_ms"), "merged_rows": by_rank("merged_rows"),
                     "qhull_helpers": by_rank("qhull_helpers"), "windows_per_s_triangulations_given": given,
                     "qhull_l3_domains": by_rank("qhull_domains")},
        "host_glue_share": mine_rec["host_glue_share"],
        "host_glue_share_means": "1 - (wall time inside libsame_hip calls, summed over the worker threads) / (wall time of the window walk "
                                 "x "
                                 "threads + wall time of the merge and the table behind it, which one thread runs), rank 0: Python / numpy "
                                 "/ scipy glue, waiting for the Qhull helpers and the seam exchange included",
        "python_share": mine_rec["python_share"],
        "python_share_means": "host_glue_share without the worker threads' waits for the Qhull helpers: what Python / numpy itself takes "
                              "of the threads' time (the merge, the seam exchange and the table included)",
        "qhull_wait_share": mine_rec["qhull_wait_share"],
        "qhull_wait_share_means": "the worker threads' waits for whoever triangulates -- the Qhull helpers, or with --cfg5-delaunay native "
                                  "the triangulator's threads -- (hand-over when all are busy + collecting answers) over the same "
                                  "capacity: "
                                  "host_glue_share = qhull_wait_share + python_share",
        "own_triangulator_thread_s_per_step": mine_rec["own_triangulator_thread_s_per_step"],
        "own_triangulator_thread_s_means": "--cfg5-delaunay native: seconds inside same_delaunay2d per step, summed over the "
                                           "triangulator's "
                                           "threads (host work beside the worker threads, not part of the shares above); 0 on the default "
                                           "route",
        "serial_tail_s_per_step": max(by_rank("serial_tail_s_per_step")),
        "serial_tail_means": "slowest rank's time in the window merge per step (stages 'merge: ...' of same_amd/_trace.py: keys, "
                             "de-duplication on the device, matching of contested cells, the seam rows' exchange, the common seam step) -- "
                             "what round 5 measured as exchange + merge on rank 0; all of it but `unsharded_s_per_step` is per-rank work "
                             "on "
                             "the rank's own rows.  The columns of the rows that stay are `table_gather_s_per_step` (inside the window "
                             "pass "
                             "in round 5, when the pre-merge table was laid out first)",
        "serial_tail_without_wait_s_per_step": max(t_ - (x_ - min(by_rank("seam_exchange_s_per_step")))
                                                   for t_, x_ in zip(by_rank("serial_tail_s_per_step"),
                                                                     by_rank("seam_exchange_s_per_step"))),
        "serial_tail_without_wait_means": "serial_tail_s_per_step without a rank's WAIT for the last rank to reach the seam exchange (its "
                                          "time "
                                          "in the all-gather beyond the last arriver's): the merge WORK after a rank's last window.  The "
                                          "wait is the deal's imbalance as the host's CPUs made it (`seam_wait_s_per_step`) and already in "
                                          "`value`: the step ends when the last rank ends",
        "merge_stages_s_per_step_rank0": mine_rec["merge_stages_s_per_step"],
        "table_gather_s_per_step": max(by_rank("table_gather_s_per_step")),
        "after_windows_s_per_step": max(a_ + b_ for a_, b_ in zip(by_rank("serial_tail_s_per_step"), by_rank("table_gather_s_per_step"))),
        "unsharded_s_per_step": min(by_rank("seam_exchange_s_per_step")) + max(by_rank("seam_step_s_per_step")),
        "unsharded_means": "the part of a step that is not dealt with the windows: the all-gather of the seam rows (as the last rank to "
                           "arrive sees it: the others' longer stay in the call is their wait for that rank, `seam_wait_s_per_step`, the "
                           "deal's imbalance) and the common seam step every rank runs on them (0 at one rank)",
        "seam_wait_s_per_step": max(by_rank("seam_exchange_s_per_step")) - min(by_rank("seam_exchange_s_per_step")),
        "amdahl_bound_at_8_ranks": None if amdahl is None else amdahl["value"], "amdahl": amdahl,
        "seam_exchange": None if comm is None else {
            "rows_sent_per_step_by_rank": by_rank("seam_rows_sent_per_step"), "ms_by_rank": by_rank("seam_gather_ms"),
            "bytes_per_row": 41,
            "timed_with": ("HIP events on the stream the all-gather ran on (same_comm_gather_time)" if not comm.synchronous
                           else "host wall time of the host-transport exchange (no RCCL communicator)")},
        "threads_per_rank": n_workers,
        "windows_per_s_triangulations_given": None if given[0] is None else sum(v or 0.0 for v in given),
        "windows_per_s_triangulations_given_merged": None if given_merged[0] is None else sum(v or 0.0 for v in given_merged),
        "windows_per_s_triangulations_given_best_pass": None if given[0] is None else sum(
            r["windows_per_s_triangulations_given_best_pass"] or 0.0 for r in every),
        "windows_per_s_triangulations_given_merged_best_pass": None if given[0] is None else sum(
            r["windows_per_s_triangulations_given_merged_best_pass"] or 0.0 for r in every),
        "windows_per_s_triangulations_given_means": "DIAGNOSTIC, not a throughput: the rate of three extra passes of the product function "
                                                    "in "
                                                    "which every window's Delaunay simplices are remembered from an earlier pass, summed "
                                                    "over the ranks -- the pre-merge table of every window (as round 5 measured it) and, "
                                                    "`_merged`, the timed step's own call (window merge included); `_best_pass`: the "
                                                    "fastest of the three (the host's CPUs are "
                                                    "shared with other tenants: one slow pass in three is common)",
        "native_delaunay": _native_record(every),
        "window_calls_only_windows_per_s": mine_rec["window_calls_only_windows_per_s"],
        "window_calls_only_means": "DIAGNOSTIC, rank 0: windows.iter_device_windows over this rank's windows with the triangulations "
                                   "remembered and nothing done with the results -- the two batched library calls per eight windows and "
                                   "the generator's own Python, on one thread and on the product function's worker threads",
        "runtime_calls_per_window": mine_rec["runtime_calls_per_window"],
        "runtime_calls_per_pass_merge": mine_rec["runtime_calls_per_pass_merge"],
        "runtime_calls_per_window_means": "kernel launches / hipMemsetAsync fills / hipMemcpyAsync copies / stream waits the library "
                                          "issued "
                                          "per window on rank 0, counted by the library itself (same_ctx_stat) over the timed passes: the "
                                          "stage, filter + finish and collect calls; what the window merge asks for once per PASS (resolve "
                                          "+ "
                                          "finish on the accumulated rows: a sort's worth of launches) is runtime_calls_per_pass_merge",
        "qhull": {"helpers": _qp.pool().n, "helpers_all_ranks": sum(r["qhull_helpers"] for r in every),
                  "ranks_on_this_host": every[0]["local_world"], "l3_domains_used": len(_qp.pool().domains), "cpu_budget": _qp.cpu_budget(),
                  "waiting_s_per_step_rank0": sum(v["seconds"] for k_, v in stages.items() if k_.startswith("triangulate")) / steps,
                  "what": "helper processes that run scipy.spatial.Delaunay for the windows ahead (a6 stays on the host); waiting = the "
                          "worker threads' time in the hand-over (all helpers busy) and in collecting an answer, summed over the threads"},
        "stages_rank0": stages, "library_calls_rank0_top": [{"entry_point": nme, "seconds": sec} for nme, sec in lib_top],
        "merged_matches": int(sum(r["merged_rows"] for r in every)),
        "roofline": {"bound": "hbm",
                     "kernel": "window pipeline: gather / latency-bound kernels that take a group of eight windows per launch "
                                               "(greedy rounds, prune and the compactions lead)",
                     "achieved": touched / lib_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": touched / lib_s / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_step": touched,
                     "note": "touched bytes per step (SURVEY 8d per-unit figures: pairs x (2 s (T+2) + 8 + s), triangles x (74 + 133), 16 "
                             "k "
                             "per aligned cell) over the slowest rank's time inside libsame_hip per step; this configuration is bound by "
                             + ("the host -- the triangulator's threads (own_triangulator_thread_s_per_step) and what a process can submit "
                                "per second --, not by HBM or by the GPU" if getattr(args, "cfg5_delaunay", "qhull") == "native"
                                else "the host's Delaunay calls (Qhull: qhull_wait_share), not by HBM or by the GPU")},
        "product_function": "same_amd.sliding_window_incumbent(merge=True) (sliding_window_matching's arguments; the greedy MIP start as "
                            "every "
                            "window's solution; merge_window_matches_unique_ref's table) -- the timed step IS a call of it per rank"
                            + (", on frames made resident before the timed region" if on_device else ""),
        "api_path_windows_per_s": None if api_path is None else api_path["api_path_windows_per_s"], "api_path": api_path,
        "cpu_baseline": cpu, "parity_spot_check": parity}


def _api_path_record(same_amd, incumbent_of_prepared, frame_args, r_df, m_df, cols, op, plan, on_device, _trace):
    import numpy as np

    from . import bench_solver_double

    pipeline = "device" if on_device else "frames"
    t0 = time.perf_counter()
    res = same_amd.sliding_window_matching(*frame_args, commonCT=cols, optim_params=dict(op), _pipeline=pipeline,
                                           _solve=lambda prep, _outprefix: (incumbent_of_prepared(prep, cols, True)[0], {}))
    t_a = time.perf_counter() - t0
    # (b): a corner of the section holding a handful of windows, so that the default run stays within minutes
    xs = sorted({w["box"][0] for w in plan})
    ys = sorted({w["box"][2] for w in plan})
    x_hi, y_hi = plan[0]["box"][1] + (xs[1] - xs[0] if len(xs) > 1 else 0.0), plan[0]["box"][3] + (ys[1] - ys[0] if len(ys) > 1 else 0.0)
    corner = lambda df: df[(df["X"] < x_hi) & (df["Y"] < y_hi)].reset_index(drop=True)
    r_c, m_c = corner(r_df), corner(m_df)
    bench_solver_double.install()
    _trace.reset()
    cwd = os.getcwd()
    import tempfile
    try:
        with tempfile.TemporaryDirectory() as work:      # run_same writes matching_model.lp into the working directory
            os.chdir(work)
            t0 = time.perf_counter()
            res_b = same_amd.sliding_window_matching(r_c, m_c, commonCT=cols, optim_params=dict(op),
                                                     gurobi_params=dict(init_method="greedy"),
                                                     _pipeline=pipeline)
            t_b = time.perf_counter() - t0
    finally:
        os.chdir(cwd)
        bench_solver_double.uninstall()
    n_b = max(1, int(res_b["window_id"].nunique()) if len(res_b) else 1)
    rep = {name: sec for name, (_c, sec) in _trace.report().items() if not name.startswith("lib:")}
    solver_side = sum(sec for name, sec in rep.items() if name in ("MIP start", "solve (incl. lazy sweeps)", "post-solve sweeps + tables"))
    return {"what": "sliding_window_matching (the reference's signature, src/same.py:297-307) on this job's frames, after the timed "
                    "region, rank 0, one "
                    "thread: (a) all windows with the greedy incumbent standing in for the solver half of run_same; (b) the windows of one "
                    "corner of "
                    "the section with a do-nothing gurobipy double: run_same's own Python around the solver",
            "pipeline": pipeline, "windows": len(plan), "seconds": t_a, "api_path_windows_per_s": len(plan) / t_a, "matches": int(len(res)),
            "with_solver_double": {"windows": n_b, "cells": [int(len(m_c)), int(len(r_c))], "seconds_per_window": t_b / n_b,
                                   "windows_per_s": n_b / t_b,
                                   "solver_side_python_s_per_window": solver_side / n_b,
                                   "solver_side_means": "MIP start + optimize() of the double incl. one lazy callback + post-solve tables "
                                                        "(stage "
                                                        "markers of same_amd/_trace.py); the rest is model assembly (one Python object per "
                                                        "pair, "
                                                        "constraint and triangle) and the pre-MIP path",
                                   "matches": int(len(res_b))}}
