#!/usr/bin/env python3
"""bench.py -- one timed pass of the pre-MIP hot path per step, on N MI355X (one process per GPU).

`python3 bench.py --gpus N` starts its own N ranks: the parent touches no GPU, spawns N fresh children with
RANK / LOCAL_RANK / WORLD_SIZE set, relays rank 0's single JSON line and exits non-zero if any child does.  Under
`python -m torch.distributed.run` (RANK already in the environment) the process is a rank itself.  Either way the ranks
meet through same_amd/rendezvous.py (loopback TCP, plain Python); no torch anywhere, so the only librccl mapped is the
/opt/rocm one libsame_hip.so links against.

Workload `dense100k`, weak scaling (the configuration BASELINE.json's metric is quoted on; it fits one GPU):
  per rank 100 000 reference cells x 100 000 aligned cells, 20 type columns, fp64:
    1. dense L1 cost build, rows x n_ref -> 80 GB resident in HBM          (dominant kernel)
    2. radius-25 / k=32 KNN prune (caller-held grid index of the refs) + costs of the padded candidate lists
    3. [N > 1] RCCL all-gather of the pruned candidate lists (idx int32 + cost fp64), overlapped on a second stream
    4. Delaunay-triangle classes (radius/angle/type), weights and source signs
    5. orientation sweep (lazy-constraint body), XY-order sweep, signed-area flips under a nearest-reference matching
  Inputs are resident in HBM before the timed region (the Delaunay triangulation itself is an input: scipy/Qhull on the
  host, as in the reference).  Every rank owns its own block of 100 000 aligned rows against the replicated refs.
At N > 1 the same run then measures BASELINE cfg 4 as a second, embedded record (`strong_cfg4`): ONE 200k x 200k problem;
  ranks own aligned-row blocks of it (dense build in 25k-row chunks through the resident buffer), all-gather the candidate
  lists, derive the common matching, and sweep disjoint triangle blocks of the one triangulation (flag all-gather + counter
  all-reduce).  `--scaling strong --workload cfg4` runs that configuration as the main record instead.
`--workload cfg5` (BASELINE cfg 5) is a different step -- whole sliding windows dealt to the ranks in runs of the plan, fp32 costs,
  the window merge per rank with one exchange of the seam rows -- see same_amd/bench_cfg5.py; every line of the default workload -- at
  1 rank or N -- carries it as the sub-record `cfg5`, measured in the same job after the timed region (on its ranks, contexts and
  communicator; `--embed-cfg5 off` skips it).
value = aligned-ref cell pairs covered per second by the whole job.

`roofline` is for the dense kernel: algorithmic bytes s*N_r*rows + s*(T+2)*(N_r+rows) over its mean launch
time, HIP events on the stream it runs on.  Ceilings (same kernel at T=0, memset, device copy) are measured AFTER the timed
loop on the warm chip, `telemetry` is board power / shader clock sampled from sysfs while the dense kernel loops,
`valu_floor_ms_at_held_clock` / `valu_busy_frac` price the kernel's 2T+5 fp64 instructions per output at that clock,
`sweep` repeats the measurement at the type counts of the reference's real datasets.
At N > 1 the line explains itself: `rccl` is what the communicator reports (ncclCommCount, not the launcher's word),
`gather` times the all-gather on its own stream and against a second loop without it, `per_rank_dense_ms` shows slow dies.
`cpu_baseline` times the CPU oracle (scalar C port of the reference's arithmetic, 1 thread) on a bounded row sample of
the same workload, rank 0, N=1 only (section bench_oracle_legs at the end of this file; the line itself is put together by
same_amd/bench_report.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from same_amd.bench_common import Env, arm_rank_watchdog, comm_report, make_comm, note, stats3          # noqa: E402  (no GPU, no torch)
from same_amd.bench_problem import STRONG_CHUNK_BYTES, STRONG_OF, WORKLOADS, Problem, gather_report   # noqa: E402,F401
from same_amd.bench_report import N_GT1_KEYS, N_GT1_STRONG_KEYS, add_multi_rank_parts, dense_line, roofline, strong_record   # noqa: E402

_OVERLAPPED = " (overlapped on a second stream)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip ceilings / telemetry leg / T sweep (profiling runs)")
    ap.add_argument("--no-strong-record", action="store_true", help="N > 1: skip the embedded ONE-problem (cfg 4) record")
    ap.add_argument("--cpu-sample-rows", type=int, default=20000)  # 10-20 s of single-thread oracle work at dense100k
    ap.add_argument("--tail-stream", default="own", choices=("own", "shared"),
                    help="own: prune / costs / triangle maps / sweeps run on a second context (stream) beside the dense build; "
                         "shared: everything on one stream, strictly in order")
    ap.add_argument("--dense", default="exact", choices=("exact", "q32"),
                    help="exact: the bit-exact fp64 dense kernel (default, the reported kernel); q32: run the step with the opt-in "
                         "fixed-point build instead (every output within 1e-6 relative of the exact one; NOT reference arithmetic)")
    ap.add_argument("--spread", default="auto", choices=("auto", "on", "off"),
                    help="where the dense cost block comes from: 'on' = same_dev_alloc_spread (1 GiB chunks laid over the card's three "
                         "HBM regions), 'off' = plain hipMalloc, 'auto' = spread only for store-bound type counts (T <= 12); the T = 20 "
                         "headline is bound by fp64 issue and runs on a plain block")
    ap.add_argument("--cfg5-cells", type=int, default=1_000_000, help="--workload cfg5: cells per section")
    ap.add_argument("--embed-cfg5", choices=("auto", "on", "off"), default="auto",
                    help="after the timed loop, run BASELINE cfg 5 (the step of `--workload cfg5`) on this job's ranks and embed its "
                         "numbers as `cfg5` (auto: with the default workload at any rank count, and with cfg2 / cfg4 at N > 1)")
    ap.add_argument("--cfg5-pipeline", choices=("device", "frames"), default="device",
                    help="--workload cfg5: both run the product function same_amd.sliding_window_incumbent; 'device' on frames resident "
                         "on the GPU (two library calls per window), 'frames' through its general route on host frames (every window's "
                         "frames cut on the host, every kernel through host buffers: the pipeline of rounds 1-3, kept measurable)")
    ap.add_argument("--cfg5-threads", type=int, default=None,
                    help="--workload cfg5: worker threads (contexts) walking this rank's windows (default: 2 on the device pipeline where "
                         "the rank has 8 CPUs or more, else 1)")
    ap.add_argument("--cfg5-deal", choices=("block", "round_robin"), default="block",
                    help="--workload cfg5: how the windows are dealt to the ranks: 'block' = runs of the plan (strips of the window grid: "
                         "the window merge only has the strips' borders to settle between ranks), 'round_robin' = every N-th window, "
                         "heaviest first")
    ap.add_argument("--cfg5-delaunay", choices=("qhull", "native"), default="qhull",
                    help="--workload cfg5: who triangulates the windows in the TIMED step: 'qhull' = scipy.spatial.Delaunay in helper "
                         "processes, the reference's own call (default; the opt-in route is then measured after the timed region as "
                         "`native_delaunay`), 'native' = optim_params['hip_delaunay'] = 'native' (same_amd/delaunay.py)")
    ap.add_argument("--dry-launch", action="store_true", help="ranks only rendezvous (no GPU): launcher / control-plane check")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "cfg4" if args.scaling == "strong" else "dense100k"
    return args


# ======================================================================================================================
# one rank
# ======================================================================================================================
def run_rank(args):
    import numpy as np

    from same_amd import qhull_pool
    from same_amd.rendezvous import HostGroup

    # stdout carries exactly ONE line (the JSON): native libraries print there too (RCCL writes a version banner to
    # stdout when the first communicator is created), so keep the real stdout aside and point fd 1 at stderr meanwhile
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    group = HostGroup(timeout=float(os.environ.get("SAME_BENCH_RDV_TIMEOUT", "900")))
    if group.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={group.world}")
    local_rank = int(os.environ.get("LOCAL_RANK", str(group.rank)))
    arm_rank_watchdog(group.rank)
    note(group, f"host group up: world {group.world} (rendezvous: loopback TCP, plain Python)")
    # do the ranks on this host share CPUs (then the Qhull helper budget is divided) or is every rank bound to CPUs of its own?
    qhull_pool.learn_cpu_sharing(group)

    if args.dry_launch:   # control-plane check, no GPU: id broadcast, barrier, max -- what the real run does on the host side
        uid = group.bcast_bytes(bytes(range(128)) if group.rank == 0 else b"")
        group.barrier()
        mx = group.max(float(group.rank + 1))
        ranks = group.allgather_object({"rank": group.rank, "pid": os.getpid(), "id_ok": uid == bytes(range(128)),
                                        "local_world": qhull_pool.local_world()[0], "qhull_helpers": qhull_pool.default_workers()})
        if group.rank == 0:
            os.write(json_fd, (json.dumps({"dry_launch": True, "world": group.world, "max_of_rank_plus_1": mx, "ranks": ranks,
                                           "line_keys_at_n_gt_1": list(N_GT1_KEYS) + ["strong_cfg4"],
                                           "strong_record_keys": list(N_GT1_STRONG_KEYS)}) + "\n").encode())
        group.barrier()
        group.close()
        return

    from same_amd import _lib

    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libsame_hip has no CPU fallback")
    if args.workload == "cfg5":
        return run_cfg5_workload(args, group, json_fd, local_rank)
    strong = args.scaling == "strong"
    ctx = _lib.Context(local_rank % _lib.device_count())          # the dense build's context (one context = one stream)
    # The rest of the step (prune, candidate costs, gather, triangle maps, sweeps) does not read the dense block, so it
    # runs on a context of its own: its small latency-bound kernels fill in beside the 16 ms dense kernel instead of
    # queueing behind it, and the per-step read-back of the sweep waits for that stream only.
    tctx = _lib.Context(ctx.device) if args.tail_stream == "own" else ctx
    comm, transport = make_comm(args, group, tctx)
    if comm is not None and not comm.synchronous and not strong:
        transport += _OVERLAPPED
    env = Env(args, group, ctx, tctx, comm, transport)
    rccl = comm_report(env, np) if comm is not None else None

    prob = Problem(env, args.workload, strong)
    n_ref, T, rows, n_mov, Tr, prob_k, prob_radius = prob.n_ref, prob.T, prob.rows, prob.n_mov, prob.Tr, prob.k, prob.radius
    dD = prob.dD
    # the block the timed loop stores into (the T sweep after it may re-take the block spread over the HBM regions)
    headline_buffer = dict(dD.spread_info) if dD.spread_info else {
        "spread": False, "what": "plain hipMalloc (--spread auto: same_dev_alloc_spread only for store-bound type counts, T <= 12)"}
    use_q32, q_l2 = prob.use_q32, getattr(prob, "q_l2", None)
    note(group, f"inputs resident ({rows} of {n_mov} aligned x {n_ref} ref, {Tr} triangles); gather transport: {transport}")

    note(group, f"warm-up ({args.warmup}) + timed loop ({args.steps} steps)")
    dt, dense_ms = prob.timed_loop(args.steps, args.warmup)
    note(group, f"{args.steps} steps in {dt:.3f} s")

    # ---- N > 1 (or the forced communicator): the line explains itself ------------------------------------------------
    gather = gather_hidden_ms = per_rank = parity_transport = strong_rec = None
    if comm is not None:
        gather_steps = list(prob.gather_ms)
        steps_ng = max(1, min(args.steps, 5))
        note(group, "second loop without the candidate-list gather")
        dt_ng, _ = prob.timed_loop(steps_ng, 1, gather=False)       # the same step without the candidate-list gather
        prob.gather_ms = gather_steps
        gather, gather_hidden_ms = gather_report(prob, dt, args.steps, dt_ng, steps_ng)
        mine = {"rank": group.rank, "device": ctx.device, "pci": ctx.pci_bus_id(), "dense_ms": stats3([m for m, _ in dense_ms]),
                "gather_ms": stats3(gather_steps)}
        every = group.allgather_object(mine)
        per_rank = {"dense_ms_min_mean_max_over_ranks": stats3([r["dense_ms"][1] for r in every]),
                    "dense_ms_by_rank": [r["dense_ms"][1] for r in every],
                    "gather_ms_by_rank": [(r["gather_ms"] or [None, None])[1] for r in every],
                    "device_by_rank": [r["device"] for r in every], "pci_by_rank": [r["pci"] for r in every]}
        prob.step()                                                    # one more step WITH the gather: the lists the check reads
        if group.rank == 0:
            parity_transport = prob.transport_check()
        note(group, f"gather {gather['ms']} ms per step on its stream; step {gather['step_ms_with_gather']:.3f} ms with it, "
                    f"{gather['step_ms_without_gather']:.3f} ms without")

    # ---- after the timed region, rank 0 at N=1: ceilings, operating point, T sweep (all on the warm chip) ------------
    match = prob.dmatch.download((n_mov,), np.int32)
    extras = {}
    if group.rank == 0 and group.world == 1 and not args.no_extras and not strong:
        from same_amd.bench_extras import measure_extras

        extras = measure_extras(args, env, prob, headline_buffer)      # may re-take the cost block spread over the HBM regions
        dD = prob.dD

    # ---- CPU baseline leg (rank 0, N=1, untimed region): see bench_oracle_legs below ------------------------------------
    cpu = None
    parity = parity_transport or ("not checked in this run (the oracle only runs in the cpu_baseline leg: N=1 without "
                                  "--no-cpu-baseline)")
    if group.rank == 0 and group.world == 1 and not args.no_cpu_baseline:
        cpu, parity = dense_oracle_leg(args, group, ctx, prob, match, rebuild_first_chunk=bool(strong or extras))
        if parity_transport:
            parity += "; " + parity_transport

    # ---- N > 1, weak run: BASELINE cfg 4 as an embedded record (ONE problem over the ranks; the resident block is reused) ----
    if comm is not None and not strong and not args.no_strong_record:
        sname = STRONG_OF.get(args.workload, args.workload)
        note(group, f"embedded strong record: {sname} as ONE problem over {group.world} rank(s)")
        prob.close(keep_dense=True)
        sp = Problem(env, sname, True, dense_buf=dD)
        s_steps = max(1, min(args.steps, 5))
        s_dt, s_dense = sp.timed_loop(s_steps, 1)
        s_gather_steps = list(sp.gather_ms)
        s_ng_steps = max(1, min(s_steps, 3))
        s_dt_ng, _ = sp.timed_loop(s_ng_steps, 1, gather=False)
        sp.gather_ms = s_gather_steps
        s_gather, s_hidden = gather_report(sp, s_dt, s_steps, s_dt_ng, s_ng_steps)
        s_every = group.allgather_object(stats3([m for m, _ in s_dense]) if s_dense else None)
        sp.step()
        s_check = sp.transport_check() if group.rank == 0 else None
        if group.rank == 0:
            strong_rec = strong_record(np, group, sp, transport, s_steps, s_dt, s_dense, s_every, s_gather, s_hidden, s_check)
        sp.close(keep_dense=True)
        prob = None

    # ---- BASELINE cfg 5 as an embedded record, IN THIS JOB (its ranks, their contexts, the communicator): the same at 1 rank or 8 ----
    cfg5_rec = None
    # auto: the default workload at any rank count, and every full-size workload at N > 1 (`tiny` is the test suite's: it asks with "on")
    want_cfg5 = args.embed_cfg5 == "on" or (args.embed_cfg5 == "auto" and not strong and not args.no_extras
                                            and (args.workload == "dense100k" or (group.world > 1 and args.workload != "tiny")))
    if want_cfg5:
        from same_amd import bench_cfg5

        note(group, "embedded cfg5 record: the window configuration on this job's ranks")
        try:
            what = transport.replace(_OVERLAPPED, "").replace("pruned lists", "the window merge's seam rows")
            # cpu_baseline is used at one rank: four windows through the oracle, as parity check and CPU figure
            line5 = bench_cfg5.run(args, group, tctx, comm, what, 3, 1, cpu_baseline=cfg5_oracle_leg)
            cfg5_rec = bench_cfg5.record(line5) if group.rank == 0 else None
        except SystemExit:
            raise                                     # a parity failure inside the record is a failure of the line
        except Exception as e:  # noqa: BLE001 -- costs the sub-record, never the main line; every rank raises alike or none does
            cfg5_rec = {"error": f"{type(e).__name__}: {e}"}
        if group.rank == 0:
            said = f"{cfg5_rec['windows_per_s']:.0f} windows/s" if "windows_per_s" in cfg5_rec else cfg5_rec.get("error", "?")
            note(group, "embedded cfg5 record: " + said)

    if group.rank == 0:
        roof, chunk_rows = roofline(args.workload, (n_ref, T, use_q32), dense_ms, extras, headline_buffer, q_l2)
        shape = (n_ref, T, prob_k, prob_radius, rows, n_mov, Tr, use_q32, strong)
        out = dense_line(args, group, shape, dt, roof, chunk_rows, tctx is not ctx, comm, transport, cpu, parity)
        if comm is not None:
            add_multi_rank_parts(out, args, rccl, gather, gather_hidden_ms, per_rank, strong_rec)
        if want_cfg5:   # an error entry, never a lost line: the headline number does not depend on it
            out["cfg5"] = cfg5_rec if cfg5_rec is not None else {"error": "no record"}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    group.barrier()  # rank 0 has finished its spot check / report: tear the communicator down together
    if prob is not None:
        prob.close(keep_dense=True)
    dD.free()
    if comm is not None:
        comm.close()
    if tctx is not ctx:
        tctx.close()
    group.close()


def run_cfg5_workload(args, group, json_fd, local_rank):
    """`--workload cfg5`: the window configuration as the line itself (same_amd/bench_cfg5.py)."""
    from same_amd import _lib, bench_cfg5

    os.environ.setdefault("SAME_HIP_DEVICE", str(local_rank % _lib.device_count()))
    ctx = _lib.default_context()
    # a device all-gather (RCCL; host transport if that fails)
    comm, transport = make_comm(args, group, ctx, what="the window merge's seam rows")
    out = bench_cfg5.run(args, group, ctx, comm, transport, args.steps, args.warmup,
                         cpu_baseline=None if args.no_cpu_baseline else cfg5_oracle_leg)
    if group.rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    group.barrier()
    if comm is not None:
        comm.close()
    group.close()


# ======================================================================================================================
# bench_oracle_legs -- the `cpu_baseline` legs: the ONLY code outside tests/ and __graft_entry__.smoke() that imports oracle/.
# Both run on rank 0 at one rank, after the timed region: the oracle (scalar C port of the reference's arithmetic, one thread)
# works through a bounded sample of the workload -- that time is `cpu_baseline` -- and its outputs double as the parity check of
# what the GPU produced in this very run (a mismatch refuses the line).
# ======================================================================================================================
def dense_oracle_leg(args, group, ctx, prob, match, rebuild_first_chunk):
    """The cpu_baseline leg of the default step: rows [0, S) of the dense build, the prune and the pair costs, and the whole triangle /
    sweep pass, through the oracle (timed) -- and the same outputs of this run's GPU step compared bit for bit.  Also the best-effort
    multi-core CPU lines of SURVEY 8d.  -> (cpu_baseline record, parity text)."""
    import numpy as np

    from oracle import same_oracle as orc

    mov, ref, tris, use_q32 = prob.mov, prob.ref, prob.tris, prob.use_q32
    n_ref, k, radius, rows, n_mov, Tr, ld = prob.n_ref, prob.k, prob.radius, prob.rows, prob.n_mov, prob.Tr, prob.ld
    if rebuild_first_chunk:   # the resident block was reused by the probes: rebuild this rank's first chunk for the check
        ctx.check(prob.dense_launch(prob.rb, min(prob.rb + prob.chunk_rows, prob.re)), "dense")
        ctx.sync()
    S = min(args.cpu_sample_rows, rows, prob.chunk_rows)
    c0 = time.perf_counter()
    want_dense = orc.dense_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], 1.0, 0, S)
    exact_dense = want_dense
    if use_q32:   # the fixed-point build is checked against ITS twin bit for bit, and against the exact costs within the tolerance
        want_dense = orc.dense_cost_q32(mov["types"], ref["types"], mov["xy"], ref["xy"], 1.0, prob.q_off, prob.q_l2, 0, S)
        if float(np.max(np.abs(want_dense - exact_dense) / exact_dense)) > 1e-6:
            raise SystemExit("fixed-point dense costs are outside 1e-6 relative of the exact ones: refusing to report a number")
    oi, _, _ = orc.knn_prune(mov["xy"], ref["xy"], radius, k, 0, S)
    rr, cc = np.nonzero(oi >= 0)
    want_pc = orc.pair_cost_arrays(mov["types"], ref["types"], mov["xy"], ref["xy"], np.column_stack((rr, oi[rr, cc])), 1.0)
    c1 = time.perf_counter()
    orc.tri_classify(mov["xy"], tris, radius, 15, mov["cell_type"])
    orc.tri_sign_weight(mov["xy"], mov["size"], tris)
    och, oviol, _ = orc.orient_sweep(tris, prob.sign0, ref["xy"], match)
    orc.xyorder_sweep(mov["xy"], ref["xy"], tris, match)
    orc.area_flip(mov["xy"], ref["xy"], tris, match)
    c2 = time.perf_counter()
    t_cpu = (c1 - c0) + (c2 - c1) * S / n_mov
    # parity of this run's GPU outputs with what the baseline just computed
    ok = True
    for i in np.random.default_rng(0).choice(S, min(8, S), replace=False):
        ok &= bool(np.array_equal(prob.dD.download((n_ref,), np.float64, offset_bytes=int(i) * ld * 8), want_dense[i]))
    ok &= bool(np.array_equal(prob.didx.download((S, k), np.int32), oi))
    got_pc = prob.dcost.download((S, k), np.float64)
    ok &= bool(np.array_equal(got_pc[rr, cc], want_pc))
    ok &= (och == prob.last["checked"]) and bool(np.array_equal(oviol, prob.last["viol"]))
    if not ok:
        raise SystemExit("bench outputs differ from the oracle: refusing to report a number")
    parity = f"dense rows, pruned lists and pair costs of rows [0,{S}) and the orientation sweep equal the oracle bit-for-bit"
    if use_q32:
        parity = (f"dense rows of [0,{S}) equal the fixed-point build's oracle twin bit-for-bit and are within 1e-6 relative of the "
                  f"exact fp64 costs on all {S} x {n_ref} pairs; pruned lists, pair costs and the orientation sweep equal the oracle "
                  "bit-for-bit")
    note(group, f"cpu baseline sample done ({t_cpu:.1f} s), parity check passed")
    del want_dense
    # best-effort multi-core CPU lines: the dense sample split over host threads (ctypes releases the GIL),
    # and the radius query + top-k of the prune with scipy's cKDTree on all cores (src/utils.py:714,722 with workers=-1)
    from concurrent.futures import ThreadPoolExecutor

    from scipy.spatial import cKDTree

    nthr = max(1, min(16, os.cpu_count() or 1))
    cuts = np.linspace(0, S, nthr + 1).astype(int)
    m0 = time.perf_counter()
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(lambda be: orc.dense_cost(mov["types"], ref["types"], mov["xy"], ref["xy"], 1.0, int(be[0]), int(be[1])),
                    zip(cuts[:-1], cuts[1:])))
    t_mt = time.perf_counter() - m0
    k0 = time.perf_counter()
    tree = cKDTree(ref["xy"])
    balls = tree.query_ball_point(mov["xy"][:S], radius, workers=-1)
    kd_pairs = 0
    for i, b in enumerate(balls):
        b = np.asarray(b, dtype=np.int64)
        d = np.linalg.norm(ref["xy"][b] - mov["xy"][i], axis=1)
        kd_pairs += len(b[np.argsort(d)[:k]])
    t_kd = time.perf_counter() - k0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    cpu = {"value": S * n_ref / t_cpu, "unit": "cell-pairs/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
           "host_cpus": os.cpu_count(),
           "dense_only_threaded": {"value": S * n_ref / t_mt, "unit": "cell-pairs/s", "cores": nthr,
                                   "sample": f"dense cost of the same {S} rows split over {nthr} host threads"},
           "knn_ckdtree_threaded": {"value": S * n_ref / t_kd, "unit": "dense-equivalent cell-pairs/s", "cores": os.cpu_count(),
                                    "sample": f"scipy cKDTree(refs) build + query_ball_point(rows [0,{S}), r={radius:g}, workers=-1) + "
                                              f"per-row norm/argsort top-{k} as src/utils.py:722-728 ({kd_pairs} pairs kept, "
                                              f"{t_kd:.2f} s)"},
           "sample": f"rows [0,{S}) of {n_mov} x {n_ref} refs: dense cost + knn prune + pair costs "
                     f"({c1 - c0:.2f} s) plus the full {Tr}-triangle classify/sign/sweep pass scaled by {S}/{n_mov} "
                     f"({c2 - c1:.3f} s unscaled); oracle/same_oracle.c, gcc -O2, 1 thread of {os.cpu_count()}",
           "reference_note": "the reference itself (pure Python/pandas) cannot travel to this box; measured in the survey "
                             "container (BASELINE.md section 3, 1 of 8 vCPU): 1.0-1.3e3 pairs/s pair-cost loop, 6.5e3 triangles/s "
                             "filter, 2.7e5 triangles/s lazy sweep, 1.6e7 dense-equivalent cell-pairs/s KNN at 10k x 10k"}
    return cpu, parity


def cfg5_oracle_leg(st):
    """The cpu_baseline leg of the cfg 5 step (same_amd/bench_cfg5.py calls it on rank 0 at one rank): four windows through the CPU
    oracle -- timed: that is `cpu_baseline` -- and the same windows on the GPU, through prepare_same_inputs and through the timed
    device-resident path, compared bit for bit."""
    import numpy as np

    import same_amd

    plan, my_plan, r_df, m_df, cols = st["plan"], st["my_plan"], st["r_df"], st["m_df"], st["cols"]
    op, on_device = st["op"], st["on_device"]
    from scipy.spatial import Delaunay

    from oracle import same_oracle as orc

    sample = [w for w in my_plan if w["n_mov"] > 1000][:4] or my_plan[:1]
    t_cpu, done_pairs, checks = 0.0, 0, []
    for w in sample:
        c0 = time.perf_counter()                 # the oracle's part of this window only: the GPU re-runs below are not the CPU's time
        x0, x1, y0, y1 = w["box"]
        rs, ms = same_amd.subset_data(r_df, x0, x1, y0, y1), same_amd.subset_data(m_df, x0, x1, y0, y1)
        na, nr, pairs = orc.find_knn_within_radius(ms, rs, 25, 8)
        pairs = np.asarray(pairs, dtype=np.int64)
        axy, rxy = na[["X", "Y"]].to_numpy(), nr[["X", "Y"]].to_numpy()
        c32 = orc.pair_cost_arrays(na[cols].to_numpy(), nr[cols].to_numpy(), axy, rxy, pairs, 1.0, dtype=np.float32)
        kept = orc.filter_triangles_by_radius(axy, Delaunay(axy).simplices, 25, aligned_df=na, ignore_same_type_triangles=True,
                                              min_angle_deg=15)
        tri = np.asarray(kept, dtype=np.int64).reshape(-1, 3)
        signs = orc.source_signs(na, tri)
        kw = dict(valid_pairs=[tuple(p) for p in pairs.tolist()], costs=c32.astype(np.float64), n_aligned=len(na), n_ref=len(nr),
                  aligned_sizes=na["size"].to_numpy(dtype=float), no_match_penalty=100, max_matches=1, init_method="greedy", verbose=False)
        och, _ = orc.compute_mip_start_pairs(**kw)
        xo = np.zeros(len(pairs))
        xo[[c[2] for c in och]] = 1.0
        ochecked, oviol = orc.lazy_orientation_sweep(xo, pairs, tri, signs, rxy, len(na))
        done_pairs += w["n_mov"] * w["n_ref"]
        t_cpu += time.perf_counter() - c0
        # the same window on the GPU, compared
        prep = same_amd.prepare_same_inputs(rs, ms, cols, optim_params=op, verbose=False)
        ok = (np.array_equal(np.asarray(prep.valid_pairs, dtype=np.int64), pairs)
              and np.array_equal(np.array(prep.costs).astype(np.float32), c32)
              and np.array_equal(np.asarray(prep.aligned_delaunay, dtype=np.int64).reshape(-1, 3), tri)
              and list(prep.source_signs) == list(signs))
        gch, _ = same_amd.compute_mip_start_pairs(**dict(kw, valid_pairs=prep.valid_pairs, costs=prep.costs))
        sw = same_amd.LazyOrientationSweep(prep.valid_pairs, tri, prep.source_signs, rxy, prep.n_aligned)
        gchecked, gviol, _ = sw.sweep(xo)
        sw.bound.close()
        as_tuples = lambda rows: [tuple(int(q) for q in v) for v in rows]
        ok = ok and gch == och and gchecked == ochecked and as_tuples(gviol) == as_tuples(oviol)
        checks.append((w, na, nr, pairs, c32, tri, signs, och, ochecked, oviol))
        if not ok:
            raise SystemExit("cfg5 window outputs differ from the oracle: refusing to report a number")
    if on_device:       # and what the timed path itself computes for these windows (csrc/window.hip), as the API hands it on
        got = [p for _w, p in same_amd.iter_prepared_windows(r_df, m_df, cols, [c[0] for c in checks], optim_params=op, pipeline="device")]
        for (w, na, nr, pairs, c32, tri, signs, och, ochecked, oviol), prep in zip(checks, got):
            nr_rows, match_o = nr["Cell_Num_Old"].to_numpy(), np.full(len(na), -1, np.int64)
            for hit in och:
                match_o[hit[0]] = nr_rows[hit[1]]
            if isinstance(prep, Exception):
                raise SystemExit(f"cfg5: the device-resident path refused a window the oracle ran: {prep}")
            dw = prep.device
            ok = (np.array_equal(prep.rows_m, na["Cell_Num_Old"].to_numpy())
                  and np.array_equal(prep.rows_r, nr_rows) and np.array_equal(np.asarray(prep.valid_pairs, dtype=np.int64), pairs)
                  and np.array_equal(prep.costs_array.astype(np.float32), c32) and np.array_equal(prep.triangles_array, tri)
                  and np.array_equal(prep.signs_array, np.asarray(signs, dtype=np.float64))
                  and np.array_equal(dw.match_row, match_o) and dw.stats["checked"] == ochecked and dw.stats["flipped"] == len(oviol))
            if not ok:
                raise SystemExit("cfg5 window outputs of the device-resident path differ from the oracle: refusing to report a number")
    parity = (f"{len(sample)} windows: pairs, fp32 pair costs, kept triangles, source signs, greedy start and the orientation sweep under "
              "it equal the oracle bit-for-bit"
              + (" -- through prepare_same_inputs and through the device-resident window path" if on_device else ""))
    cpu = {"value": done_pairs / t_cpu, "unit": "cell-pairs/s", "cores": 1, "kind": "port", "host_cpus": os.cpu_count(),
           "sample": f"{len(sample)} of {len(plan)} windows (prune, fp32 pair costs, Qhull + triangle filter, signs, greedy start, "
                     f"orientation sweep) through oracle/same_oracle.{{c,py}} in {t_cpu:.2f} s, 1 thread (frame subsetting included; the "
                     "GPU re-runs of the same windows for the comparison are not in this time)",
           "reference_note": "the reference's own loop (src/same.py:507-593) also solves a MIP per window, which has no counterpart on "
                             "this box"}
    return cpu, parity


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        from same_amd.bench_launch import launch

        sys.exit(launch(args, __file__))
    run_rank(args)


if __name__ == "__main__":
    main()
