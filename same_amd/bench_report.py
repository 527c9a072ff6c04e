"""How bench.py's JSON line is put together from what a run measured: the `roofline` object of the dense kernel with its notes, the
embedded one-problem record (cfg 4) and the line itself.  Arithmetic on numbers the run already has -- no GPU call, no oracle."""
import json
import os

from .bench_common import HBM_PEAK_GBS, ROOT, baseline_metric, stats3
from .bench_problem import STRONG_OF, dense_kernel_label

FP64_ISSUE_PEAK_T = 39.3               # T lane-instructions/s: the 78.6 TFLOP/s fp64 vector spec counts an FMA as two
SIMDS, FP64_LANES_PER_CLK = 1024, 16   # 256 CUs x 4 SIMDs; a wave64 fp64 instruction occupies its SIMD for 4 cycles

# what the line carries at N > 1 on top of the N = 1 keys, so that the one 8-GPU run explains itself (checked before the line is
# written; listed by --dry-launch so the CPU suite can hold the contract)
N_GT1_KEYS = ("rccl", "gather", "gather_hidden_ms", "per_rank_dense_ms", "per_rank", "cfg5")
N_GT1_STRONG_KEYS = ("config", "scaling", "value", "ms_per_step", "dense_kernel_ms", "per_rank_dense_ms", "gather", "gather_hidden_ms",
                     "parity_spot_check")

_OVERLAPPED = " (overlapped on a second stream)"


def strong_record(np, group, sp, transport, s_steps, s_dt, s_dense, s_every, s_gather, s_hidden, s_check):
    """The embedded record of ONE problem over the ranks (BASELINE cfg 4): its own value, its dense launches, its gather."""
    ms_launch = float(np.mean([m for m, _ in s_dense])) if s_dense else None
    rows_launch = float(np.mean([r for _, r in s_dense])) if s_dense else 0.0
    s_bytes = 8.0 * sp.n_ref * rows_launch + 8.0 * (sp.T + 2) * (sp.n_ref + rows_launch)
    gbs = (s_bytes / (ms_launch * 1e-3) / 1e9) if ms_launch else None
    sweeps = ", sweeps over triangle blocks (flag all-gather + counter all-reduce)" if sp.sharded is not None else ""
    return {"config": {"workload": sp.workload_text(),
                       "parallelism": f"aligned-row blocks x{group.world}, {transport.replace(_OVERLAPPED, '')}" + sweeps},
            "scaling": "strong", "value": float(sp.n_ref) * sp.n_mov * s_steps / s_dt, "unit": "cell-pairs/s", "steps": s_steps,
            "warmup": 1, "ms_per_step": s_dt / s_steps * 1e3, "rows_this_rank": sp.rows, "dense_launches_per_step": sp.n_chunks,
            "dense_kernel_ms": ms_launch, "dense_GBs": gbs, "dense_frac_of_hbm_peak": (gbs / HBM_PEAK_GBS) if gbs else None,
            "per_rank_dense_ms": stats3([e[1] for e in s_every if e]), "dense_ms_by_rank": [e[1] if e else None for e in s_every],
            "gather": s_gather, "gather_hidden_ms": s_hidden, "parity_spot_check": s_check,
            "sweep_outputs": {"checked": int(sp.last["checked"]), "flipped": int(len(sp.last["viol"]))}}


def _traffic(workload):
    """HBM bytes per launch of the dense kernel from the committed PMC passes (profiles/traffic.json), or (None, None)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    try:
        tj = json.load(open(path))
        return tj.get(workload), tj.get("_source")
    except Exception:
        return None, None


def _buffer_note(si):
    checked = (f"verified: one store over the finished range ran at {si['final_store_gbps']} GB/s against a same-region level of "
               f"{si['same_region_level_gbps']}" if si.get("verified")
               else "NOT verified: a store over the finished range did not reach the fast level")
    return (f"the cost block is {si['chunks_gib']} GiB mapped round-robin from the card's three HBM regions ({si['per_region']} GiB per "
            f"region, {si['straddling']} straddling; found by timed stores in {si['seconds']:.1f} s before the timed region; {checked}): "
            "a streaming store confined to one region runs ~20 % below one spread over them")


def _ceiling_notes(roof, c, achieved, T):
    msg = []
    c["frac_of_T0_store_rate"] = achieved / c["same_kernel_T0_store_only_GBs"]
    roof["measured_ceilings"] = c
    if c.get("device_copy_GBs"):
        roof["frac_of_measured_copy_bw"] = achieved / c["device_copy_GBs"]
        msg.append(f"a device-to-device copy on this box moves {c['device_copy_GBs']:.0f} GB/s (read + written): the kernel's "
                   f"{achieved:.0f} GB/s is {roof['frac_of_measured_copy_bw']:.2f} of that")
    if "plain_hipMalloc_buffer" in c:
        pb = c["plain_hipMalloc_buffer"]
        msg.append(f"a plain hipMalloc buffer of the same size in this process: T=0 store {pb['same_kernel_T0_store_only_GBs']:.0f} "
                   f"GB/s, hipMemsetAsync {pb['hipMemsetAsync_GBs']:.0f} GB/s"
                   + (f", this kernel {pb['bench_kernel_ms']:.2f} ms" if pb.get("bench_kernel_ms") else ""))
    msg.append(f"on this box the same kernel with T=0 (same stores, 5 instead of {2 * T + 5} VALU ops per output) streams "
               f"{c['same_kernel_T0_store_only_GBs']:.0f} GB/s and hipMemsetAsync {c['hipMemsetAsync_GBs']:.0f} GB/s, so the T={T} "
               f"build runs at {c['frac_of_T0_store_rate']:.2f} of its own store-only rate")
    return msg


def _telemetry_note(roof, t, T, t_dense, lane_instr, use_q32):
    """Board power / clock while the kernel looped; prices the kernel's fp64 instructions at the clock the board held."""
    roof["telemetry"] = t
    if not (t.get("available") and t.get("power")):
        return "board power / clock could not be read from sysfs on this box"
    clk = t.get("sclk_steady") or t.get("sclk_hwmon") or t.get("sclk_dpm")
    pw = t.get("power_steady") or t["power"]
    launch_ms = t.get("dense_ms_during_window") or t_dense * 1e3
    if clk and not use_q32:
        # every fp64 VALU instruction of a wave64 holds its SIMD for 4 cycles: the time the launch's instructions need at the clock the
        # board held while this kernel looped, and the share of the launch they fill
        floor_ms = lane_instr / (SIMDS * FP64_LANES_PER_CLK * clk["mean"] * 1e6) * 1e3
        roof["valu_floor_ms_at_held_clock"] = floor_ms
        roof["valu_busy_frac"] = floor_ms / launch_ms
        roof["frac_of_binding_ceiling"] = (floor_ms / (t_dense * 1e3)) if T >= 12 else roof["frac"]
        roof["held_clock_mhz"], roof["board_power_w"], roof["board_power_cap_w"] = clk["mean"], pw["mean"], t.get("power_cap_w")
    crit = t.get("temperature_crit_c") or {}
    temps = "".join(f", {n} {v['mean']:.0f} C" + (f" (critical {crit[n]:.0f})" if crit.get(n) else "")
                    for n, v in (t.get("temperature_steady") or {}).items() if v)
    text = (f"while the kernel looped the board drew {pw['mean']:.0f} W in steady state (max {t['power']['max']:.0f} W"
            + (f", cap {t['power_cap_w']:.0f} W" if t.get("power_cap_w") else "") + ")"
            + (f" at a shader clock of {clk['mean']:.0f} MHz (min {clk['min']:.0f})" if clk else "") + temps)
    if clk and not use_q32:
        text += (f"; at that clock the {2 * T + 5}-instruction fp64 VALU floor is {roof['valu_floor_ms_at_held_clock']:.1f} ms of the "
                 f"{launch_ms:.1f} ms launch (VALU busy {roof['valu_busy_frac']:.2f}): the bound of this kernel is fp64 issue under the "
                 "board power cap, not HBM")
    return text


def roofline(workload, prob_shape, dense_ms, extras, headline_buffer, q_l2=None):
    """The `roofline` object of the dense kernel: algorithmic bytes s*N_r*rows + s*(T+2)*(N_r+rows) over its mean launch
    time (HIP events on the stream it runs on).  prob_shape = (n_ref, T, use_q32); dense_ms = [(ms, rows)] per timed launch."""
    import numpy as np

    n_ref, T, use_q32 = prob_shape
    t_dense = float(np.mean([m for m, _ in dense_ms])) * 1e-3               # mean launch duration
    rows_launch = float(np.mean([r for _, r in dense_ms]))                   # rows one launch covers (== rows unless chunked)
    dense_bytes = 8.0 * n_ref * rows_launch + 8.0 * (T + 2) * (n_ref + rows_launch)
    traffic, traffic_src = _traffic(workload)
    lane_instr = (2 * T + 5) * float(n_ref) * rows_launch                    # fp64 VALU lane-instructions of one launch
    valu_rate = lane_instr / t_dense / 1e12
    if use_q32:
        dense_bytes = 8.0 * n_ref * rows_launch + (4.0 * T + 16.0) * (n_ref + rows_launch)
        traffic, traffic_src = None, "not collected for the fixed-point build"
    achieved = dense_bytes / t_dense / 1e9
    kernel = (f"dense_cost_q32_kernel<{T},double> (opt-in fixed-point build, --dense q32)" if use_q32 else dense_kernel_label("f64", T))
    binding = None
    if not use_q32:
        # `bound` names the roofline BASELINE.json prices the kernel against; what actually limits the T = 20 fp64 kernel is said here
        binding = ("fp64 VALU issue under the board power cap (T >= ~12: 2T+5 fp64 lane-instructions per 8-byte output; HBM only binds "
                   "at the reference datasets' T = 3 / 5 / 8)" if T >= 12 else "hbm")
    roof = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src or "profiles/traffic.json (separate rocprofv3 --pmc passes: WRITE_SIZE + 2*FETCH_SIZE)",
            "algorithmic_bytes_per_launch": dense_bytes, "kernel_ms": t_dense * 1e3, "launches_timed": len(dense_ms),
            # secondary ceiling: (2T+5) fp64 VALU lane-instructions per output against the vector issue peak
            "valu_fp64": None if use_q32 else {"lane_instr_per_output": 2 * T + 5, "achieved_Tinstr_s": valu_rate,
                                               "peak_Tinstr_s": FP64_ISSUE_PEAK_T, "frac": valu_rate / FP64_ISSUE_PEAK_T},
            "valu_floor_ms_at_held_clock": None, "valu_busy_frac": None, "frac_of_measured_copy_bw": None, "binding": binding,
            "frac_of_binding_ceiling": None, "target_frac": 0.70, "target_met": bool(achieved / HBM_PEAK_GBS >= 0.70),
            "valu_floor_note": "fp64 VALU floor of this kernel: 44 lane-instructions per output = 11.2 ms per launch at the 2.4 GHz spec "
                               "clock, so the best possible HBM fraction is 0.89 at spec clock and about 0.64 at the ~1.7 GHz the board "
                               "holds under its power cap"}
    msg = ["frac is against the 8.0 TB/s HBM spec as BASELINE.json asks (`bound`); `binding` is what limits this kernel on this board "
           "and `frac_of_binding_ceiling` = valu_floor_ms_at_held_clock / kernel_ms is how close the launch is to THAT ceiling"]
    if use_q32:
        msg.append("THIS LINE WAS RUN WITH --dense q32: the step's dense build is the opt-in fixed-point kernel (exact integer type sums "
                   f"on a 2^-{q_l2 if q_l2 is not None else '?'} grid, sums too small for the grid recomputed in fp64: every output within "
                   "1e-6 relative of the reference's fp64 cost, which is BASELINE.json's tolerance) -- not the reference's arithmetic; "
                   "the default run reports the bit-exact kernel")
    roof["output_buffer"] = headline_buffer
    if headline_buffer.get("spread"):
        msg.append(_buffer_note(headline_buffer))
    if "ceilings" in extras:
        msg += _ceiling_notes(roof, extras["ceilings"], achieved, T)
    if "telemetry" in extras:
        msg.append(_telemetry_note(roof, extras["telemetry"], T, t_dense, lane_instr, use_q32))
    for key in ("pruned_path", "triangle_maps_and_sweeps", "realistic_matching"):
        if key in extras:
            roof[key] = extras[key]
    if "sweep" in extras:
        roof["sweep"] = extras["sweep"]
        msg.append("sweep = same measurement at other type counts (the reference's datasets have T = 3, 5, 8)")
        ctl = [e for e in extras["sweep"] if e.get("opt_in")]
        if ctl:
            msg.append(f"control: the opt-in fixed-point build writes the same {dense_bytes / 1e9:.0f} GB with integer v_sad_u32 in place "
                       "of the fp64 adds, every output within 1e-6 relative of this kernel's (BASELINE's own tolerance for fp64 costs; max "
                       f"{ctl[0]['max_rel_diff_vs_exact_on_16_rows']:.1e} on 16 sampled rows), in {ctl[0]['ms']:.2f} ms = "
                       f"{ctl[0]['frac']:.3f} of the HBM spec -- the gap to this kernel is the energy of the fp64 arithmetic, not memory "
                       "traffic")
    roof["note"] = "; ".join(msg)
    return roof, int(rows_launch)


def dense_line(args, group, shape, dt, roof, chunk_rows, two_streams, comm, transport, cpu, parity):
    """The line of the default workload (and of cfg2 / cfg4 / tiny) without its N > 1 and cfg 5 parts.
    shape = (n_ref, T, k, radius, rows, n_mov, Tr, use_q32, strong)."""
    n_ref, T, k, radius, rows, n_mov, Tr, use_q32, strong = shape
    total_rows = n_mov if strong else rows * group.world
    what = (f"ONE problem of {n_mov} aligned x {n_ref} ref cells, aligned-row blocks and triangle blocks over {group.world} rank(s), "
            f"dense build in {chunk_rows}-row chunks" if strong else f"{rows} aligned x {n_ref} ref cells per GPU")
    arithmetic = "fixed-point (every output within 1e-6 relative of the fp64 one) " if use_q32 else "fp64 "
    workload = (f"{args.workload}: {what}, T={T} type cols, {arithmetic}dense L1 cost + r={radius:g}/k={k} KNN prune + pair costs + {Tr} "
                "Delaunay triangles classify/sign + orientation / XY-order / area-flip sweeps")
    sweeps = ", sweeps over triangle blocks (flag all-gather + counter all-reduce)" if (strong and comm is not None) else ""
    return {"metric": baseline_metric(), "value": float(n_ref) * total_rows * args.steps / dt, "unit": "cell-pairs/s",
            "n_gpus": group.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32+f64" if use_q32 else "f64",
            "data": "synthetic",
            "config": {"workload": workload,
                       "streams": ("dense build on one stream, prune / costs / triangle maps / sweeps on a second (own context)"
                                   if two_streams else "one stream, in order"),
                       "parallelism": f"aligned-row blocks x{group.world}" + (", " + transport if comm is not None else "") + sweeps},
            "roofline": roof, "cpu_baseline": cpu, "parity_spot_check": parity}


def add_multi_rank_parts(out, args, rccl, gather, gather_hidden_ms, per_rank, strong_rec):
    """What the line carries at N > 1 so that the one 8-GPU run explains itself; refuses a line that lacks any of it."""
    out["rccl"], out["gather"], out["gather_hidden_ms"] = rccl, gather, gather_hidden_ms
    out["gather_hidden_ms_means"] = ("ms_per_step of the timed loop minus ms_per_step of a second, shorter loop of the same step without "
                                     "the candidate-list all-gather: what the gather costs the step (about 0 = fully hidden behind the "
                                     "dense build)")
    out["per_rank_dense_ms"] = per_rank["dense_ms_min_mean_max_over_ranks"]
    out["per_rank"] = per_rank
    if strong_rec is not None:
        out["strong_cfg4" if STRONG_OF.get(args.workload) == "cfg4" else "strong_record"] = strong_rec
        missing = [k for k in N_GT1_STRONG_KEYS if k not in strong_rec]
        if missing:
            raise SystemExit(f"the embedded strong record lacks {missing}")
    missing = [k for k in N_GT1_KEYS if out.get(k) is None and k not in ("gather_hidden_ms", "cfg5")]   # cfg5 joins afterwards
    if missing:
        raise SystemExit(f"the N > 1 line lacks {missing}")
