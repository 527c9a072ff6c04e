"""Per-kernel roofline table for the bench.py workload from a rocprofv3 --kernel-trace --stats summary.

Algorithmic (touched) bytes per launch follow SURVEY 8(d) / DESIGN.md; the dense kernel is bounded by
HBM, the others are gather/latency-bound and are reported as GB/s of touched bytes plus absolute time."""
import json, sys
import pandas as pd

stats = pd.read_csv(sys.argv[1])
line = json.load(open(sys.argv[2]))
n_r = rows = 100_000; T = 20; k = 32
Tr = int(line["config"]["workload"].split(" Delaunay")[0].split("+ ")[-1])
bytes_of = {
    "dense_cost_kernel": 8 * n_r * rows + 8 * (T + 2) * (n_r + rows),
    "knn_grid_kernel": 16 * n_r * 3 + 16 * rows + 4 * k * rows + 4 * rows, "first_candidate_kernel": 4 * rows * k // 8 + 4 * rows,
    "padded_cost_kernel": rows * k * (2 * 8 * (T + 2) + 4 + 8), "padded_cost_lds_kernel": rows * k * (2 * 8 * (T + 2) + 4 + 8),
    "bbox_kernel": 16 * n_r, "grid_count_kernel": 16 * n_r + 8 * n_r, "grid_scatter_kernel": 16 * n_r + 28 * n_r,
    "grid_scan_kernel": 8 * 16641,
    "tri_classify_kernel": Tr * (12 + 48 + 12 + 1 + 16), "tri_sign_weight_kernel": Tr * (12 + 48 + 24 + 1 + 8),
    "orient_flag_kernel": Tr * 74, "compact_mask_kernel": Tr // 8 + 4 * 8000,
    "xyorder_kernel": Tr * (12 + 3 * 40) + Tr * 4 + rows, "area_flip_kernel": Tr * (12 + 12 + 96 + 16 + 4),
}
rowsout = []
for _, r in stats.iterrows():
    name = next((k_ for k_ in bytes_of if k_ in r["Name"]), None)
    if not name:
        continue
    if "dense_cost_kernel<double, 0," in r["Name"]:   # the T=0 launches bench.py makes to measure the store-only ceiling
        ms = r["AverageNs"] / 1e6
        gbs = 8 * n_r * rows / r["AverageNs"]
        rowsout.append(("dense_cost_kernel<T=0> (store-only ceiling probe)", int(r["Calls"]), ms, 8 * n_r * rows / 1e6, gbs, 100 * gbs / 8000, r["Percentage"]))
        continue
    ms = r["AverageNs"] / 1e6
    gbs = bytes_of[name] / r["AverageNs"]
    rowsout.append((name, int(r["Calls"]), ms, bytes_of[name] / 1e6, gbs, 100 * gbs / 8000, r["Percentage"]))
print("| kernel | launches | avg ms | algorithmic MB / launch | GB/s | % of 8 TB/s | % of GPU time |")
print("|---|---|---|---|---|---|---|")
for n, c, ms, mb, gbs, pct, share in sorted(rowsout, key=lambda x: -x[6]):
    print(f"| `{n}` | {c} | {ms:.4f} | {mb:.2f} | {gbs:.0f} | {pct:.1f} | {share:.2f} |")
print(f"\nbench line: {line['ms_per_step']:.2f} ms/step, {line['value']:.3e} cell-pairs/s, dense kernel {line['roofline']['kernel_ms']:.2f} ms "
      f"(live HIP-event mean over the timed steps), frac {line['roofline']['frac']:.3f}")

if len(sys.argv) > 3:   # the kernel trace of the same run: per-launch durations (the --stats average includes bench.py's warm-up steps)
    tr = pd.read_csv(sys.argv[3]).sort_values("Start_Timestamp")
    dd = tr[tr["Kernel_Name"].str.contains("dense_cost_kernel<double, 20")]
    dur = (dd["End_Timestamp"] - dd["Start_Timestamp"]) / 1e6
    w = int(line["warmup"])
    print(f"\nper-launch durations of the dense kernel in this trace (ms): {dur.round(2).tolist()}; the first {w} are bench.py's warm-up steps "
          f"(first touch of the 80 GB block, clocks ramping), the {len(dur) - w} timed launches average {dur.iloc[w:].mean():.2f} ms under the "
          f"profiler vs the live HIP-event mean printed above.")
