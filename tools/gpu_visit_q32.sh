#!/bin/bash
# Second GPU visit of a tree: the opt-in fixed-point step, the dense probes on a spread buffer, the energy table, and the
# instruction counters of the fixed-point kernel.  Usage (on the box, from the repo root): bash tools/gpu_visit_q32.sh <tag>
set -o pipefail
tag=${1:-visit_q32}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
echo "== bench --dense q32" && timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --dense q32 > $out/bench_q32.json 2> $out/bench_q32.err || { tail -20 $out/bench_q32.err; exit 1; }
tail -2 $out/bench_q32.err
echo "== bench default (same box)" && timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
echo "== dense probes" && timeout -k 10 300 python3 tools/dense_probe.py 100000 0,3,8,12,16,20,24,32 > $out/dense_probe_f64.log 2>&1 && timeout -k 10 300 python3 tools/dense_probe.py 100000 0,8,20,32 q32 > $out/dense_probe_q32.log 2>&1 && timeout -k 10 300 python3 tools/dense_probe.py 100000 0,8,20 f32 > $out/dense_probe_f32.log 2>&1 || { tail -5 $out/dense_probe_*.log; exit 1; }
tail -n 4 $out/dense_probe_f64.log $out/dense_probe_q32.log $out/dense_probe_f32.log
echo "== energy table" && timeout -k 10 400 python3 tools/energy_table.py > $out/energy_table.md 2> $out/energy_table.err || { tail -5 $out/energy_table.err; exit 1; }
cat $out/energy_table.md
cd /tmp && export TMPDIR=/tmp
echo "== pmc SQ, fixed-point kernel" && timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_SQ_q32 -- python3 $root/tools/dense_probe.py 100000 20 q32 > $out/pmc_SQ_q32.log 2> $out/pmc_SQ_q32.err || { tail -20 $out/pmc_SQ_q32.err; exit 1; }
python3 $root/tools/pmc_summary.py $out/pmc_SQ_q32 $out/pmc_SQ_q32_summary.csv | grep -i "q32\|Kernel_Name" | cut -c1-200
rm -rf $out/pmc_SQ_q32
echo "== done"
