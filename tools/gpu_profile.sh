#!/bin/bash
# rocprofv3 passes over the bench command (kernel stats; WRITE_SIZE and FETCH_SIZE in their own passes), summaries under gpurun_out/<tag>/
set -o pipefail
tag=${1:-prof}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2"
echo "== kernel trace + stats" && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > $out/stats_bench.json 2> $out/stats.err || { tail -20 $out/stats.err; exit 1; }
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp "$f" $out/bench_kernel_stats.csv; head -8 $out/bench_kernel_stats.csv | cut -c1-220
for c in WRITE_SIZE FETCH_SIZE; do
  echo "== pmc $c" && timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- $cmd > $out/pmc_${c}_bench.json 2> $out/pmc_$c.err || { tail -20 $out/pmc_$c.err; exit 1; }
  python3 $root/tools/pmc_summary.py $out/pmc_$c $out/pmc_${c}_summary.csv | grep -i "dense\|Kernel_Name" | cut -c1-200
done
echo "== pmc SQ" && timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_SQ -- $cmd > $out/pmc_SQ_bench.json 2> $out/pmc_SQ.err || { tail -20 $out/pmc_SQ.err; exit 1; }
python3 $root/tools/pmc_summary.py $out/pmc_SQ $out/pmc_SQ_summary.csv | grep -i "dense\|Kernel_Name" | cut -c1-200
rm -rf $out/stats $out/pmc_WRITE_SIZE $out/pmc_FETCH_SIZE $out/pmc_SQ   # raw traces are large; the summaries are what is kept
echo "== done"
