#!/bin/bash
set -o pipefail
tag=${1:-visit6}
out=gpurun_out/$tag
mkdir -p $out
echo "== q32 tests" && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "q32 or dense_cost_all_T" > $out/pytest_q32.log 2>&1; rc=$?; tail -4 $out/pytest_q32.log; [ $rc -eq 0 ] || exit $rc
echo "== q32 probe" 
for c in 4 2; do echo "-- SAME_DENSE_Q32_CPL=$c"; SAME_DENSE_Q32_CPL=$c timeout -k 10 300 python3 tools/dense_probe.py 100000 0,3,8,20,32 q32 2>&1 | tee -a $out/dense_q32.log; done
echo "-- fp64 reference kernel, same box"; timeout -k 10 300 python3 tools/dense_probe.py 100000 0,8,20 2>&1 | tee -a $out/dense_q32.log
echo "== per-kernel stats, one stream" && cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --tail-stream shared > $OLDPWD/$out/stats_bench_shared.json 2> $OLDPWD/$out/stats.err || { tail -20 $OLDPWD/$out/stats.err; exit 1; }
cd $OLDPWD; f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp "$f" $out/bench_kernel_stats_shared.csv; t=$(find $out/stats -name "*kernel_trace.csv" | head -1); cp "$t" $out/kernel_trace_shared.csv; rm -rf $out/stats
python3 tools/kernel_table.py $out/bench_kernel_stats_shared.csv $out/stats_bench_shared.json $out/kernel_trace_shared.csv > $out/kernel_roofline.md 2>&1 || tail -5 $out/kernel_roofline.md
rm -f $out/kernel_trace_shared.csv
head -40 $out/kernel_roofline.md
echo "== done"
