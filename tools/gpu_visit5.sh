#!/bin/bash
set -o pipefail
tag=${1:-visit5}
out=gpurun_out/$tag
mkdir -p $out
echo "== crossover column-resident vs row-blocked dense kernel" 
for m in 49 33; do echo "-- SAME_DENSE_ROWBLOCK_MIN_T=$m"; SAME_DENSE_ROWBLOCK_MIN_T=$m timeout -k 10 300 python3 tools/dense_probe.py 50000 33,36,40,41,44,48 2>&1 | tee -a $out/dense_crossover.log; done
for m in 49 21; do echo "-- f32 SAME_DENSE_ROWBLOCK_MIN_T=$m"; SAME_DENSE_ROWBLOCK_MIN_T=$m timeout -k 10 300 python3 tools/dense_probe.py 50000 21,24,32,40,41,48 f32 2>&1 | tee -a $out/dense_crossover.log; done
echo "== window pipeline, whole plan" && timeout -k 10 900 python3 tools/window_bench.py 1000000 > $out/window_pipeline.log 2>&1 || { tail -20 $out/window_pipeline.log; exit 1; }
cat $out/window_pipeline.log
echo "== done"
