#!/bin/bash
set -o pipefail
tag=${1:-visit4}
out=gpurun_out/$tag
mkdir -p $out
echo "== pytest -m gpu" && timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=8 > $out/pytest_gpu.log 2>&1; rc=$?; tail -4 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
echo "== generic dense kernel (T > 48)" && timeout -k 10 300 python3 tools/dense_probe.py 50000 48,49,64,128 > $out/dense_generic.log 2>&1 || { tail $out/dense_generic.log; exit 1; }
cat $out/dense_generic.log
timeout -k 10 300 python3 tools/dense_probe.py 50000 49,64,128 f32 >> $out/dense_generic.log 2>&1; tail -3 $out/dense_generic.log
echo "== window pipeline" && timeout -k 10 600 python3 tools/window_bench.py 1000000 16 > $out/window_pipeline.log 2>&1 || { tail -20 $out/window_pipeline.log; exit 1; }
cat $out/window_pipeline.log
echo "== window main-process profile" && timeout -k 10 600 python3 tools/window_profile.py 1000000 16 > $out/window_profile.log 2>&1 || { tail $out/window_profile.log; exit 1; }
head -40 $out/window_profile.log | cut -c1-150
echo "== done"
