#!/bin/bash
# second visit: GPU tests (new fp32 / cfg5 / window pipeline), window + collapse pipelines, then the profile passes
set -o pipefail
tag=${1:-visit2}
out=gpurun_out/$tag
mkdir -p $out
echo "== pytest -m gpu" && timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=12 > $out/pytest_gpu.log 2>&1; rc=$?; tail -6 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
echo "== window pipeline 1M cells fp64" && timeout -k 10 600 python3 tools/window_bench.py 1000000 16 > $out/window_pipeline.log 2>&1 || { tail -20 $out/window_pipeline.log; exit 1; }
cat $out/window_pipeline.log
echo "== window pipeline 1M cells fp32 costs" && timeout -k 10 600 python3 tools/window_bench.py 1000000 16 f32 > $out/window_pipeline_f32.log 2>&1 || { tail -20 $out/window_pipeline_f32.log; exit 1; }
tail -4 $out/window_pipeline_f32.log
echo "== metacell collapse 100k" && timeout -k 10 600 python3 tools/metacell_time.py 100000 > $out/metacell_100k.log 2>&1 || { tail -20 $out/metacell_100k.log; exit 1; }
tail -5 $out/metacell_100k.log
bash tools/gpu_profile.sh $tag
