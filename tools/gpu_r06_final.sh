#!/bin/bash
# round 6, the visit at the round's last tree: GPU suite, smoke, the default line, the cfg 5 line at one rank and as two ranks on the one
# GPU (host transport: RCCL refuses two ranks on one device), the product function's profile, then the rocprofv3 passes.
# Outputs under gpurun_out/<tag>/ ; the summaries that are judged are copied to profiles/r06_* afterwards.
set -o pipefail
tag=${1:-r06_final}
out=gpurun_out/$tag; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu" && timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=12 > $out/pytest_gpu.log 2>&1; rc=$?; tail -4 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
echo "== smoke" && timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
echo "== bench (default)" && timeout -k 10 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
echo "== bench (cfg5, 1 rank)" && timeout -k 10 400 python3 bench.py --workload cfg5 --steps 5 --warmup 1 > $out/bench_cfg5.json 2> $out/bench_cfg5.err || { tail -20 $out/bench_cfg5.err; exit 1; }
python3 tools/cfg5_lines.py $out/bench_cfg5.json
echo "== bench (cfg5, 2 ranks on the one GPU)" && timeout -k 10 400 python3 bench.py --workload cfg5 --steps 5 --warmup 1 --gpus 2 > $out/bench_cfg5_2ranks.json 2> $out/bench_cfg5_2ranks.err || { tail -20 $out/bench_cfg5_2ranks.err; exit 1; }
python3 tools/cfg5_lines.py $out/bench_cfg5_2ranks.json
echo "== bench (cfg5, 2 ranks, round-robin deal)" && timeout -k 10 400 python3 bench.py --workload cfg5 --steps 5 --warmup 1 --gpus 2 --cfg5-deal round_robin --no-extras > $out/bench_cfg5_2ranks_round_robin.json 2> $out/bench_cfg5_2ranks_rr.err || { tail -20 $out/bench_cfg5_2ranks_rr.err; exit 1; }
python3 tools/cfg5_lines.py $out/bench_cfg5_2ranks_round_robin.json
echo "== bench (cfg5, 1 rank, the timed step on the opt-in triangulator)" && timeout -k 10 400 python3 bench.py --workload cfg5 --steps 5 --warmup 1 --cfg5-delaunay native > $out/bench_cfg5_native.json 2> $out/bench_cfg5_native.err || { tail -20 $out/bench_cfg5_native.err; exit 1; }
python3 tools/cfg5_lines.py $out/bench_cfg5_native.json
echo "== bench (cfg5, 2 ranks on the one GPU, opt-in triangulator)" && timeout -k 10 400 python3 bench.py --workload cfg5 --steps 5 --warmup 1 --gpus 2 --cfg5-delaunay native --no-extras > $out/bench_cfg5_native_2ranks.json 2> $out/bench_cfg5_native_2ranks.err || { tail -20 $out/bench_cfg5_native_2ranks.err; exit 1; }
python3 tools/cfg5_lines.py $out/bench_cfg5_native_2ranks.json
echo "== product function profile" && timeout -k 10 300 python3 tools/incumbent_profile.py 1000000 1 > $out/incumbent_profile_merged.log 2>&1 && timeout -k 10 300 python3 tools/incumbent_profile.py 1000000 0 > $out/incumbent_profile_plain.log 2>&1; head -4 $out/incumbent_profile_merged.log
echo "== own triangulator (opt-in): profile, fuzz soak" && timeout -k 10 400 python3 tools/native_delaunay_profile.py > $out/native_delaunay.log 2>&1 || { tail -20 $out/native_delaunay.log; exit 1; }
grep "workers\|same_delaunay2d\|scipy" $out/native_delaunay.log | tail -22
SAME_FUZZ_ROUNDS=${NATIVE_SOAK_ROUNDS:-40} timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -s -k native_triangulator > $out/fuzz_native_soak.log 2>&1; rc=$?; tail -3 $out/fuzz_native_soak.log; [ $rc -eq 0 ] || exit $rc
bash tools/gpu_r06_profile.sh $tag/prof | tail -30
