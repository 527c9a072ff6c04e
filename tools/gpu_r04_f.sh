#!/bin/bash
# round 4, visit F: fuzz soaks at the round's last tree -- 300 rounds as shipped, then 100 rounds with every
# look-back scan forced to recompute its predecessors (SAME_SCAN_FORCE_RECOMPUTE=1).  Logs: profiles/r04_fuzz_soak_*.log
set -o pipefail
out=gpurun_out/${1:-r04f}; mkdir -p $out
SAME_FUZZ_ROUNDS=300 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -s --durations=8 > $out/fuzz_soak_300_rounds.log 2>&1; rc=$?
tail -14 $out/fuzz_soak_300_rounds.log; [ $rc -ne 0 ] && exit $rc
SAME_SCAN_FORCE_RECOMPUTE=1 SAME_FUZZ_ROUNDS=100 timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -s --durations=8 > $out/fuzz_soak_100_rounds_forced_recompute.log 2>&1; rc=$?
tail -14 $out/fuzz_soak_100_rounds_forced_recompute.log; exit $rc
