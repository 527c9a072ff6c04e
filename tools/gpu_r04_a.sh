#!/bin/bash
# round 4, visit A: the rewritten window path -- smoke, the window / greedy GPU tests, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04a_smoke.log 2>&1 || { tail -30 gpurun_out/r04a_smoke.log; exit 1; }
tail -2 gpurun_out/r04a_smoke.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_run_same.py tests/test_gpu_fuzz.py::test_fuzz_device_windows tests/test_gpu_fuzz.py::test_device_window_argument_checks \
    tests/test_gpu_parity.py::test_greedy_chain_is_resolved_in_batches tests/test_gpu_parity.py::test_greedy_match_equals_sequential_scan \
    -m gpu -q -s > gpurun_out/r04a_window_tests.log 2>&1 || { tail -60 gpurun_out/r04a_window_tests.log; exit 1; }
tail -8 gpurun_out/r04a_window_tests.log
grep "per window" gpurun_out/r04a_window_tests.log
