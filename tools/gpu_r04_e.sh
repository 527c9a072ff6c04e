#!/bin/bash
# round 4, visit E: the many-block scan test, the metacell tests (batched disjoint greedy), the C demo
set -o pipefail
out=gpurun_out/${1:-r04e}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py::test_window_stage_with_thousands_of_scan_blocks tests/test_boundary.py tests/test_gpu_parity.py -m gpu -q -k "thousands or abi_demo or collapse or disjoint or metacell or greedy" --durations=5 > $out/pytest.log 2>&1; rc=$?
tail -12 $out/pytest.log; exit $rc
