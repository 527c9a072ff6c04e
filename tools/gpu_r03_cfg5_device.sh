#!/bin/bash
# cfg 5 with the sections resident on the device against the column pipeline, by worker threads.
set -o pipefail
out=gpurun_out/${1:-r04o}; mkdir -p $out
shift
cfgs=("$@")
[ ${#cfgs[@]} -eq 0 ] && cfgs=("columns 4" "device 1" "device 2" "device 4" "device 6")
timeout -k 10 500 python3 -m pytest tests/test_gpu_run_same.py -x -q -k "cfg5 or device_windows" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for cfg in "${cfgs[@]}"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --workload cfg5 --cfg5-pipeline $1 --cfg5-threads $2 --steps 3 --warmup 1 --no-cpu-baseline > $out/cfg5_$1_t$2.json 2>> $out/err.log || { tail -20 $out/err.log; exit 1; }
  python3 - $out/cfg5_$1_t$2.json $1 $2 <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} threads {sys.argv[3]}: {d['windows_per_s']:.1f} windows/s, {d['ms_per_step']:.0f} ms/step, host glue {d['host_glue_share']:.3f}, in library {d['per_rank']['in_library_s_per_step'][0]:.3f} s/step, merged {d['merged_matches']}")
for k, v in d['stages_rank0'].items():
    print(f"     {k:55s} {v['seconds'] / 3 * 1e3:8.1f} ms/step")
for e in d['library_calls_rank0_top'][:5]:
    print(f"     lib {e['entry_point']:51s} {e['seconds'] / 3 * 1e3:8.1f} ms/step")
P
done
