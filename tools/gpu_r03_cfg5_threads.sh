#!/bin/bash
# cfg 5: windows/s against worker threads, with the stage table of each run.
set -o pipefail
out=gpurun_out/${1:-r04m}; mkdir -p $out
for t in ${2:-1 2 4}; do
  timeout -k 10 200 python3 bench.py --workload cfg5 --cfg5-pipeline columns --cfg5-threads $t --steps 3 --warmup 1 --no-cpu-baseline > $out/cfg5_t$t.json 2>> $out/err.log || exit 1
  python3 - $out/cfg5_t$t.json $t <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"threads {sys.argv[2]}: {d['windows_per_s']:.1f} windows/s, {d['ms_per_step']:.0f} ms/step, host glue {d['host_glue_share']:.3f}, in library {d['per_rank']['in_library_s_per_step'][0]:.3f} s/step, merged {d['merged_matches']}")
for k, v in d['stages_rank0'].items():
    print(f"     {k:55s} {v['seconds'] / 3 * 1e3:8.1f} ms/step")
for e in d['library_calls_rank0_top']:
    print(f"     lib {e['entry_point']:51s} {e['seconds'] / 3 * 1e3:8.1f} ms/step")
P
done
