#!/bin/bash
# cfg 5 (device pipeline): windows/s against the number of Qhull helpers and worker threads.
set -o pipefail
out=gpurun_out/${1:-r04s}; mkdir -p $out
for cfg in "12 2" "14 2" "16 2" "14 4" "16 4" "20 4"; do
  set -- $cfg
  SAME_QHULL_WORKERS=$1 timeout -k 10 200 python3 bench.py --workload cfg5 --cfg5-threads $2 --steps 3 --warmup 1 --no-cpu-baseline > $out/cfg5_h$1_t$2.json 2>> $out/err.log || { tail -20 $out/err.log; exit 1; }
  python3 - $out/cfg5_h$1_t$2.json $1 $2 <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d['stages_rank0']
print(f"helpers {sys.argv[2]} threads {sys.argv[3]}: {d['windows_per_s']:.1f} windows/s, {d['ms_per_step']:.0f} ms/step, host glue {d['host_glue_share']:.3f}, in library {d['per_rank']['in_library_s_per_step'][0]:.3f} s/step, merge {st['merge (device de-duplication + host matching)']['seconds'] / 3 * 1e3:.0f} ms/step, merged {d['merged_matches']}")
P
done
