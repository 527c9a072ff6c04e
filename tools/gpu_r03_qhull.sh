#!/bin/bash
# Qhull helper placement: the pool's scaling on this host, then cfg 5 with and without placement.
set -o pipefail
out=gpurun_out/${1:-r04k}; mkdir -p $out
timeout -k 10 200 python3 tools/qhull_scaling_probe.py 13000 > $out/qhull_scaling.log 2>&1 || exit 1
for cfg in "0 8 3" "1 8 3" "1 12 3" "1 12 4" "1 12 6"; do
  set -- $cfg
  echo "== pin $1 helpers $2 threads $3" >> $out/cfg5.log
  SAME_QHULL_PIN=$1 SAME_QHULL_WORKERS=$2 timeout -k 10 200 python3 bench.py --workload cfg5 --cfg5-pipeline columns --cfg5-threads $3 --steps 3 --warmup 1 --no-cpu-baseline > $out/cfg5_pin$1_h$2_t$3.json 2>> $out/cfg5.log || exit 1
  python3 - $out/cfg5_pin$1_h$2_t$3.json >> $out/cfg5.log <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"   {d['windows_per_s']:.1f} windows/s, {d['ms_per_step']:.0f} ms/step, host glue {d['host_glue_share']:.3f}, merged {d['merged_matches']}")
P
done
cat $out/qhull_scaling.log; grep -A1 "^== pin" $out/cfg5.log
