#!/bin/bash
set -o pipefail
tag=${1:-visit7}
out=gpurun_out/$tag
mkdir -p $out
echo "== forced comm weak" && SAME_BENCH_FORCE_COMM=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 5 > $out/bench_comm.json 2> $out/bench_comm.err || { tail -20 $out/bench_comm.err; exit 1; }
echo "== forced comm strong" && SAME_BENCH_FORCE_COMM=1 timeout -k 10 600 python3 bench.py --scaling strong --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_strong.json 2> $out/bench_strong.err || { tail -20 $out/bench_strong.err; exit 1; }
echo "== 2 ranks weak (host transport)" && timeout -k 10 300 python3 bench.py --gpus 2 --workload cfg2 --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_2rank.json 2> $out/bench_2rank.err || { tail -20 $out/bench_2rank.err; exit 1; }
echo "== 3 ranks strong, cfg2 (host transport)" && timeout -k 10 300 python3 bench.py --gpus 3 --workload cfg2 --scaling strong --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_3rank_strong.json 2> $out/bench_3rank_strong.err || { tail -20 $out/bench_3rank_strong.err; exit 1; }
python3 -c "
import json
for f in ('bench_comm','bench_strong','bench_2rank','bench_3rank_strong'):
    d=json.load(open('$out/'+f+'.json')); print(f, '%.4g' % d['value'], round(d['ms_per_step'],3), '|', d['parity_spot_check'][:120])
"
echo "== pre-MIP stages" && for n in 10000 50000 200000; do timeout -k 10 300 python3 tools/premip_time.py $n 2>&1 | tee -a $out/premip_stages.log; done
echo "== done"
