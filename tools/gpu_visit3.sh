#!/bin/bash
set -o pipefail
tag=${1:-visit3}
out=gpurun_out/$tag
mkdir -p $out
echo "== bench default" && timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
tail -2 $out/bench.err
echo "== bench shared stream (round-1 ordering) for comparison" && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --tail-stream shared --no-cpu-baseline --no-extras > $out/bench_shared.json 2> $out/bench_shared.err || { tail -20 $out/bench_shared.err; exit 1; }
python3 -c "
import json
for f in ('bench','bench_shared'):
    d=json.load(open('$out/'+f+'.json')); print(f, round(d['ms_per_step'],3), 'ms/step, dense', round(d['roofline']['kernel_ms'],3), 'ms, value %.4g' % d['value'])
"
echo "== generic dense kernel (T > 48)" && timeout -k 10 300 python3 tools/dense_probe.py 50000 48,49,64,128 > $out/dense_generic.log 2>&1 || { tail $out/dense_generic.log; exit 1; }
cat $out/dense_generic.log
echo "== metacell" && timeout -k 10 300 python3 tools/metacell_time.py 100000 > $out/metacell_100k.log 2>&1 || { tail $out/metacell_100k.log; exit 1; }
cat $out/metacell_100k.log
echo "== window main-process profile" && timeout -k 10 600 python3 tools/window_profile.py 1000000 16 > $out/window_profile.log 2>&1 || { tail $out/window_profile.log; exit 1; }
head -70 $out/window_profile.log | cut -c1-160
echo "== done"
