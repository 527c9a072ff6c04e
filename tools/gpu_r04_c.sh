#!/bin/bash
# round 4, visit C: the bench-facing GPU tests, 1 / 2 ranks of cfg 5 on the one GPU, the N = 2 line of another workload with its
# cfg5 record, and the kernel trace of the cfg 5 step (launches / fills / copies per window as rocprof counts them)
set -o pipefail
tag=${1:-r04c}
root=$PWD
out=$root/gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_run_same.py -m gpu -q -k "bench or cfg5" > $out/pytest_bench.log 2>&1 || { tail -40 $out/pytest_bench.log; exit 1; }
tail -2 $out/pytest_bench.log
run() {   # tag, args...
  t=$1; shift
  timeout -k 10 400 python3 bench.py --workload cfg5 --steps 3 --warmup 1 "$@" > $out/cfg5_$t.json 2> $out/cfg5_$t.err || { tail -20 $out/cfg5_$t.err; exit 1; }
  python3 - $out/cfg5_$t.json $t <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]}: {d['windows_per_s']:.1f} windows/s, {d['ms_per_step']:.0f} ms/step, host glue {d['host_glue_share']:.3f}, in library {d['per_rank']['in_library_s_per_step']} s/step, "
      f"qhull wait {d['per_rank']['qhull_wait_s_per_step']}, helpers {d['per_rank']['qhull_helpers']}, exchange {d.get('table_allgather')}, calls {d.get('runtime_calls_per_window')}, triangulations given: {d.get('windows_per_s_triangulations_given')}")
P
}
run 1m --no-cpu-baseline
run 1m_2rank --gpus 2 --no-cpu-baseline
timeout -k 10 500 python3 bench.py --gpus 2 --workload cfg2 --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_2rank_cfg2.json 2> $out/bench_2rank_cfg2.err || { tail -20 $out/bench_2rank_cfg2.err; exit 1; }
python3 -c "import json;d=json.loads(open('$out/bench_2rank_cfg2.json').read().strip().splitlines()[-1]);c=d['cfg5'];print('2 ranks, cfg2 line: cfg5 record', c.get('windows_per_s'), c.get('per_rank',{}).get('windows_per_s'), c.get('table_allgather'), c.get('qhull'))"
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace + stats of the cfg5 step"
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --cfg5-threads 1 > $out/trace_bench.json 2> $out/trace.err || { tail -20 $out/trace.err; exit 1; }
for f in $(find $out/trace -name "*_stats.csv"); do cp $f $out/cfg5_$(basename $f | sed 's/^[0-9]*_//'); done
ls $out
head -40 $out/cfg5_kernel_stats.csv | cut -c1-160
python3 - $out/trace_bench.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("windows per step", sum(d['per_rank']['windows']), "steps", d['steps'], "warmup", d['warmup'], "calls/window (library's own count)", d['runtime_calls_per_window'])
P
rm -rf $out/trace
echo "== done"
