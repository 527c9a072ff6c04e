#!/bin/bash
# round 4, visit B: whole GPU suite, cfg 5 at 1M cells (1 and 2 ranks on the one GPU) and at 4M cells
set -o pipefail
out=gpurun_out/${1:-r04b}; mkdir -p $out
echo "== pytest -m gpu" && timeout -k 10 1100 python3 -m pytest tests -m gpu -q --durations=12 > $out/pytest_gpu.log 2>&1; rc=$?; tail -5 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
run() {   # tag, args...
  tag=$1; shift
  timeout -k 10 400 python3 bench.py --workload cfg5 --steps 3 --warmup 1 "$@" > $out/cfg5_$tag.json 2> $out/cfg5_$tag.err || { tail -20 $out/cfg5_$tag.err; exit 1; }
  python3 - $out/cfg5_$tag.json $tag <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]}: {d['windows_per_s']:.1f} windows/s, {d['ms_per_step']:.0f} ms/step, host glue {d['host_glue_share']:.3f}, python {d['python_share']:.3f}, "
      f"in library {d['per_rank']['in_library_s_per_step']} s/step, qhull wait {d['per_rank']['qhull_wait_s_per_step']}, helpers {d['qhull']['helpers']}, merged {d['merged_matches']}, calls {d.get('runtime_calls_per_window')}")
for k, v in d['stages_rank0'].items():
    print(f"     {k:60s} {v['seconds'] / d['steps'] * 1e3:8.1f} ms/step")
for e in d['library_calls_rank0_top'][:6]:
    print(f"     lib {e['entry_point']:56s} {e['seconds'] / d['steps'] * 1e3:8.1f} ms/step")
P
}
run 1m
run 1m_2rank --gpus 2 --no-cpu-baseline
timeout -k 10 400 python3 bench.py --gpus 2 --workload cfg2 --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_2rank_cfg2.json 2> $out/bench_2rank_cfg2.err || { tail -20 $out/bench_2rank_cfg2.err; exit 1; }
python3 -c "import json;d=json.loads(open(\"$out/bench_2rank_cfg2.json\").read().strip().splitlines()[-1]);c=d[\"cfg5\"];print(\"2 ranks cfg2 line: cfg5 record\", c.get(\"windows_per_s\"), c.get(\"per_rank\",{}).get(\"windows_per_s\"), c.get(\"table_allgather\"), c.get(\"qhull\"))"
run 4m --cfg5-cells 4000000 --no-cpu-baseline
run 1m_t1 --cfg5-threads 1 --no-cpu-baseline
run 1m_t3 --cfg5-threads 3 --no-cpu-baseline
echo "== done"
