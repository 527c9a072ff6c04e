for cfg in "15 2" "16 2" "18 2" "20 2" "24 2" "28 2" "20 3" "24 3"; do
  set -- $cfg
  SAME_QHULL_WORKERS=$1 timeout -k 10 200 python3 bench.py --workload cfg5 --cfg5-threads $2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('helpers $1 threads $2:', round(d['windows_per_s'],1), 'windows/s', round(d['ms_per_step']), 'ms/step, python_share', round(d['python_share'],2), 'qhull wait', round(d['qhull']['waiting_s_per_step_rank0'],3))"
done
