#!/bin/bash
# One GPU-box visit: GPU tests, the bench line, sysfs telemetry probe, forced-communicator and strong-scaling legs.
# Usage (on the box, from the repo root): bash tools/gpu_visit.sh <tag>
set -o pipefail
tag=${1:-visit}
out=gpurun_out/$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== sysfs probe" | tee $out/sysfs.log
python3 - >> $out/sysfs.log 2>&1 <<'PY'
import glob, os, sys
sys.path.insert(0, ".")
from same_amd import _lib
ctx = _lib.Context(0)
bus = ctx.pci_bus_id(); print("bus", bus, ctx.info())
d = f"/sys/bus/pci/devices/{bus}"
print(os.path.exists(d), sorted(os.listdir(d))[:80] if os.path.exists(d) else None)
for h in glob.glob(d + "/hwmon/hwmon*"):
    for f in sorted(os.listdir(h)):
        p = os.path.join(h, f)
        if os.path.isfile(p):
            try: print(f, "=", open(p).read().strip()[:80])
            except Exception as e: print(f, "ERR", e)
for f in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent", "current_compute_partition", "current_memory_partition"):
    try: print(f, "=", open(os.path.join(d, f)).read().strip().replace("\n", " | "))
    except Exception as e: print(f, "ERR", e)
PY
echo "== pytest -m gpu" && timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=15 > $out/pytest_gpu.log 2>&1; rc=$?; tail -5 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
echo "== smoke" && timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
echo "== bench (default)" && timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
tail -3 $out/bench.err
echo "== bench forced communicator (size-1 RCCL), which librccl" && SAME_BENCH_FORCE_COMM=1 NCCL_DEBUG=VERSION timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 5 > $out/bench_comm.json 2> $out/bench_comm.err || { tail -20 $out/bench_comm.err; exit 1; }
grep -i "librccl\|RCCL version\|rccl" $out/bench_comm.err | head -5
python3 - > $out/maps.log 2>&1 <<'PY'
import ctypes, os, sys
sys.path.insert(0, ".")
from same_amd import _lib
_lib.load()
print([l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l][:2])
PY
cat $out/maps.log
echo "== bench strong scaling cfg4 on one GPU (forced communicator: real collectives)" && SAME_BENCH_FORCE_COMM=1 timeout -k 10 600 python3 bench.py --scaling strong --steps 3 --warmup 1 > $out/bench_strong.json 2> $out/bench_strong.err || { tail -20 $out/bench_strong.err; exit 1; }
tail -2 $out/bench_strong.err
echo "== 2-rank launch rehearsal on the one GPU (RCCL refuses duplicate devices -> host transport)" && timeout -k 10 300 python3 bench.py --gpus 2 --workload cfg2 --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_2rank.json 2> $out/bench_2rank.err || { tail -20 $out/bench_2rank.err; exit 1; }
tail -2 $out/bench_2rank.err
echo "== 3 ranks on the one GPU, tiny (uneven blocks; host transport)" && timeout -k 10 300 python3 bench.py --gpus 3 --workload tiny --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_3rank.json 2> $out/bench_3rank.err || { tail -20 $out/bench_3rank.err; exit 1; }
tail -1 $out/bench_3rank.err
echo "== the driver's N>1 command form: bench.py as ranks of torch.distributed.run (2 ranks on the one GPU)" && timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload cfg2 --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_torchrun_2rank.json 2> $out/bench_torchrun_2rank.err || { tail -20 $out/bench_torchrun_2rank.err; exit 1; }
tail -1 $out/bench_torchrun_2rank.err
echo "== cfg5 (1M cells, windows), 1 rank and 2 ranks on the one GPU" && timeout -k 10 300 python3 bench.py --workload cfg5 --steps 3 --warmup 1 > $out/bench_cfg5.json 2> $out/bench_cfg5.err || { tail -20 $out/bench_cfg5.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload cfg5 --steps 3 --warmup 1 --gpus 2 > $out/bench_cfg5_2rank.json 2> $out/bench_cfg5_2rank.err || { tail -20 $out/bench_cfg5_2rank.err; exit 1; }
python3 - <<PY
import json
for f in ("bench_cfg5.json", "bench_cfg5_2rank.json"):
    d = json.load(open("$out/" + f)); print(f, "%.1f windows/s, %.0f ms/step, host glue %.2f" % (d["windows_per_s"], d["ms_per_step"], d["host_glue_share"]))
PY
echo "== done"
