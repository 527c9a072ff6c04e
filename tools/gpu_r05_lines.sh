#!/bin/bash
# round 5, the other cfg 5 lines at the round's last tree: 2 ranks on the one GPU (own launcher and the driver's torchrun form), the frame
# pipeline, four times the section, 20 steps.  Outputs under gpurun_out/<tag>/ ; copied to profiles/r05_bench_cfg5_* / r05_rehearsal_* afterwards.
set -o pipefail
tag=${1:-r05_lines}
out=gpurun_out/$tag; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift; echo "== $name" && timeout -k 10 500 "$@" > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }; }
run cfg5_2ranks_one_gpu python3 bench.py --gpus 2 --workload cfg5 --steps 3 --warmup 1
run torchrun_2ranks_with_cfg5_record python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2
run cfg5_frames python3 bench.py --workload cfg5 --cfg5-pipeline frames --steps 2 --warmup 1
run cfg5_4m_cells python3 bench.py --workload cfg5 --cfg5-cells 4000000 --steps 2 --warmup 1
run cfg5_20_steps python3 bench.py --workload cfg5 --steps 20 --warmup 1 --no-extras
python3 tools/cfg5_lines.py $out/cfg5_2ranks_one_gpu.json $out/torchrun_2ranks_with_cfg5_record.json $out/cfg5_frames.json $out/cfg5_4m_cells.json $out/cfg5_20_steps.json
