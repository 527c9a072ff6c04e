#!/bin/bash
# round 6: the rocprofv3 passes of tools/gpu_profile.sh over the default bench command, then kernel stats of the window configuration
# (bench.py --workload cfg5: the window kernels and the new merge kernels -- collect / resolve / finish).  Outputs under gpurun_out/<tag>/.
set -o pipefail
tag=${1:-r06_prof}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_profile.sh $tag | tail -16 || exit 1
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --workload cfg5 --no-cpu-baseline --no-extras --steps 3 --warmup 1"
echo "== cfg5: kernel trace + stats" && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg5_stats -- $cmd > $out/cfg5_under_rocprof.json 2> $out/cfg5_stats.err || { tail -20 $out/cfg5_stats.err; exit 1; }
f=$(find $out/cfg5_stats -name "*kernel_stats.csv" | head -1); cp "$f" $out/cfg5_kernel_stats.csv; head -12 $out/cfg5_kernel_stats.csv | cut -c1-200
rm -rf $out/cfg5_stats
echo "== done"
