#!/bin/bash
# Round-3 visit A: VALU issue-rate probe, SQ counter passes over the fp32 / fp64 dense kernels (tools/dense_probe.py), GPU tests.
set -o pipefail
tag=${1:-r03a}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== valu issue probe"
[ -x tools/probes/valu_issue ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_issue tools/probes/valu_issue.hip || { echo "cannot build tools/probes/valu_issue"; exit 1; }
timeout -k 10 120 tools/probes/valu_issue > $out/valu_issue.log 2>&1 || { cat $out/valu_issue.log; exit 1; }
cat $out/valu_issue.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L > $out/counters_avail.txt 2>&1 || true
pass() {  # name, dtype, counters...
  local name=$1 dt=$2; shift 2
  echo "== pmc $name ($dt): $*"
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_$name -- python3 $root/tools/dense_probe.py 100000 20 $dt > $out/pmc_${name}.log 2> $out/pmc_$name.err || { tail -5 $out/pmc_$name.err; return 1; }
  python3 $root/tools/pmc_summary.py $out/pmc_$name $out/pmc_${name}_summary.csv | grep -i "dense\|Kernel_Name" | cut -c1-200
  rm -rf $out/pmc_$name
}
for dt in f32 f64; do
  pass ${dt}_insts $dt SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE || exit 1
  pass ${dt}_waves $dt SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES || exit 1
  pass ${dt}_active $dt SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA || true
  pass ${dt}_fetch $dt SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM || true
done
cd $root
echo "== pytest -m gpu" && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=10 > $out/pytest_gpu.log 2>&1; rc=$?; tail -5 $out/pytest_gpu.log
echo "== done rc=$rc"; exit $rc
