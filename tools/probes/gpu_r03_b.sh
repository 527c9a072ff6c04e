#!/bin/bash
# HISTORICAL: the occupancy sweep below drives the SAME_DENSE_LDS_PAD switch, which csrc/cost.hip had only up to commit 2b8b3b4;
# at HEAD every leg runs the same kernel.  Check that commit out to repeat profiles/r03_dense_occupancy.log.
# Round-3 visit B: occupancy sweep of the fp32 / fp64 T=20 dense kernels with power + clock telemetry.
set -o pipefail
tag=${1:-r03b}
out=gpurun_out/$tag
mkdir -p $out
for kind in f32 f64; do
  for pad in 0 33000 41000 65536; do
    SAME_DENSE_LDS_PAD=$pad timeout -k 10 120 python3 tools/probes/dense_occupancy.py $kind 20 2>&1 | tail -1 | tee -a $out/occupancy.log || exit 1
  done
done
for T in 8 12 16; do timeout -k 10 120 python3 tools/probes/dense_occupancy.py f32 $T 2>&1 | tail -1 | tee -a $out/occupancy.log || exit 1; done
