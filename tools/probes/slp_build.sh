#!/bin/bash
# Round 5: libsame_hip with hipcc's SLP vectorizer left ON for cost.hip (the product builds with -fno-slp-vectorize), as a second
# library beside the product's: tools/probes/libsame_hip_slp.so.  With it hipcc packs the fp32 dense loop by itself --
# dense_cost_kernel<float,20,4>: 86 v_pk_add_f32 (the subtraction with an op_sel broadcast of the SGPR row value and neg_lo / neg_hi, and the
# accumulate) + 90 v_and_b32 (the packed encoding has no abs modifier) per row of four outputs, against 80 v_sub_f32 + 84 v_add_f32.
# Usage: bash tools/probes/slp_build.sh   (from the repo root; needs the product's objects: make -C same_amd/csrc first)
#        SAME_HIP_LIB=tools/probes/libsame_hip_slp.so SAME_SPREAD=0 python3 tools/dense_probe.py 100000 8,20 f32
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
b=$root/same_amd/csrc/build
mkdir -p $root/tools/probes/build_slp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$root/include -Wall -Wno-unused-function \
    -c $root/same_amd/csrc/cost.hip -o $root/tools/probes/build_slp/cost.o
objs=$(ls $b/*.o | grep -v /cost.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/probes/libsame_hip_slp.so $root/tools/probes/build_slp/cost.o $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $root/tools/probes/libsame_hip_slp.so
