#!/bin/bash
# Dev probe: build libsame_hip variants whose dense-cost output store carries different cache-policy bits
# (plain / nt / sc0 / sc1 / combinations) into tools/probes/build/, for tools/probes/store_variants.py to time.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"; CS="$ROOT/same_amd/csrc"
make -s -C "$CS"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$ROOT/include -Wall -Wno-unused-function"
OTHERS=$(ls $CS/build/*.o | grep -v cost.o)
build_one() {
  tag="$1"; mods="$2"
  /opt/rocm/bin/hipcc $FLAGS -DSAME_STORE_MODS="\"$mods\"" -c "$CS/cost.hip" -o "$HERE/build/cost_$tag.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/build/libsame_hip_$tag.so" "$HERE/build/cost_$tag.o" $OTHERS -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo "built $tag ($mods)"
}
build_one plain "" & build_one nt "nt" & build_one sc0 "sc0" & build_one sc1 "sc1" &
wait
build_one sc0sc1 "sc0 sc1" & build_one sc0nt "sc0 nt" & build_one sc1nt "sc1 nt" & build_one sc0sc1nt "sc0 sc1 nt" &
wait
