#!/bin/bash
# Dev probe: block->tile map and rows-per-block against buffer placement (each process: two buffers; ld=131072 classifies the placement)
cd "$(dirname "$0")/../.." || exit 1
python3 tools/probes/store_pitch.py 100000,106496,114688,131072 || exit 1
for m in 0 1 2 3 4; do
  SAME_DENSE_MAP=$m python3 tools/probes/store_pitch.py 100000,131072 || exit 1
done
for r in 32 64 128 512 1024; do
  SAME_DENSE_RPB=$r python3 tools/probes/store_pitch.py 100000,131072 || exit 1
done
