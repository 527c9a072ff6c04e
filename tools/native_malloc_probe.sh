#!/bin/bash
# Is one process on the opt-in triangulator route held by the kernel's per-process memory-map lock?  Every window allocates several
# arrays above glibc's 128 KiB mmap threshold (the triangles' block, coordinate copies, the batch's concatenated simplices): each is an
# mmap + page faults + munmap, all under one lock while 16-24 threads fault pages of their own.  The same sweep with glibc told to serve
# large blocks from the heap and keep it (MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_), interleaved with the default.
o=gpurun_out/r06_malloc; mkdir -p $o
for rep in 1 2; do
  timeout -k 10 300 python3 tools/native_delaunay_profile.py > $o/default_$rep.log 2>&1 || exit 1
  echo "default allocator settings, run $rep:"; grep "threads 16\|threads 24" $o/default_$rep.log
  MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967296 MALLOC_TOP_PAD_=268435456 timeout -k 10 300 python3 tools/native_delaunay_profile.py > $o/heap_$rep.log 2>&1 || exit 1
  echo "large blocks from the heap, run $rep:"; grep "threads 16\|threads 24" $o/heap_$rep.log
done
