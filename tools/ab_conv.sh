#!/bin/bash
# A/B of library builds on the conv microbenchmarks: [WHAT=fwd|wgrad] tools/ab_conv.sh "<shape list>" lib1.so lib2.so ...   (interleaved, 2 rounds)
shapes=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    DVD_LIB_PATH=$PWD/$lib python tools/conv_microbench.py ${WHAT:-fwd} 10 $shapes 2>&1 | grep -v amdgpu.ids
  done
done
