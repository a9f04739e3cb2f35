#!/bin/bash
# A/B of several builds of libofl_hip.so on the warp kernels (run ON the GPU box):  tools/ab_warp_libs.sh <sigma> lib1.so lib2.so ...
sigma=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $(basename $lib .so)"; OFL_HIP_LIB=$PWD/$lib python3 tools/ab_warp.py --sigma $sigma --reps 1 --only 1 2>/dev/null | grep "shear on"
  done
done
