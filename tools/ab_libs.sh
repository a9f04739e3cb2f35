#!/bin/bash
# A/B of two builds of libofl_hip.so on the same GPU box: alternates processes (tools/ab_warp.py) with OFL_HIP_LIB.
#   tools/ab_libs.sh tools/ab/lib_base.so oflibpytorch_amd/libofl_hip.so [sigma]
a=$1; b=$2; sigma=${3:-8}
for rep in 1 2; do
  for lib in "$a" "$b"; do
    echo "== $lib"; OFL_HIP_LIB=$PWD/$lib python tools/ab_warp.py --sigma $sigma --reps 1 2>/dev/null | grep "shear on"
  done
done
