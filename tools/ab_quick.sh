#!/bin/bash
# quick A/B of library builds: bench_splat timings only (run ON the GPU box):  tools/ab_quick.sh <sigma> lib1.so lib2.so ...
sigma=$1; shift
for lib in "$@"; do
  echo "== $(basename $lib .so)"
  OFL_HIP_LIB="$PWD/$lib" python3 tools/bench_splat.py --sigma $sigma 2>/dev/null | tail -2 | cut -c1-190
done
