#!/bin/bash
# A/B of builds of the row-table kernel on one box: tools/ab_rows.sh <batch> lib1.so lib2.so ...   (tools/rows_check.py per library, twice)
b=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib (B = $b)"; OFL_HIP_LIB=$PWD/$lib python tools/rows_check.py --batch $b --sigmas 2 8 --no-hostile --reps 5 2>/dev/null | grep -v amdgpu
done; done
