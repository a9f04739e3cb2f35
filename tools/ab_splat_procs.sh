#!/bin/bash
# A/B of builds of libofl_hip.so in ALTERNATING PROCESSES on one box (two libraries with the round-6 splat kernels in one process
# hung once: profiles/r6_splat_diet.txt):   tools/ab_splat_procs.sh <rounds> lib1.so lib2.so ...   ("default" = the in-tree library)
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then timeout 120 python3 tools/splat_time.py 2>&1 | grep -v amdgpu.ids
    elif [ "$lib" = r5 ]; then timeout 120 python3 tools/splat_time.py --kernel 1 2>&1 | grep -v amdgpu.ids
    else OFL_HIP_LIB=$PWD/$lib timeout 120 python3 tools/splat_time.py 2>&1 | grep -v amdgpu.ids; fi
  done
done
