#!/bin/bash
# A/B of several builds of libofl_hip.so on the SAME box (run ON the GPU box from the repo root):
#   tools/ab_splat.sh <outdir-under-gpurun_out> lib1.so lib2.so ...
# For every library: bench_splat timings (sigma 8) and the rocprofv3 kernel statistics of the same command.
set -u
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p "$root/gpurun_out/$out"
for lib in "$@"; do
  name=$(basename "$lib" .so)
  echo "== $name"
  OFL_HIP_LIB="$root/$lib" python3 "$root/tools/bench_splat.py" --sigma 8 2>/dev/null | tail -2
  (cd /tmp && OFL_HIP_LIB="$root/$lib" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/$out/$name" -- python3 "$root/tools/bench_splat.py" --sigma 8 > /dev/null 2>&1)
  f=$(ls "$root/gpurun_out/$out/$name"/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "splat_|Name" "$f" | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-120
done
