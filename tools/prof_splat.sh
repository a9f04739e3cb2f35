#!/bin/bash
# Run ON the GPU box from the repo root:  tools/prof_splat.sh <tag> [sigma]   -> gpurun_out/splat_<tag>_kernel_stats.csv
set -u
tag=$1; sigma=${2:-8}
export TMPDIR=/tmp
root=$PWD
out="$root/gpurun_out/splat_$tag"
mkdir -p "$out"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/tools/bench_splat.py" --sigma "$sigma" > "$out/stats.log" 2>&1)
f=$(ls "$out"/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | head -12 | cut -c1-160
