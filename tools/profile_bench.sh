#!/bin/bash
# Run ON the GPU box from the repo root:  tools/profile_bench.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the bench command  -> gpurun_out/prof_<tag>/stats
# 2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 hit-miss) of the same command -> gpurun_out/prof_<tag>/{fetch,write,l2}
# Summaries are written to gpurun_out/prof_<tag>/summary_*.txt; copy what should be judged into profiles/.
set -u
tag=$1
export TMPDIR=/tmp
root=$PWD
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
CMD=(python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- "${CMD[@]}" > "$out/stats.log" 2>&1)
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  set -- $pass; name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- "${CMD[@]}" > "$out/$name.log" 2>&1)
done
python3 "$root/tools/pmc_summary.py" "$out" > "$out/summary_pmc.txt" 2>&1
f=$(ls "$out"/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$out/summary_kernel_stats.csv"
grep -h '"metric"' "$out"/*.log > "$out/bench_lines.txt" 2>/dev/null
ls -R "$out" | head -40
