#!/bin/bash
# Usage: tools/prof_icache.sh <outdir-under-gpurun_out> <program> [args...]   (run ON the GPU box, from the repo root)
# Instruction-cache and issue counters per kernel (separate --pmc passes with --kernel-trace only).
set -u
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p "$root/gpurun_out/$out"
pass() {
  name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$root/gpurun_out/$out/$name" -- "${PROG[@]}" > "$root/gpurun_out/$out/$name.log" 2>&1)
}
PROG=("$@")
pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass if SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass w1 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pass w2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/$out" > "$root/gpurun_out/$out/summary.txt" 2>&1
