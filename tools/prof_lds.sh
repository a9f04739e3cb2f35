#!/bin/bash
# LDS conflict counters of a command: tools/prof_lds.sh <outdir-under-gpurun_out> <program> [args...]
set -u
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p "$root/gpurun_out/$out"
PROG=("$@"); [ "${PROG[0]}" = "python3" ] && PROG[1]="$root/${PROG[1]}"
(cd /tmp && timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$root/gpurun_out/$out/p" -- "${PROG[@]}" > "$root/gpurun_out/$out/p.log" 2>&1)
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/$out" warp_bwd_lds > "$root/gpurun_out/$out/summary.txt" 2>&1
