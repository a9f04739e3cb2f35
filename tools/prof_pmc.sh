#!/bin/bash
# Usage: tools/prof_pmc.sh <outdir-under-gpurun_out> <program> [args...]   (run ON the GPU box, from the repo root)
# Separate --pmc passes (TCC has 4 slots; FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md), each with
# --kernel-trace only, as gpurun requires.
set -u
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p "$root/gpurun_out/$out"
pass() {
  name=$1; shift
  (cd /tmp && timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$root/gpurun_out/$out/$name" -- "${PROG[@]}" > "$root/gpurun_out/$out/$name.log" 2>&1)
}
PROG=("$@"); case "${PROG[0]}" in ./*|tools/*) PROG[0]="$root/${PROG[0]#./}";; esac
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass tcc TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum

pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/$out" > "$root/gpurun_out/$out/summary.txt" 2>&1
