#!/bin/bash
# Usage: tools/prof_sq.sh <outdir-under-gpurun_out> <program> [args...]  -- SQ activity breakdown (8 SQ slots per pass)
set -u
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p "$root/gpurun_out/$out"
PROG=("$@"); [ "${PROG[0]}" = "python3" ] && PROG[1]="$root/${PROG[1]}"; case "${PROG[0]}" in ./*|tools/*) PROG[0]="$root/${PROG[0]#./}";; esac
pass() {
  name=$1; shift
  (cd /tmp && timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$root/gpurun_out/$out/$name" -- "${PROG[@]}" > "$root/gpurun_out/$out/$name.log" 2>&1)
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES
pass sq4 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/$out" > "$root/gpurun_out/$out/summary.txt" 2>&1
