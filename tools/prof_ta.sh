#!/bin/bash
# Usage: tools/prof_ta.sh <outdir> <program> [args...]   TA / TD / TCP busy counters, one small pass each
set -u
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p "$root/gpurun_out/$out"
PROG=("$@"); case "${PROG[0]}" in ./*|tools/*) PROG[0]="$root/${PROG[0]#./}";; esac
pass() {
  name=$1; shift
  (cd /tmp && timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$root/gpurun_out/$out/$name" -- "${PROG[@]}" > "$root/gpurun_out/$out/$name.log" 2>&1)
}
pass ta1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass ta2 TA_BUSY_avr TA_BUSY_max
pass ta3 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass td1 TD_TD_BUSY_sum TD_TC_STALL_sum
pass tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
pass tcp3 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum
pass tcc1 TCC_BUSY_sum TCC_TAG_STALL_sum
pass sqx SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/$out" > "$root/gpurun_out/$out/summary.txt" 2>&1
