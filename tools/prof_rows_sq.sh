#!/bin/bash
# Run ON the GPU box: SQ counters of the row-table warp kernel against the column kernel's rectangle (tools/rows_check.py runs both in one
# process).  usage: tools/prof_rows_sq.sh <outdir-under-gpurun_out> <sigma>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
S=${2:-0.5}
one() {  # passname counters...
  n=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/p_$n -- python3 $R/tools/rows_check.py --batch 16 --sigmas $S --no-hostile > $O/p_$n.log 2>&1)
}
one a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
one b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAVES
one c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_GDS
for n in a b c; do python3 tools/pmc_summary.py $O/p_$n warp_bwd; done > $O/summary.txt 2>&1
cat $O/summary.txt
