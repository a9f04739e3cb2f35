#!/bin/bash
# Run ON the GPU box: what changes in the apply kernel between a smooth and the bench flow?  SQ / LDS / TCP counters of
# tools/ab_warp.py --only 1 at two roughness levels (separate --pmc passes, --kernel-trace only).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r3sigma
mkdir -p $O
for s in 2 8; do
  for pass in "a SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "d FETCH_SIZE" "e WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    set -- $pass; name=$1; shift
    (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/s${s}_$name -- python3 $R/tools/ab_warp.py --sigma $s --reps 1 --only 1 --batch 16 > $O/s${s}_$name.log 2>&1)
  done
  python3 tools/pmc_summary.py $O/ warp_bwd_lds_column > /dev/null 2>&1
done
for s in 2 8; do echo "== sigma $s"; for n in a b c d e; do python3 tools/pmc_summary.py $O/s${s}_$n "warp_bwd_lds_column_kernel<4, 3" ; done; done > $O/summary.txt 2>&1
cat $O/summary.txt
