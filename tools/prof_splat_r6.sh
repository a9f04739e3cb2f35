#!/bin/bash
# Run ON the GPU box: SQ counters (instruction mix, LDS array cycles and conflicts, waits) of the gather splat, round-6 kernel
# (OFL_SPLAT_KERNEL=0) and round 5's (1).   usage: tools/prof_splat_r6.sh <outdir-under-gpurun_out>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
one() {  # name kernel passname counters...
  name=$1; k=$2; n=$3; shift 3
  (cd /tmp && OFL_SPLAT_KERNEL=$k timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${name}_$n -- python3 $R/tools/splat_once.py > $O/${name}_$n.log 2>&1)
}
run() {
  one $1 $2 a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  one $1 $2 b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAVES
  one $1 $2 c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_LDS_ATOMIC_RETURN
}
run diet 0
run r5 1
for d in diet r5; do echo "== $d"; for n in a b c; do python3 tools/pmc_summary.py $O/${d}_$n splat_gather; done; done > $O/summary.txt 2>&1
cat $O/summary.txt
