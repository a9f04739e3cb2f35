#!/bin/bash
# Run ON the GPU box: SQ instruction counters of the splat kernels (apply 's', B = 16) for the default library and, if given, a variant
# (OFL_HIP_LIB).  usage: tools/prof_splat_sq.sh <outdir-under-gpurun_out> [variant.so]
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
VAR=${2:-}
one() {  # name lib passname counters...
  name=$1; lib=$2; n=$3; shift 3
  (cd /tmp && OFL_HIP_LIB=$lib timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${name}_$n -- python3 $R/tools/bench_ops.py --only apply_s --batch 16 --iters 5 > $O/${name}_$n.log 2>&1)
}
run() {
  one $1 "$2" a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  one $1 "$2" b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAVES
}
run base ""
[ -n "$VAR" ] && run var $R/$VAR
for d in base var; do [ -d $O/${d}_a ] && { echo "== $d"; python3 tools/pmc_summary.py $O/${d}_a splat_; python3 tools/pmc_summary.py $O/${d}_b splat_; }; done > $O/summary.txt 2>&1
cat $O/summary.txt
