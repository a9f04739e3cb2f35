#!/bin/bash
# Usage: tools/prof_tcc.sh <outdir-under-gpurun_out> <program> [args...]   (run ON the GPU box, from the repo root)
# L2 <-> memory request counters of every kernel of the program: write / read requests by size, credit stalls, queue levels
# (LEVEL / REQ = mean latency in L2 clocks).  Separate --pmc passes with --kernel-trace only.
set -u
out=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p "$root/gpurun_out/$out"
pass() {
  name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$root/gpurun_out/$out/$name" -- "${PROG[@]}" > "$root/gpurun_out/$out/$name.log" 2>&1)
}
PROG=("$@"); case "${PROG[0]}" in ./*|tools/*) PROG[0]="$root/${PROG[0]#./}";; esac
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum
pass req TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_TAG_STALL_sum
pass lvl TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE
python3 "$root/tools/pmc_summary.py" "$root/gpurun_out/$out" > "$root/gpurun_out/$out/summary.txt" 2>&1
