cd /tmp && export TMPDIR=/tmp
for s in 8 16; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ovfprof$s -o t -- python3 $GRAFT_REPO_ROOT/tools/ab_libs.py --libs $GRAFT_REPO_ROOT/tools/microbench/var/ovf.so --ops apply_t combine3 --batch 64 --sigma $s --rounds 3 --iters 10 > /dev/null 2>&1
done
