set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
tools/profile_bench.sh final > gpurun_out/final/profile_bench.log 2>&1
python bench.py > gpurun_out/final/bench_final.json 2> gpurun_out/final/bench_final.err
{ python tools/bench_ops.py --batch 16; python tools/bench_ops.py --config5 --batch 16 --height 2160 --width 3840 --iters 5; python tools/bench_splat.py --sigma 2; python tools/bench_splat.py --sigma 8; python tools/bench_splat.py --sigma 8 --batch 64 --iters 5; } > gpurun_out/final/bench_ops.txt 2>/dev/null
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/ops_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py --batch 16 > $GRAFT_REPO_ROOT/gpurun_out/final/ops_stats.log 2>&1)
tools/prof_sq.sh final/sq python3 tools/bench_splat.py --sigma 8 --iters 3 > /dev/null 2>&1
tail -3 gpurun_out/final/bench_final.json | cut -c1-600
ls gpurun_out/final
