#!/bin/bash
# Run ON the GPU box from the repo root: everything under profiles/r5_* (tools/collect_profiles_r5.py copies the summaries from
# gpurun_out/r5final/ afterwards).  --pmc passes are separate runs with --kernel-trace only, the program directly after `--`.
# Every text file starts with the box's device-copy rate (the pool's boxes differ by +-5 %: VERDICT r4 item 7).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5final
mkdir -p $O
BOX=$(python3 - <<'PY'
import torch
a = torch.empty(1 << 28, dtype=torch.float32, device='cuda'); b = torch.empty_like(a); b.copy_(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): b.copy_(a)
e1.record(); torch.cuda.synchronize()
print("box: device copy %.2f TB/s (1 GiB fp32, read + write bytes / time)" % (5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12))
PY
)
echo "$BOX" > $O/box.txt
python bench.py > $O/bench.json 2> $O/bench.err
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline --no-probe --no-secondary"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- $BENCH > $O/bench_stats.log 2>&1)
f=$(ls $O/bench_stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  set -- $pass; name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/bench_$name -- $BENCH > $O/bench_$name.log 2>&1)
done
{ echo "$BOX"; python3 tools/pmc_summary.py $O warp_bwd; } > $O/bench_pmc_summary.txt 2>&1
# the other operations (B = 16), with their kernel statistics
{ echo "$BOX"; python tools/bench_ops.py --batch 16; python tools/bench_ops.py --config5 --batch 16 --height 2160 --width 3840 --iters 5; for s in 2 8 12; do python tools/bench_splat.py --sigma $s | tail -2; done; python tools/bench_splat.py --sigma 8 --batch 64 --iters 5 | tail -2; } > $O/bench_ops.txt 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ops_stats -- python3 $R/tools/bench_ops.py --batch 16 > $O/ops_stats.log 2>&1)
f=$(ls $O/ops_stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/ops_kernel_stats.csv
# backward passes
{ echo "$BOX"; python tools/bench_grad.py; } > $O/bench_grad.txt 2>/dev/null
# the many-channel warp
{ echo "$BOX"; python tools/bench_chan.py --channels 64 16 7 4; python tools/bench_chan.py --channels 64 --sigma 2; python tools/bench_chan.py --channels 64 --sigma 12; python tools/bench_chan.py --channels 64 --batch 16; } > $O/bench_chan.txt 2>/dev/null
# PMC traffic and SQ counters of the splat kernels (apply 's', B = 16)
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  set -- $pass; name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/splat_$name -- python3 $R/tools/bench_ops.py --only apply_s --batch 16 --iters 5 > $O/splat_$name.log 2>&1)
done
{ echo "$BOX"; python3 tools/pmc_summary.py $O splat_; } > $O/splat_pmc_summary.txt 2>&1
# roughness sweep of the warp, the validation wait, the step's timeline at B = 8
{ echo "$BOX"; for s in 0.5 2 4 8 12 16; do python tools/ab_warp.py --sigma $s --reps 2 --only 1 2>/dev/null | grep "shear on" | tail -1 | sed "s/^/sigma $s  /"; done; } > $O/sigma_sweep.txt
# the row-table kernel against the sheared rectangle (one process, alternating, medians), identity on hostile flows
{ echo "$BOX"; python tools/rows_check.py --batch 64 --reps 7 2>/dev/null; } > $O/rows_check.txt
{ echo "$BOX"; python tools/ab_flags.py; } > $O/flags.txt 2>/dev/null
{ echo "$BOX"; python tools/step_timeline.py; python tools/step_timeline.py --batch 64; } > $O/timeline.txt 2>/dev/null
ls $O
