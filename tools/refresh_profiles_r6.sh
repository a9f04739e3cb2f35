#!/bin/bash
# Run ON the GPU box from the repo root: everything under profiles/r6_* (tools/collect_profiles_r6.py copies the summaries from
# gpurun_out/r6final/ afterwards).  --pmc passes are separate runs with --kernel-trace only, the program directly after `--`.
# Every text file starts with the box's device-copy rate (the pool's boxes differ by +-5 %).
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6final
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
sha256sum bench.py | cut -c1-16 > $O/bench_sha16.txt
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
# the many-channel warp: timings, and the PMC traffic of the SHIPPED row-table channel loop (C = 64, sigma 2 / 8)
{ echo "$BOX"; python tools/bench_chan.py --channels 64 16 7 4; python tools/bench_chan.py --channels 64 --sigma 2; python tools/bench_chan.py --channels 64 --sigma 12; python tools/bench_chan.py --channels 64 --batch 16; } > $O/bench_chan.txt 2>/dev/null
for sg in 2 8; do for pass in "fetch FETCH_SIZE" "write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/chan_s${sg}/$name -- python3 $R/tools/chan_once.py $sg 64 8 > $O/chan_s${sg}_$name.log 2>&1)
done; done
{ echo "$BOX"; for sg in 2 8; do echo "== B = 8, C = 64, sigma $sg: algorithmic bytes per launch = (8 + 8 * 64 + 1) x 8 x 1080 x 1920 = $((521 * 8 * 1080 * 1920))"; python3 tools/pmc_summary.py $O/chan_s$sg warp_bwd_lds_chan; done; } > $O/chan_pmc.txt 2>&1
# the splat: timings per launch, occupancy sweep, PMC traffic and SQ counters (apply 's', B = 16)
{ echo "$BOX"; tools/prof_splat_kernels.sh r6final/splat_k 2 8 12; python tools/splat_occupancy.py; python tools/ab_splat_kernels.py --rounds 5; } > $O/splat_kernels.txt 2>/dev/null
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_ANY"; do
  set -- $pass; name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/splat_$name -- python3 $R/tools/bench_ops.py --only apply_s --batch 16 --iters 5 > $O/splat_$name.log 2>&1)
done
{ echo "$BOX"; python3 tools/pmc_summary.py $O splat_; } > $O/splat_pmc_summary.txt 2>&1
# the second launch unit by unit (a -DOFL_SP2_UNITTIME=1 build: tools/build_variant.sh unittime -DOFL_SP2_UNITTIME=1), and the first launch's
# dynamic instruction counts per wave
if [ -f tools/microbench/var/unittime.so ]; then
  { echo "$BOX"; for sg in 8 12; do for op in apply_s switch_ref; do OFL_HIP_LIB=$R/tools/microbench/var/unittime.so timeout 100 python tools/redo_unit_times.py --sigma $sg --op $op 2>&1 | grep -v amdgpu.ids | cut -c1-700; done; done; } > $O/redo_units.txt
fi
{ echo "$BOX"; VARIANTS=default bash tools/prof_phase_insts.sh r6final/phase_insts 2 8 2>&1 | tail -4; } > $O/phase_insts.txt
# roughness sweep of the warp, the validation wait, the step's timeline at B = 8 and 64
{ echo "$BOX"; for s in 0.5 2 4 8 12 16; do python tools/ab_warp.py --sigma $s --reps 2 --only 1 2>/dev/null | grep "shear on" | tail -1 | sed "s/^/sigma $s  /"; done; } > $O/sigma_sweep.txt
{ echo "$BOX"; python tools/ab_flags.py; } > $O/flags.txt 2>/dev/null
{ echo "$BOX"; python tools/step_timeline.py; python tools/step_timeline.py --batch 64; } > $O/timeline.txt 2>/dev/null
{ echo "$BOX"; python tools/small_once.py 2>/dev/null | tail -12; } > $O/small.txt
{ echo "$BOX"; timeout 500 python tools/fuzz_gpu.py --seconds 240 --seed 606 2>&1 | tail -12; } > $O/fuzz.txt
ls $O
