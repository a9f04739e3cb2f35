#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/r6final/ (tools/refresh_profiles_r6.sh, run on the GPU box) into profiles/r6_* and derive
profiles/r6_traffic.json (HBM bytes per launch of the dominant kernel: 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md gfx950
correction) from the two PMC passes, stamped with the sha16 of the bench.py that ran and the box's copy rate."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r6final")
DST = os.path.join(ROOT, "profiles")
names = {"bench.json": "r6_bench.json", "bench_kernel_stats.csv": "r6_bench_kernel_stats.csv", "bench_pmc_summary.txt": "r6_bench_pmc_summary.txt",
         "bench_ops.txt": "r6_bench_ops_B16.txt", "ops_kernel_stats.csv": "r6_bench_ops_kernel_stats.csv", "bench_grad.txt": "r6_bench_grad_B16.txt",
         "splat_pmc_summary.txt": "r6_splat_gather_pmc_summary.txt", "splat_kernels.txt": "r6_splat_kernels.txt",
         "sigma_sweep.txt": "r6_sigma_sweep.txt", "bench_chan.txt": "r6_bench_chan.txt", "chan_pmc.txt": "r6_chan_pmc.txt",
         "redo_units.txt": "r6_splat_second_launch_units.txt", "phase_insts.txt": "r6_splat_phase_insts.txt", "flags.txt": "r6_validation_wait.txt", "timeline.txt": "r6_step_timeline.txt", "small.txt": "r6_small_launches.txt", "fuzz.txt": "r6_fuzz.txt"}
for a, b in names.items():
    p = os.path.join(SRC, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(DST, b))
        print("profiles/" + b)
box = open(os.path.join(SRC, "box.txt")).read().strip() if os.path.exists(os.path.join(SRC, "box.txt")) else None
sha = open(os.path.join(SRC, "bench_sha16.txt")).read().strip() if os.path.exists(os.path.join(SRC, "bench_sha16.txt")) else None
summ = os.path.join(SRC, "bench_pmc_summary.txt")
if os.path.exists(summ):
    text = open(summ).read()
    best = None
    for block in re.split(r"\n(?=\S)", text):
        if "warp_bwd_rows_kernel<4, 3, true" in block:
            f = re.search(r"FETCH_SIZE\s+mean=([0-9.e+]+)", block)
            w = re.search(r"WRITE_SIZE\s+mean=([0-9.e+]+)", block)
            if f and w:
                best = (float(f.group(1)), float(w.group(1)), block.splitlines()[0].strip())
    if best:
        fetch_kib, write_kib, kname = best
        out = {"kernel": kname, "workload": "B=64 1080x1920 Flow.apply 't' C=3 + valid", "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
               "fetch_correction": 2.0, "traffic_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024,
               "algorithmic_bytes_per_launch": 35 * 64 * 1080 * 1920, "bench_py_sha16": sha, "box": box,
               "source": "profiles/r6_bench_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, same bench command, "
                         "tools/refresh_profiles_r6.sh; gfx950: FETCH_SIZE tallies 64 B per 128-B request for 16 B/lane streams -> doubled)"}
        json.dump(out, open(os.path.join(DST, "r6_traffic.json"), "w"), indent=1)
        print("profiles/r6_traffic.json", out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"])
