#!/usr/bin/env python3
"""Benchmark of the warp / compose hot path on MI355X (contract: see the task's bench.py section).

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:

    warped, valid = flow2.apply(image, target_mask, return_valid_area=True)    # 't' backward warp, C=3, 35 B/px
    flow3         = flow1.combine_with(flow2, mode=3)                          # fused composition,       27 B/px

on B x 1080 x 1920 fp32 per GPU (BASELINE.json configs[1]/[3]: Flow.apply + combine_flows mode 3).  Every step builds
its `Flow` objects from the raw tensors again -- construction is the reference's validation (`isfinite().all()`,
utils.py:98; here one fused flag reduction + one host sync per flow) -- so `value` is the streaming, validation-inclusive
rate; the same step on pre-built (already validated, flag-cached) objects is reported as `value_cached_flow_objects`.
The metric is Mpix/s = global batch * H * W / t_step ("warped+composed").

    --scaling weak   (default) B = --batch (64) PER GPU: every rank owns its own shard, no data-path collective
    --scaling strong           --batch (64) is the GLOBAL batch (BASELINE.json configs[3]): 64 / 32 / 16 / 8 per GPU at 1 / 2 / 4 / 8
At N = 1 the line also carries `strong_scaling_probe`: the same step at the 8 elements per GPU that config 4 leaves each
of 8 GPUs, with the per-step host overhead (wall time minus HIP-event kernel time) -- what bounds strong scaling.

    python bench.py                       # 1 GPU, defaults finish in a couple of minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HBM_PEAK_GBS = 8000.0                    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_APPLY, BYTES_COMBINE = 35, 27      # algorithmic B/px


def smooth_flow(n, h, w, sigma, seed, device):
    """N(0,1)*sigma at (H/40, W/40), bicubic-upsampled (align_corners) to H x W."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    lo = (torch.randn(n, 2, max(h // 40, 2), max(w // 40, 2), generator=g) * sigma).to(device)
    return F.interpolate(lo, size=(h, w), mode='bicubic', align_corners=True).contiguous()


def hole_mask(n, h, w, device):
    """~10 % invalid: two rectangles and a 50-row band."""
    m = torch.ones(n, h, w, dtype=torch.bool, device=device)
    m[:, h // 5:h // 5 + h // 8, w // 6:w // 6 + w // 5] = False
    m[:, h // 2:h // 2 + h // 10, (2 * w) // 3:(2 * w) // 3 + w // 6] = False
    m[:, (3 * h) // 4:(3 * h) // 4 + 50, :] = False
    return m


def make_inputs(n, h, w, device, seed):
    f1 = smooth_flow(n, h, w, 8.0, 1000 + seed, device)
    f2 = smooth_flow(n, h, w, 8.0, 5000 + seed, device)
    g = torch.Generator(device='cpu').manual_seed(2000 + seed)
    img = (torch.rand(min(n, 4), 3, h, w, generator=g) * 255).to(device)
    img = img.repeat((n + img.shape[0] - 1) // img.shape[0], 1, 1, 1)[:n].contiguous()   # fp32 [n,3,h,w]
    m1, m2 = hole_mask(n, h, w, device), hole_mask(n, h, w, device).flip(2)
    tm = hole_mask(n, h, w, device).flip(1)
    return f1, f2, img, m1, m2, tm


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(h, w, seconds, threads):
    """CPU baselines on the GPU box's host cores, on a bounded sample (B = 2 of the same step):
      * `value` (kind "port"): the C oracle (oracle/: scalar restatement of the reference's arithmetic, OpenMP over
        batch x rows) -- the fastest faithful CPU implementation we have;
      * `torch_ops`: the reference's own torch-CPU OP SEQUENCE (oracle/torch_ops.py: F.grid_sample + mask plumbing + its
        validation passes; pinned against the reference's fixtures) -- what the reference itself would spend here;
      * `kernel_only`: the bare ATen kernel (F.grid_sample on a ready-made grid, C = 4 and C = 3 planes), so that the
        ratio is not inflated by Python / validation overhead."""
    from oracle import oracle, torch_ops
    oracle.set_threads(threads)
    n = 2
    tens = make_inputs(n, h, w, torch.device('cpu'), 77)
    f1, f2, img, m1, m2, tm = [t.numpy() for t in tens]
    oracle.flow_apply(f2, 't', m2, img, tm)        # warm-up (page-in, thread pool)
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.flow_apply(f2, 't', m2, img, tm)
        oracle.combine_with(f1, m1, f2, m2, 3, 't')
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 200:
            break
    out = {"value": round(reps * n * h * w / el / 1e6, 3), "unit": "Mpix/s", "cores": threads, "kind": "port",
           "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(),
           "sample": "oracle (C, OpenMP) Flow.apply('t',C=3,valid)+combine mode 3 on B=%d %dx%d fp32, %d reps in %.1f s"
                     % (n, h, w, reps, el)}
    # the reference's torch-CPU op sequence, torch's own intra-op threads
    tf1, tf2, timg, tm1, tm2, ttm = tens

    def torch_step():
        torch_ops.flow_apply_t(tf2, tm2, timg, ttm)
        torch_ops.combine_mode3_t(tf1, tm1, tf2, tm2)
    torch_step()
    times = []
    t0 = time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - t0 < seconds / 2 and len(times) < 50):
        t1 = time.perf_counter()
        torch_step()
        times.append(time.perf_counter() - t1)
    med = sorted(times)[len(times) // 2]
    out["torch_ops"] = {"value": round(n * h * w / med / 1e6, 3), "unit": "Mpix/s", "torch_threads": torch.get_num_threads(),
                        "sample": "reference op sequence (F.grid_sample + mask plumbing + validation) on B=%d, median of %d"
                                  % (n, len(times))}
    # kernel only: grid_sample of (u, v, mask) + of (3 channels + mask) on ready-made grids
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    grid = torch.stack((gx, gy), -1).float().unsqueeze(0) - tf2.permute(0, 2, 3, 1)
    grid = torch_ops.normalise_coords(grid, (h, w))
    s4 = torch.cat((timg, ttm.unsqueeze(1).float()), 1)
    s3 = torch.cat((tf1, tm1.unsqueeze(1).float()), 1)
    torch_ops.kernel_only(grid, s4)
    times = []
    for _ in range(5):
        t1 = time.perf_counter()
        torch_ops.kernel_only(grid, s4)
        torch_ops.kernel_only(grid, s3)
        times.append(time.perf_counter() - t1)
    med = sorted(times)[len(times) // 2]
    out["kernel_only"] = {"value": round(n * h * w / med / 1e6, 3), "unit": "Mpix/s", "torch_threads": torch.get_num_threads(),
                          "sample": "bare F.grid_sample (C=4 image+mask, then C=3 flow+mask) on B=%d, median of 5" % n}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--batch", type=int, default=64, help="batch elements per GPU (weak) / in the whole job (strong)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the 8-elements-per-GPU strong-scaling probe")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native, distributed as ofd
    _native.load_library()

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        ofd.enable_batch_sharding()          # batch-global early-exit flags over RCCL (one tiny all-reduce per new tensor)

    h, w = args.height, args.width
    if args.scaling == "strong":
        lo, hi = ofd.shard_bounds(args.batch, rank, world) if world > 1 else (0, args.batch)
        n, global_batch = hi - lo, args.batch
    else:
        n, global_batch = args.batch, args.batch * world

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(nb, steps, warmup, seed):
        """-> dict of per-step wall ms (validation-inclusive and cached) and HIP-event kernel ms of the two launches"""
        f1, f2, img, m1, m2, tm = make_inputs(nb, h, w, dev, seed=seed)

        def step_streaming():
            a, b = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)        # validation: one fused flag reduction + sync each
            b.apply(img, target_mask=tm, return_valid_area=True)
            a.combine_with(b, 3)
        for _ in range(warmup):
            step_streaming()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):                                          # EXACTLY `steps` timed steps
            step_streaming()
        barrier()
        el_stream = time.perf_counter() - t0
        # the same step on pre-built objects (validation cached per tensor version), with HIP events round each launch
        flow1, flow2 = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
        for _ in range(max(1, warmup)):
            flow2.apply(img, target_mask=tm, return_valid_area=True)
            flow1.combine_with(flow2, 3)
        barrier()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            flow2.apply(img, target_mask=tm, return_valid_area=True)
            ev[k][1].record()
            flow1.combine_with(flow2, 3)
            ev[k][2].record()
        barrier()
        el_cached = time.perf_counter() - t0
        return {"stream_s": el_stream, "cached_s": el_cached,
                "apply_ms": sum(e[0].elapsed_time(e[1]) for e in ev) / steps,       # HIP events on the launch stream
                "comb_ms": sum(e[1].elapsed_time(e[2]) for e in ev) / steps}

    r = measure(n, args.steps, args.warmup, seed=rank)
    elapsed, el_cached = r["stream_s"], r["cached_s"]
    t_apply, t_comb = r["apply_ms"], r["comb_ms"]
    if world > 1:
        t = torch.tensor([elapsed, el_cached], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, el_cached = float(t[0].item()), float(t[1].item())

    probe = None
    if world == 1 and not args.no_probe and (n, h, w) == (64, 1080, 1920):
        # BASELINE.json configs[3] on 8 GPUs leaves 8 elements per GPU: the same step at B = 8 on this GPU
        k8 = max(args.steps, 50)
        p8 = measure(8, k8, 5, seed=11)
        ms8, msc8 = p8["stream_s"] / k8 * 1e3, p8["cached_s"] / k8 * 1e3
        probe = {"batch": 8, "steps": k8, "ms_per_step": round(ms8, 4), "ms_per_step_cached": round(msc8, 4),
                 "kernel_ms_per_step": round(p8["apply_ms"] + p8["comb_ms"], 4),
                 "host_overhead_ms_per_step_cached": round(msc8 - (p8["apply_ms"] + p8["comb_ms"]), 4),
                 "implied_speedup_at_8_gpus": round((elapsed / args.steps * 1e3) / ms8, 2),
                 "implied_speedup_at_8_gpus_cached": round((el_cached / args.steps * 1e3) / msc8, 2),
                 "note": "global B=64 over 8 GPUs = this step at B=8 per GPU; speed-up = t(B=64 on 1 GPU) / t(B=8)"}

    # measured device-copy ceiling: a 1 GiB fp32 copy on the same stream, read + write bytes / time
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b

    if rank == 0:
        if args.traffic_bytes is None and (n, h, w) == (64, 1080, 1920):
            for name in ("r2_traffic.json", "r1_traffic.json"):          # PMC passes are separate runs (tools/profile_bench.sh)
                tj = os.path.join(ROOT, "profiles", name)
                if os.path.exists(tj):
                    with open(tj) as fh:
                        args.traffic_bytes = json.load(fh).get("traffic_bytes_per_launch")
                    break
        px = n * h * w
        gpx = global_batch * h * w
        ms_step = elapsed / args.steps * 1e3
        ach = BYTES_APPLY * px / (t_apply * 1e-3) / 1e9
        out = {
            "metric": "Mpix/s warped+composed (Flow(...) x2 from raw tensors, Flow.apply 't' C=3 + valid area, then "
                      "Flow.combine_with mode=3 -- the methods combine_flows / apply_flow wrap -- 1080p fp32)",
            "value": round(gpx / (elapsed / args.steps) / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "B=%d/GPU %dx%d fp32: Flow(f,'t',mask) x2 (validation), Flow.apply('t', C=3 image, "
                                   "target+flow masks, valid area) + combine_with(mode=3, 't', masks)" % (n, h, w),
                       "batch_per_gpu": n, "global_batch": global_batch, "height": h, "width": w,
                       "parallelism": "batch-sharded x%d (no data-path collective)" % world,
                       "bytes_per_px": BYTES_APPLY + BYTES_COMBINE},
            "roofline": {"bound": "hbm", "kernel": "warp_bwd_lds_column_kernel<4,3,valid> (Flow.apply 't')",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": args.traffic_bytes,
                         "algorithmic_bytes_per_launch": BYTES_APPLY * px,
                         "avg_launch_ms": round(t_apply, 4),
                         "device_copy_GBs": round(copy_gbs, 1), "frac_of_device_copy": round(ach / copy_gbs, 4)},
            "kernels": {"apply_ms": round(t_apply, 4), "apply_GBs": round(ach, 1),
                        "combine3_ms": round(t_comb, 4),
                        "combine3_GBs": round(BYTES_COMBINE * px / (t_comb * 1e-3) / 1e9, 1),
                        "validation_ms_per_step": round(ms_step - el_cached / args.steps * 1e3, 4)},
            "value_cached_flow_objects": round(gpx / (el_cached / args.steps) / 1e6, 1),
        }
        if probe is not None:
            out["strong_scaling_probe"] = probe
        if not args.no_cpu_baseline and world == 1:          # (the CPU baseline is a single-node, N = 1 figure)
            # 64 OpenMP threads is where the oracle peaks on the 2 x 64-core host of the GPU box (tools/cpu_threads_probe.py)
            out["cpu_baseline"] = cpu_baseline(h, w, args.cpu_seconds, min(os.cpu_count() or 1, 64))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
