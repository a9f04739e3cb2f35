#!/usr/bin/env python3
"""Benchmark of the warp / compose hot path on MI355X (contract: see the task's bench.py section).

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:

    warped, valid = flow2.apply(image, target_mask, return_valid_area=True)    # 't' backward warp, C=3, 35 B/px
    flow3         = flow1.combine_with(flow2, mode=3)                          # fused composition,       27 B/px

on B x 1080 x 1920 fp32 per GPU (BASELINE.json configs[1]/[3]: Flow.apply + combine_flows mode 3; B = 64 per GPU,
weak scaling: every rank owns its own B-element shard, no data-path collective).  The metric is
Mpix/s = ranks * B * H * W / t_step ("warped+composed").

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


def cpu_baseline(h, w, seconds, threads):
    """The CPU oracle (oracle/: C restatement of the reference's algorithm, OpenMP over batch x rows) on a bounded
    sample of the same workload: B=2 of the same step, repeated for ~`seconds`."""
    import numpy as np
    from oracle import oracle
    oracle.set_threads(threads)
    n = 2
    f1, f2, img, m1, m2, tm = [t.numpy() for t in make_inputs(n, h, w, torch.device('cpu'), 77)]
    oracle.flow_apply(f2, 't', m2, img, tm)        # warm-up (page-in, thread pool)
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.flow_apply(f2, 't', m2, img, tm)
        oracle.combine_with(f1, m1, f2, m2, 3, 't')
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds or reps >= 200:
            break
    return {"value": round(reps * n * h * w / el / 1e6, 3), "unit": "Mpix/s", "cores": threads, "kind": "port",
            "sample": "oracle (C, OpenMP) Flow.apply('t',C=3,valid)+combine mode 3 on B=%d %dx%d fp32, %d reps in %.1f s"
                      % (n, h, w, reps, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="batch elements PER GPU (weak scaling)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        ofd.enable_batch_sharding()          # batch-global early-exit flags over RCCL (tiny, cached)

    n, h, w = args.batch, args.height, args.width
    f1, f2, img, m1, m2, tm = make_inputs(n, h, w, dev, seed=rank)
    flow1, flow2 = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)

    def step():
        warped, valid = flow2.apply(img, target_mask=tm, return_valid_area=True)
        flow3 = flow1.combine_with(flow2, 3)
        return warped, valid, flow3

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        warped, valid = flow2.apply(img, target_mask=tm, return_valid_area=True)
        ev[k][1].record()
        flow3 = flow1.combine_with(flow2, 3)
        ev[k][2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    t_apply = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps      # ms per launch, HIP events, same stream
    t_comb = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # secondary figure: the same step with the Flow objects rebuilt from raw tensors every step
    # (construction = one fused validation pass + host sync per flow)
    barrier()
    t1 = time.perf_counter()
    k2 = max(2, args.steps // 4)
    for _ in range(k2):
        a, b = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
        b.apply(img, target_mask=tm, return_valid_area=True)
        a.combine_with(b, 3)
    barrier()
    el2 = time.perf_counter() - t1

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
            tj = os.path.join(ROOT, "profiles", "r1_traffic.json")      # PMC passes are separate runs (tools/profile_bench.sh)
            if os.path.exists(tj):
                with open(tj) as fh:
                    args.traffic_bytes = json.load(fh).get("traffic_bytes_per_launch")
        px = n * h * w
        ms_step = elapsed / args.steps * 1e3
        ach = BYTES_APPLY * px / (t_apply * 1e-3) / 1e9
        out = {
            "metric": "Mpix/s warped+composed (Flow.apply 't' C=3 +valid area, then combine_flows mode=3, 1080p fp32)",
            "value": round(world * px / (elapsed / args.steps) / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "B=%d/GPU %dx%d fp32: Flow.apply('t', C=3 image, target+flow masks, valid area) + "
                                   "combine_with(mode=3, 't', masks)" % (n, h, w),
                       "batch_per_gpu": n, "global_batch": n * world, "height": h, "width": w,
                       "parallelism": "batch-sharded x%d (no data-path collective)" % world,
                       "bytes_per_px": BYTES_APPLY + BYTES_COMBINE},
            "roofline": {"bound": "hbm", "kernel": "warp_bwd_lds_kernel<3,valid> (Flow.apply 't')",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": args.traffic_bytes,
                         "algorithmic_bytes_per_launch": BYTES_APPLY * px,
                         "avg_launch_ms": round(t_apply, 4),
                         "device_copy_GBs": round(copy_gbs, 1), "frac_of_device_copy": round(ach / copy_gbs, 4)},
            "kernels": {"apply_ms": round(t_apply, 4), "apply_GBs": round(ach, 1),
                        "combine3_ms": round(t_comb, 4),
                        "combine3_GBs": round(BYTES_COMBINE * px / (t_comb * 1e-3) / 1e9, 1),
                        "step_GBs": round((BYTES_APPLY + BYTES_COMBINE) * px / (ms_step * 1e-3) / 1e9, 1)},
            "value_with_flow_construction": round(world * px / (el2 / k2) / 1e6, 1),
        }
        if not args.no_cpu_baseline and world == 1:          # (the CPU baseline is a single-node, N = 1 figure)
            # 64 OpenMP threads is where the oracle peaks on the 2 x 64-core host of the GPU box (tools/cpu_threads_probe.py)
            out["cpu_baseline"] = cpu_baseline(h, w, args.cpu_seconds, min(os.cpu_count() or 1, 64))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
