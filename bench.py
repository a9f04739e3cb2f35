#!/usr/bin/env python3
"""Benchmark of the warp / compose hot path on MI355X (contract: see the task's bench.py section).

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:

    warped, valid = flow2.apply(image, target_mask, return_valid_area=True)    # 't' backward warp, C=3, 35 B/px
    flow3         = flow1.combine_with(flow2, mode=3)                          # fused composition,       27 B/px

on B x 1080 x 1920 fp32 per GPU (BASELINE.json configs[1]/[3]: Flow.apply + combine_flows mode 3).  Every step builds
its `Flow` objects from the raw tensors again -- construction is the reference's validation (`isfinite().all()`,
utils.py:98; here one fused flag reduction whose last block hands the words to the host: one launch, one wait per flow) --
so `value` is the streaming, validation-inclusive rate; the same step on pre-built (already validated, flag-cached) objects
is reported as `value_cached_flow_objects`.  The metric is Mpix/s = global batch * H * W / t_step ("warped+composed").

Timing: `--blocks` R blocks (default max(5, 500 / steps)) of EXACTLY `--steps` K steps each, every block bracketed by a
barrier + `torch.cuda.synchronize()` on both sides (MAX over the ranks per block); `value` / `ms_per_step` are the MEDIAN
block, `timing` carries min / median / max.  `roofline` is the dominant kernel (Flow.apply 't'), its average launch duration
from HIP events on the launch stream over >= 100 launches; `traffic` comes from an earlier rocprofv3 --pmc pass of this
command (`traffic_source` says which).  `secondary` (N = 1): the other single-GPU configurations of BASELINE.json --
configs[1] B = 1 apply 't', configs[2] B = 16 apply 's', configs[4] per GPU (B = 16 4K fp16: switch_ref + mode 1) -- each with
ms, algorithmic B/px and the fraction of 8 TB/s.

    --scaling weak   (default) B = --batch (64) PER GPU: every rank owns its own shard, no data-path collective
    --scaling strong           --batch (64) is the GLOBAL batch (BASELINE.json configs[3]): 64 / 32 / 16 / 8 per GPU at 1 / 2 / 4 / 8
At N = 1 the line also carries `strong_scaling_probe`: the same step at the 8 elements per GPU that config 4 leaves each
of 8 GPUs, with the per-step host overhead (wall time minus HIP-event kernel time) -- what bounds strong scaling.
Under torch.distributed.run the process group (backend "nccl" = RCCL) is initialised at every N, also N = 1.

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


def _event_ms(fn, iters, warm=3):
    """mean / median / min ms of `fn` over `iters` calls, HIP events on the launch stream"""
    for _ in range(warm):
        fn()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(iters)]
    torch.cuda.synchronize()
    for i in range(iters):
        ev[i][0].record()
        fn()
        ev[i][1].record()
    torch.cuda.synchronize()
    t = sorted(e[0].elapsed_time(e[1]) for e in ev)
    return sum(t) / iters, t[iters // 2], t[0]


def _loop_ms(fn, iters, blocks=5, warm=10):
    """ms per call of `fn` issued `iters` times back to back, ONE event pair round the loop (the way the headline step is
    timed): mean / median / min over `blocks` loops.  For a 2-Mpx launch an event pair per call measures the events."""
    for _ in range(warm):
        fn()
    t = []
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / iters)
    t.sort()
    return sum(t) / blocks, t[blocks // 2], t[0]


def secondary_lines(ofl, dev):
    """The other single-GPU configurations of BASELINE.json, timed the same way (HIP events round the whole operation, inputs
    resident, objects built inside the timed call where the config says so): ms, algorithmic B/px and the
    fraction of the 8 TB/s HBM peak those bytes / that time come to."""
    out = []

    def line(name, px, bpp, stats, extra=None):
        mean, median, lo = stats
        d = {"config": name, "ms": round(mean, 4), "ms_median": round(median, 4), "ms_min": round(lo, 4),
             "bytes_per_px": bpp, "Mpix_s": round(px / (mean * 1e-3) / 1e6, 1),
             "achieved_GBs": round(bpp * px / (mean * 1e-3) / 1e9, 1), "frac": round(bpp * px / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if extra:
            d.update(extra)
        out.append(d)
    h, w = 1080, 1920
    # config 2: B = 1 1080p fp32 't' Flow.apply (one 2-Mpx launch: bound by whichever of the host path and the 16-us kernel is longer)
    f1, f2, img, m1, m2, tm = make_inputs(1, h, w, dev, seed=2)
    fl = ofl.Flow(f2, 't', m2)
    pair = _event_ms(lambda: fl.apply(img, target_mask=tm, return_valid_area=True), 200, 10)
    line("configs[1]: B=1 1080x1920 fp32 Flow.apply 't' (C=3, masks, valid area)", h * w, BYTES_APPLY,
         _loop_ms(lambda: fl.apply(img, target_mask=tm, return_valid_area=True), 200),
         {"timing": "200 calls back to back per event pair, 5 loops (mean / median / min of the loops)",
          "ms_one_call_between_its_own_events": round(pair[1], 4)})
    # config 3: B = 16 1080p 's' forward-splat warp
    from oflibpytorch_amd import _native
    f1, f2, img, m1, m2, tm = make_inputs(16, h, w, dev, seed=3)
    fs = ofl.Flow(f1, 's', m1)
    _native.collect_splat_stats = True
    st3 = _event_ms(lambda: fs.apply(img, target_mask=tm, return_valid_area=True), 50, 5)
    stats = _native._last_splat_stats.cpu().tolist() if _native._last_splat_stats is not None else [0, 0, 0]
    _native.collect_splat_stats = False
    st3 = _event_ms(lambda: fs.apply(img, target_mask=tm, return_valid_area=True), 50, 2)
    line("configs[2]: B=16 1080x1920 fp32 Flow.apply 's' (forward splat, C=3, masks, valid area; sigma 8)", 16 * h * w, 35, st3,
         {"fold_tiles_on_lds_atomics": int(stats[1]), "images_on_two_pass_path": int(stats[2])})
    fs0 = ofl.Flow(smooth_flow(16, h, w, 8.0, 1000, dev), 's', m1)
    line("configs[2] on the flow of rounds 1-2 (sigma 8, seed 1000: profiles/r2_bench_ops_B16.txt)", 16 * h * w, 35,
         _event_ms(lambda: fs0.apply(img, target_mask=tm, return_valid_area=True), 50, 3))
    del fs0
    fs2 = ofl.Flow(smooth_flow(16, h, w, 2.0, 1003, dev), 's', m1)
    line("configs[2] on a smooth flow (sigma 2)", 16 * h * w, 35,
         _event_ms(lambda: fs2.apply(img, target_mask=tm, return_valid_area=True), 50, 3))
    line("switch_ref 's'->'t' B=16 1080x1920 fp32 (sigma 8)", 16 * h * w, 18, _event_ms(lambda: fs.switch_ref(), 50, 3))
    del f1, f2, img, m1, m2, tm, fs, fs2, fl
    # many channels: Flow.apply 't' of an N-C-H-W feature tensor, B = 8, C = 64 -- ONE launch that walks the channels
    # inside the block (warp_bwd_lds_chan_kernel); algorithmic bytes: flow 8 + flow mask 1 + every plane read once and written once
    torch.cuda.empty_cache()
    cf = ofl.Flow(smooth_flow(8, h, w, 8.0, 1003, dev), 't', hole_mask(8, h, w, dev))
    feat = torch.rand(8, 64, h, w, device=dev)
    line("many channels: B=8 1080x1920 C=64 fp32 Flow.apply 't' (feature tensor, flow mask; one launch, channel loop in the block; sigma 8)",
         8 * h * w, 8 + 8 * 64 + 1, _event_ms(lambda: cf.apply(feat), 20, 3),
         {"note": "bound by memory-side traffic, not by the bytes counted here: halo lines of a tile's box are re-fetched per channel group "
                  "(profiles/r5_chan_pmc.txt: 2.26 x the algorithmic reads at sigma 8, 1.48 x at sigma 2, ~6.4 TB/s of fabric traffic either way)"})
    del cf, feat
    torch.cuda.empty_cache()
    # config 5: B = 16 (the per-GPU share of B = 128 on 8 GPUs) 2160x3840, flows STORED in fp16: switch_ref s->t, then mode 1
    h, w = 2160, 3840
    g1 = smooth_flow(16, h, w, 8.0, 1005, dev).half()
    g2 = smooth_flow(16, h, w, 8.0, 5005, dev).half()
    m = torch.ones(16, h, w, dtype=torch.bool, device=dev)

    def cfg5():
        return ofl.Flow(g1, 's', m).switch_ref().combine_with(ofl.Flow(g2, 't', m), 1)
    line("configs[4] per GPU: B=16 2160x3840 fp16-stored flows, Flow() x2 + switch_ref 's'->'t' + combine_with mode 1", 16 * h * w, 25,
         _event_ms(cfg5, 10, 2), {"note": "10 B/px switch_ref + 15 B/px mode 1 't' at fp16 storage; validation of both operands inside"})
    torch.cuda.empty_cache()
    return out


def _sharded_child(route, steps):
    """bench.py --batch 8 in a child process under RANK=0 WORLD_SIZE=1: init_process_group("nccl"), enable_batch_sharding(), the
    collectives forced on -- the validation route every rank of an 8-GPU job takes.  -> the child's JSON line, or None"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--batch", "8", "--steps", str(steps), "--warmup", "5", "--blocks", "5",
           "--no-secondary", "--no-cpu-baseline", "--no-probe", "--force-collectives", "--flag-route", route]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception:  # noqa: BLE001
        return None


def _self_launch(n):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a child, relay the JSON line rank 0
    prints, return the child's exit code (with its stderr on failure).  Nothing here touches the GPU."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not lines:
        sys.stderr.write(res.stderr[-8000:])
        sys.stderr.write("\nbench.py: the %d-rank child (%s) failed with exit code %d\n" % (n, " ".join(cmd), res.returncode))
        return res.returncode if res.returncode != 0 else 1
    print(lines[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=0,
                    help="timed blocks of EXACTLY --steps steps each (0: max(5, 500 / steps)); `value` is the median block")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (BASELINE configs 2, 3, 5)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: strong (a global batch of --batch, BASELINE configs[3]) at N > 1, weak at N = 1 (the same job there)")
    ap.add_argument("--batch", type=int, default=64, help="batch elements per GPU (weak) / in the whole job (strong)")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks as child processes even at --gpus 1 (what `--gpus N`, N > 1, does when RANK is not in the environment)")
    ap.add_argument("--n1-ms", type=float, default=None, help="ms_per_step of the N = 1 run of the same job: adds speedup_vs_n1")
    ap.add_argument("--no-weak", action="store_true", help="N > 1, strong scaling: skip the weak-scaling figure (B = --batch per GPU)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the 8-elements-per-GPU strong-scaling probe")
    ap.add_argument("--force-collectives", action="store_true",
                    help="(the sharded probe's child) run the batch-sharded validation route on a communicator of ONE rank")
    ap.add_argument("--flag-route", choices=("host", "comm"), default="host",
                    help="batch sharding: flag words cross the ranks through shared memory (one node) or through the communicator")
    ap.add_argument("--shared-image", action="store_true",
                    help="the reference's 1 <-> N broadcast (utils.py:527-537): ONE B = 1 image + target mask owned by rank 0, sent to "
                         "every rank with distributed.broadcast_operand (RCCL broadcast over xGMI) once per timed block, INSIDE the "
                         "timed region, and warped by every rank's shard of the flows through the stride-0 batch broadcast")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (args.gpus > 1 or args.self_launch) and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as FRESH child processes, before this process has made
        # any GPU call (a process that has touched the GPU must never be replaced), and relay rank 0's line
        sys.exit(_self_launch(args.gpus))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d is running under WORLD_SIZE=%d: launch it with torch.distributed.run --nproc-per-node %d, "
                 "or without RANK / WORLD_SIZE in the environment (it then starts its ranks itself)" % (args.gpus, world, args.gpus))
    if args.scaling is None:
        # N > 1 measures BASELINE.json configs[3] as north_star states it: a GLOBAL batch of 64 sharded over the GPUs (strong
        # scaling, 8 elements per GPU at N = 8; utils.py:527-537, flow_class.py:896-897 partition on the batch axis) -- the
        # weak-scaling figure (B = 64 per GPU) rides along as `weak_scaling`.  N = 1: both are the same job.
        args.scaling = "strong" if world > 1 else "weak"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native, distributed as ofd
    _native.load_library()

    import torch.distributed as dist
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # under torch.distributed.run (also at N = 1)
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # "nccl" = RCCL over xGMI
        ofd.USE_HOST_EXCHANGE = args.flag_route == "host"
        ofd._force_collectives = bool(args.force_collectives)
        ofd.enable_batch_sharding()          # batch-global early-exit flags: 5 bits per new tensor cross the ranks (shared memory on one node, else one tiny all-reduce); a no-op at N = 1

    h, w = args.height, args.width
    if args.scaling == "strong":
        lo, hi = ofd.shard_bounds(args.batch, rank, world) if world > 1 else (0, args.batch)
        n, global_batch = hi - lo, args.batch
    else:
        n, global_batch = args.batch, args.batch * world

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def measure(nb, steps, warmup, seed, blocks):
        """`blocks` timed blocks of EXACTLY `steps` steps each (barrier + synchronize on both sides of every block), validation
        inclusive and on cached objects; then HIP events round every launch of max(steps, 100) more steps.
        -> per-block seconds (both flavours) and the per-launch kernel times in ms"""
        f1, f2, img, m1, m2, tm = make_inputs(nb, h, w, dev, seed=seed)
        bcast_ev = []
        if args.shared_image:
            # ONE image and ONE target mask for the whole job: rank 0 owns them, the other ranks hold empty buffers of the same shape
            g = torch.Generator(device='cpu').manual_seed(2000)
            own_img = (torch.rand(1, 3, h, w, generator=g) * 255).to(dev)
            own_tm = hole_mask(1, h, w, dev).flip(1)
            img = own_img.clone() if rank == 0 else torch.zeros_like(own_img)
            tm = own_tm.clone() if rank == 0 else torch.zeros_like(own_tm)

        def share():
            """the shared operand's trip: one broadcast of the image (24.9 MB at 1080p) and one of its mask (2.1 MB), timed by HIP events"""
            if not args.shared_image:
                return
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ofd.broadcast_operand(img, src=0)
            ofd.broadcast_operand(tm.view(torch.uint8), src=0)          # (bool bytes travel as uint8)
            e1.record()
            bcast_ev.append((e0, e1))

        def step_streaming():
            a, b = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)        # validation: one fused flag reduction + read-back each
            b.apply(img, target_mask=tm, return_valid_area=True)
            a.combine_with(b, 3)
        for _ in range(warmup):
            step_streaming()
        stream_s = []
        share()
        if args.shared_image and rank != 0:
            torch.cuda.synchronize()
            assert torch.equal(img, own_img) and torch.equal(tm, own_tm), "broadcast_operand did not deliver rank 0's operand"
        ranks_seen = 1
        for _ in range(blocks):
            barrier()
            t0 = time.perf_counter()
            share()                                                     # (inside the timed region, once per block)
            for _ in range(steps):                                      # EXACTLY `steps` timed steps per block
                step_streaming()
            ranks_seen = dist.get_world_size() if dist.is_initialized() else 1      # the communicator as seen INSIDE the timed region
            barrier()
            stream_s.append(time.perf_counter() - t0)
        # the same step on pre-built objects (validation cached per tensor version)
        flow1, flow2 = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)

        def step_cached():
            flow2.apply(img, target_mask=tm, return_valid_area=True)
            flow1.combine_with(flow2, 3)
        for _ in range(max(1, warmup)):
            step_cached()
        cached_s = []
        for _ in range(blocks):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_cached()
            barrier()
            cached_s.append(time.perf_counter() - t0)
        # per-launch kernel times: HIP events on the launch stream (torch's current stream is the one the C ABI gets)
        k = max(steps, 100)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(k)]
        barrier()
        for i in range(k):
            ev[i][0].record()
            flow2.apply(img, target_mask=tm, return_valid_area=True)
            ev[i][1].record()
            flow1.combine_with(flow2, 3)
            ev[i][2].record()
        barrier()
        ta = sorted(e[0].elapsed_time(e[1]) for e in ev)
        tc = sorted(e[1].elapsed_time(e[2]) for e in ev)
        bms = sorted(e0.elapsed_time(e1) for e0, e1 in bcast_ev[1:]) if len(bcast_ev) > 1 else []
        flow2.apply(img, target_mask=tm, return_valid_area=True)
        kernel_apply = _native.last_kernel_name()                  # the instantiation the launcher picked for THIS launch
        flow1.combine_with(flow2, 3)
        kernel_comb = _native.last_kernel_name()
        return {"stream_s": stream_s, "cached_s": cached_s, "launches": k, "broadcast_ms": (bms[len(bms) // 2] if bms else None),
                "rccl_ranks": ranks_seen, "kernel_apply": kernel_apply, "kernel_combine3": kernel_comb,
                "apply_ms": sum(ta) / k, "comb_ms": sum(tc) / k, "apply_ms_median": ta[k // 2], "comb_ms_median": tc[k // 2],
                "apply_ms_min": ta[0], "comb_ms_min": tc[0]}

    def med(v):
        return sorted(v)[len(v) // 2]

    blocks = args.blocks if args.blocks > 0 else max(5, -(-500 // args.steps))
    r = measure(n, args.steps, args.warmup, rank, blocks)
    t_apply, t_comb = r["apply_ms"], r["comb_ms"]
    stream_blocks, cached_blocks = r["stream_s"], r["cached_s"]
    if world > 1:                                  # every block's time is the MAX over the ranks
        t = torch.tensor([stream_blocks, cached_blocks], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        stream_blocks, cached_blocks = [float(v) for v in t[0].tolist()], [float(v) for v in t[1].tolist()]
    elapsed, el_cached = med(stream_blocks), med(cached_blocks)          # the median block is the reported one

    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_weak and not args.shared_image:
        # the weak-scaling figure beside the headline: B = --batch on EVERY GPU (a job `world` times as large)
        wk = measure(args.batch, args.steps, args.warmup, 100 + rank, max(3, blocks // 2))
        t = torch.tensor(wk["stream_s"], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wms = med([float(v) for v in t.tolist()]) / args.steps * 1e3
        weak = {"scaling": "weak", "batch_per_gpu": args.batch, "global_batch": args.batch * world, "ms_per_step": round(wms, 4),
                "value": round(args.batch * world * h * w / (wms * 1e-3) / 1e6, 1), "unit": "Mpix/s"}

    probe = None
    if world == 1 and not args.no_probe and not args.shared_image and (n, h, w) == (64, 1080, 1920):
        # BASELINE.json configs[3] on 8 GPUs leaves 8 elements per GPU: the same step at B = 8 on this GPU
        k8 = max(args.steps, 100)
        p8 = measure(8, k8, 5, 11, 5)
        ms8, msc8 = med(p8["stream_s"]) / k8 * 1e3, med(p8["cached_s"]) / k8 * 1e3
        probe = {"batch": 8, "steps": k8, "blocks": 5, "ms_per_step": round(ms8, 4), "ms_per_step_cached": round(msc8, 4),
                 "ms_per_step_min_max": [round(min(p8["stream_s"]) / k8 * 1e3, 4), round(max(p8["stream_s"]) / k8 * 1e3, 4)],
                 "kernel_ms_per_step": round(p8["apply_ms"] + p8["comb_ms"], 4),
                 "host_overhead_ms_per_step_cached": round(msc8 - (p8["apply_ms"] + p8["comb_ms"]), 4),
                 "implied_speedup_at_8_gpus": round((elapsed / args.steps * 1e3) / ms8, 2),
                 "implied_speedup_at_8_gpus_cached": round((el_cached / args.steps * 1e3) / msc8, 2),
                 "note": "global B=64 over 8 GPUs = this step at B=8 per GPU; speed-up = t(B=64 on 1 GPU) / t(B=8), median blocks"}
        # the same step THROUGH THE SHARDED VALIDATION ROUTE: a fresh child (RCCL before any other GPU work) with
        # init_process_group("nccl", world_size=1) + enable_batch_sharding() + the collectives forced on, once per flag route
        torch.cuda.empty_cache()
        for route, key in (("host", "sharded"), ("comm", "sharded_through_the_communicator")):
            got = _sharded_child(route, k8)
            if got is None:
                probe[key] = {"error": "child failed"}
                continue
            probe[key] = {"flag_route": "shared-memory exchange of the 5-bit word between the ranks of the node (distributed._HostExchange)"
                          if route == "host" else "ofl_flag_words_or_i32 + RCCL all-reduce (MAX) + copy + polled event",
                          "ms_per_step": got["ms_per_step"], "ms_per_step_cached": round(got["ms_per_step"] - got["kernels"]["validation_ms_per_step"], 4),
                          "launcher": got["config"]["launcher"]}
            probe["implied_speedup_at_8_gpus_" + key] = round((elapsed / args.steps * 1e3) / got["ms_per_step"], 2)

    secondary = None
    if world == 1 and not args.no_secondary and rank == 0:
        secondary = secondary_lines(ofl, dev)

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
    # a stream with the apply kernel's byte mix -- 22 bytes read for 13 written (flow 8 + image 12 + two masks | image 12 + mask 1) --
    # as plain torch elementwise work: out[13 parts] = f(in[22 parts]) over 1 GiB; what a streaming kernel of this mix reaches here
    parts = a.numel() // 22
    src22, dst13 = a[:22 * parts].view(22, parts), b[:13 * parts].view(13, parts)
    torch.add(src22[:13], src22[9:22], out=dst13)                       # (reads all 22 rows -- rows 9..12 twice, from cache -- writes 13)
    e0.record()
    for _ in range(5):
        torch.add(src22[:13], src22[9:22], out=dst13)
    e1.record()
    torch.cuda.synchronize()
    mix_gbs = 5 * 35 * parts * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b, src22, dst13

    if rank == 0:
        traffic_source = "--traffic-bytes" if args.traffic_bytes is not None else None
        if args.traffic_bytes is None and not args.shared_image and (n, h, w) == (64, 1080, 1920):
            for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):   # PMC passes are separate runs (tools/refresh_profiles_r5.sh)
                tj = os.path.join(ROOT, "profiles", name)
                if os.path.exists(tj):
                    with open(tj) as fh:
                        args.traffic_bytes = json.load(fh).get("traffic_bytes_per_launch")
                    traffic_source = ("profiles/%s: an EARLIER rocprofv3 --pmc pass of this command (2 x FETCH_SIZE + WRITE_SIZE per "
                                      "launch, MI355X_MICROARCH.md gfx950 correction), not measured in this run" % name)
                    break
        px = n * h * w
        gpx = global_batch * h * w
        ms_step = elapsed / args.steps * 1e3
        # --shared-image: the B = 1 image + target mask (13 B/px) are read once per launch, not once per batch element
        bytes_apply = (BYTES_APPLY - 13) + 13.0 / max(n, 1) if args.shared_image else BYTES_APPLY
        ach = bytes_apply * px / (t_apply * 1e-3) / 1e9
        per_step = sorted(b / args.steps * 1e3 for b in stream_blocks)
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
                       "parallelism": ("batch-sharded x%d; shared B=1 image + target mask: RCCL broadcast from rank 0 once per block "
                                       "(distributed.broadcast_operand), stride-0 batch broadcast in the warp" % world) if args.shared_image
                       else "batch-sharded x%d (no data-path collective)" % world,
                       "launcher": "torch.distributed (nccl)" if dist.is_initialized() else "single process",
                       "bytes_per_px": round(bytes_apply + BYTES_COMBINE, 3)},
            "timing": {"blocks": len(stream_blocks), "steps_per_block": args.steps, "reported": "median block",
                       "ms_per_step_min": round(per_step[0], 4), "ms_per_step_median": round(per_step[len(per_step) // 2], 4),
                       "ms_per_step_max": round(per_step[-1], 4),
                       "value_min": round(gpx / (per_step[-1] * 1e-3) / 1e6, 1), "value_max": round(gpx / (per_step[0] * 1e-3) / 1e6, 1)},
            "roofline": {"bound": "hbm", "kernel": r["kernel_apply"] + " (Flow.apply 't'; the name the library reports for the launch: ofl_last_kernel_name)",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": args.traffic_bytes,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(bytes_apply * px),
                         "avg_launch_ms": round(t_apply, 4), "median_launch_ms": round(r["apply_ms_median"], 4),
                         "min_launch_ms": round(r["apply_ms_min"], 4), "launches_timed": r["launches"],
                         "device_copy_GBs": round(copy_gbs, 1),
                         "read_mostly_stream_GBs": round(mix_gbs, 1),
                         "read_mostly_stream_note": "torch.add over 1 GiB with the apply kernel's byte mix (22 B read : 13 B written); a same-box "
                                                    "reference point for a streaming kernel of that mix, not a ceiling"},
            "rccl_ranks": r["rccl_ranks"],
            "kernels": {"apply_kernel": r["kernel_apply"], "combine3_kernel": r["kernel_combine3"],
                        "apply_ms": round(t_apply, 4), "apply_GBs": round(ach, 1),
                        "combine3_ms": round(t_comb, 4), "combine3_ms_median": round(r["comb_ms_median"], 4),
                        "combine3_GBs": round(BYTES_COMBINE * px / (t_comb * 1e-3) / 1e9, 1),
                        "combine3_frac": round(BYTES_COMBINE * px / (t_comb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "validation_ms_per_step": round(ms_step - el_cached / args.steps * 1e3, 4)},
            "value_cached_flow_objects": round(gpx / (el_cached / args.steps) / 1e6, 1),
        }
        if args.n1_ms is not None:
            out["speedup_vs_n1"] = round(args.n1_ms / ms_step, 3)
        if weak is not None:
            out["weak_scaling"] = weak
        if args.shared_image:
            out["broadcast_ms"] = None if r["broadcast_ms"] is None else round(r["broadcast_ms"], 4)
            out["broadcast_note"] = ("median HIP-event time of the two broadcasts (image 3 x H x W fp32 + mask H x W bytes) per timed block; "
                                     "at one rank without --force-collectives there is nobody to send to and the call returns at once")
        if secondary is not None:
            out["secondary"] = secondary
        if probe is not None:
            out["strong_scaling_probe"] = probe
        if not args.no_cpu_baseline and world == 1:          # (the CPU baseline is a single-node, N = 1 figure)
            # 64 OpenMP threads is where the oracle peaks on the 2 x 64-core host of the GPU box (tools/cpu_threads_probe.py)
            out["cpu_baseline"] = cpu_baseline(h, w, args.cpu_seconds, min(os.cpu_count() or 1, 64))
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
