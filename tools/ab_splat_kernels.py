#!/usr/bin/env python3
"""Round 6: the diet gather kernel (OFL_OPT_SPLAT_PATH 0) against round 5's (1) in ONE process, interleaved rounds, results
compared bit for bit; apply 's' (3 channels + valid area), switch_ref, mode 1 't' on the bench flows.

    python tools/ab_splat_kernels.py [--batch 16] [--sigma 2 8 12] [--rounds 7] [--iters 10]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--sigma", type=float, nargs="+", default=[2.0, 8.0, 12.0])
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--ops", nargs="+", default=["apply_s", "switch_ref", "combine1"])
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
_native.collect_splat_stats = True
_a = torch.empty(1 << 28, dtype=torch.float32, device=dev); _b = torch.empty_like(_a); _b.copy_(_a); torch.cuda.synchronize()
_e0, _e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_e0.record()
for _ in range(5):
    _b.copy_(_a)
_e1.record(); torch.cuda.synchronize()
print("box: device copy %.2f TB/s (1 GiB fp32, read + write bytes / time)" % (5 * 2 * _a.numel() * 4 / (_e0.elapsed_time(_e1) * 1e-3) / 1e12))
del _a, _b


def flat(res):
    if isinstance(res, ofl.Flow):
        return [res.vecs, res.mask]
    return list(res) if isinstance(res, tuple) else [res]


for sigma in a.sigma:
    f1 = bench.smooth_flow(n, h, w, sigma, 1000, dev)
    f2 = bench.smooth_flow(n, h, w, sigma, 5000, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    S, T1, T2 = ofl.Flow(f1, 's', m1), ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
    OPS = {"apply_s": (35, lambda: S.apply(img, target_mask=tm, return_valid_area=True)),
           "switch_ref": (18, lambda: S.switch_ref()),
           "combine1": (27, lambda: T1.combine_with(T2, 1))}
    for op in a.ops:
        bpp, fn = OPS[op]
        outs, stats = {}, {}
        for k in (1, 0):
            _native.set_splat_gather_kernel(k)
            outs[k] = [t.clone() for t in flat(fn())]
            torch.cuda.synchronize()
            stats[k] = _native._last_splat_stats.cpu().tolist()
        same = all(torch.equal(x, y) or (x.dtype.is_floating_point and bool(((x == y) | (x.isnan() & y.isnan())).all())) for x, y in zip(outs[0], outs[1]))
        nd = sum(int((x != y).sum()) for x, y in zip(outs[0], outs[1]))
        times = {0: [], 1: []}
        for rnd in range(a.rounds):
            for k in (1, 0):
                _native.set_splat_gather_kernel(k)
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / a.iters)
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        print("sigma %4.1f B=%d %-10s round 5 %.4f ms -> diet %.4f ms (%+.1f %%)  %s  fold tiles %d -> %d, two-pass images %d -> %d" % (
            sigma, n, op, med[1], med[0], 100.0 * (med[0] / med[1] - 1.0), "bit-identical" if same else "DIFFERENT (%d values)" % nd,
            stats[1][1], stats[0][1], stats[1][2], stats[0][2]), flush=True)
_native.set_splat_gather_kernel(0)
