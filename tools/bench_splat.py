#!/usr/bin/env python3
"""Timing of the forward-splat family (apply 's', switch_ref) on B x 1080p for a given flow roughness, with the gather
path's exactness statistics.   python tools/bench_splat.py [--batch 16] [--sigma 8]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
_native.collect_splat_stats = True

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--sigma", type=float, default=8.0)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
f1 = bench.smooth_flow(n, h, w, a.sigma, 1000, dev)
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
A = ofl.Flow(f1, 's', m1)


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters


px = n * h * w
for name, bpp, fn in (("apply 's' C=3 +valid", 35, lambda: A.apply(img, target_mask=tm, return_valid_area=True)),
                      ("switch_ref s->t", 18, lambda: A.switch_ref())):
    t = timeit(fn)
    st = _native._last_splat_stats.cpu().tolist()
    print("sigma %4.1f  %-22s %8.3f ms  %8.1f Mpix/s  %6.1f GB/s (%d B/px)   [launch fallback %d, tiles on LDS atomics %d, images on the two-pass path %d]"
          % (a.sigma, name, t * 1e3, px / t / 1e6, bpp * px / t / 1e9, bpp, st[0], st[1], st[2]))
