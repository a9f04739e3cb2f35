#!/usr/bin/env python3
"""Many-channel backward warp (Flow.apply 't' of an N-C-H-W feature tensor): the channel-loop kernel (one launch) against the
launches of 3 channels (`ofl_set_option(OFL_OPT_WARP_PATH, 5)`), bit for bit and timed.  Algorithmic bytes: 8 + 8 C + 1 B/px
(flow + flow mask read once, every plane read once and written once); + 2 B/px with a target mask and the valid area.

    python tools/bench_chan.py [--batch 8] [--channels 64 16 7 4] [--sigma 8]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--channels", type=int, nargs="+", default=[64, 16, 7, 4])
ap.add_argument("--sigma", type=float, default=8.0)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
lib = _native.load_library()
f = bench.smooth_flow(n, h, w, a.sigma, 1003, dev)
m = bench.hole_mask(n, h, w, dev)
tm = bench.hole_mask(n, h, w, dev).flip(1)
fl = ofl.Flow(f, 't', m)


def timed(fn):
    fn(); fn()
    t = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / a.iters)
    return sorted(t)[2]


for C in a.channels:
    g = torch.Generator(device='cpu').manual_seed(C)
    feat = torch.rand(n, C, h, w, generator=g).to(dev) if n * C * h * w < 2 ** 31 else torch.rand(n, C, h, w, device=dev)
    for name, fn, bpp in (("plain", lambda: fl.apply(feat), 8 + 8 * C + 1),
                          ("valid", lambda: fl.apply(feat, target_mask=tm, return_valid_area=True), 8 + 8 * C + 3)):
        lib.ofl_set_option(1, 5)
        ref = fn()
        t3 = timed(fn)
        lib.ofl_set_option(1, 0)
        got = fn()
        t1 = timed(fn)
        lib.ofl_set_option(1, 6)               # the channel loop on the sheared rectangle (the default stages per-row extents)
        rect = fn()
        t6 = timed(fn)
        lib.ofl_set_option(1, 0)
        tup = lambda v: v if isinstance(v, tuple) else (v,)
        same = all(torch.equal(x, y) for x, y in zip(tup(ref), tup(got))) and all(torch.equal(x, y) for x, y in zip(tup(rect), tup(got)))
        px = n * h * w
        print("C=%3d %-5s B=%d sigma %.0f: launches of 3 %.3f ms (%.3f of 8 TB/s)   channel loop, rectangle %.3f ms (%.3f)   channel loop, row extents %.3f ms (%.3f of 8 TB/s)   %s"
              % (C, name, n, a.sigma, t3, bpp * px / (t3 * 1e-3) / 8e12, t6, bpp * px / (t6 * 1e-3) / 8e12, t1, bpp * px / (t1 * 1e-3) / 8e12, "bit-identical" if same else "DIFFERENT"))
    del feat
