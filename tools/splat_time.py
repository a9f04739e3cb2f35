#!/usr/bin/env python3
"""Median HIP-event time of apply 's' (C = 3 + valid area) and switch_ref over a list of roughnesses; OFL_HIP_LIB picks the library
(A/B of builds in ALTERNATING PROCESSES on one box: tools/ab_splat_procs.sh).   python tools/splat_time.py [--sigma 2 8 12] [--batch 16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
ap = argparse.ArgumentParser()
ap.add_argument("--sigma", type=float, nargs="+", default=[2.0, 8.0, 12.0])
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--kernel", type=int, default=0)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
_native.collect_splat_stats = True
_native.set_splat_gather_kernel(a.kernel)
tag = os.path.basename(os.environ.get("OFL_HIP_LIB", "default")) + ("" if a.kernel == 0 else " (round-5 kernel)")
out = []
for sigma in a.sigma:
    f1 = bench.smooth_flow(n, h, w, sigma, 1000, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    S = ofl.Flow(f1, 's', m1)
    for name, fn in (("apply_s", lambda: S.apply(img, target_mask=tm, return_valid_area=True)), ("switch_ref", lambda: S.switch_ref())):
        fn(); fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / a.iters)
        st = _native._last_splat_stats.cpu().tolist()
        out.append("%s %.0f: %.4f (fold %d, banded %d%s)" % (name, sigma, sorted(ts)[len(ts) // 2], st[1], st[3], (", dbg %d %d" % (st[4], st[5])) if (st[4] or st[5]) else ""))
print("%-28s %s" % (tag, " | ".join(out)), flush=True)
