#!/usr/bin/env python3
"""A/B of warp-kernel options in ONE process on the bench workload (B x 1080p): alternates the settings and prints
HIP-event times of Flow.apply ('t', C=3, masks) and combine_with(mode 3)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--sigma", type=float, default=8.0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--only", type=int, default=-1, help="run only with shear on (1) or off (0)")
args = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = args.batch, 1080, 1920
f1 = bench.smooth_flow(n, h, w, args.sigma, 1000, dev); f2 = bench.smooth_flow(n, h, w, args.sigma, 5000, dev)
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
flow1, flow2 = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)


def time_ops(k=10):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(2):
        flow2.apply(img, target_mask=tm, return_valid_area=True); flow1.combine_with(flow2, 3)
    torch.cuda.synchronize()
    ta = tc = 0.0
    for _ in range(k):
        ev[0].record(); flow2.apply(img, target_mask=tm, return_valid_area=True)
        ev[1].record(); flow1.combine_with(flow2, 3); ev[2].record()
        torch.cuda.synchronize()
        ta += ev[0].elapsed_time(ev[1]); tc += ev[1].elapsed_time(ev[2])
    return ta / k, tc / k


def set_shear(on):
    try:
        _native.set_warp_shear(on)
    except Exception:   # a build without the option
        pass


for rep in range(args.reps):
    for name, fn in (("shear on ", lambda: set_shear(True)), ("shear off", lambda: set_shear(False))):
        if args.only >= 0 and (name.strip() == "shear on") != bool(args.only):
            continue
        fn()
        a, c = time_ops()
        px = n * h * w
        print("%s  apply %.4f ms (%.1f%% of 8 TB/s)   combine3 %.4f ms (%.1f%%)" % (name, a, 35 * px / a / 8e7, c, 27 * px / c / 8e7), flush=True)
set_shear(True)
