#!/usr/bin/env python3
"""Does the gather splat run faster in PASSES of fewer images (bin + gather of a pass back to back: the pass's flow may still sit in the 256 MB
Infinity Cache when the gather reads it again)?  apply 's' and switch_ref, B = 16, passes of 16 / 8 / 4 / 2 images.  python tools/splat_pass_probe.py [--sigma 2 8]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
ap = argparse.ArgumentParser()
ap.add_argument("--sigma", type=float, nargs="+", default=[2.0, 8.0])
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = 16, 1080, 1920
for sigma in a.sigma:
    f1 = bench.smooth_flow(n, h, w, sigma, 1000, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    S = ofl.Flow(f1, 's', m1)
    for name, fn in (("apply_s", lambda: S.apply(img, target_mask=tm, return_valid_area=True)), ("switch_ref", lambda: S.switch_ref())):
        out = []
        for k in (0, 8, 4, 2, 0):
            _native.set_splat_pass_images(k)
            fn(); fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            out.append("%s images per pass %.4f ms" % (k if k else "all", sorted(ts)[2]))
        _native.set_splat_pass_images(0)
        print("sigma %g %s: %s" % (sigma, name, " | ".join(out)), flush=True)
