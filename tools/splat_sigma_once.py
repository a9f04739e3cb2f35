#!/usr/bin/env python3
"""One timed pass of apply 's' / switch_ref at a given roughness (OFL_HIP_LIB picks the library); progress line per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
h, w = 1080, 1920
_native.collect_splat_stats = True
f1 = bench.smooth_flow(n, h, w, sigma, 1000, dev)
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
S = ofl.Flow(f1, 's', m1)
for name, fn in (("apply_s", lambda: S.apply(img, target_mask=tm, return_valid_area=True)), ("switch_ref", lambda: S.switch_ref())):
    for i in range(3):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print("sigma %.1f B=%d %s call %d: %.3f ms  stats %s" % (sigma, n, name, i, (time.perf_counter() - t0) * 1e3, _native._last_splat_stats.cpu().tolist()), flush=True)
