#!/usr/bin/env python3
"""A few launches of apply 's' and switch_ref on the bench flow (B = 16, sigma 8) for the rocprofv3 recipes; OFL_SPLAT_KERNEL picks
the gather kernel (0 round 6, 1 round 5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
n, h, w = int(os.environ.get("OFL_BATCH", "16")), 1080, 1920
_native.set_splat_gather_kernel(int(os.environ.get("OFL_SPLAT_KERNEL", "0")))
f1 = bench.smooth_flow(n, h, w, float(os.environ.get("OFL_SIGMA", "8")), 1000, dev)
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
S = ofl.Flow(f1, 's', m1)
for _ in range(5):
    S.apply(img, target_mask=tm, return_valid_area=True)
    S.switch_ref()
torch.cuda.synchronize()
