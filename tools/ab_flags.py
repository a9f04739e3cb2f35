#!/usr/bin/env python3
"""HIP-event time of the validation reduction (ofl_flow_flags_f32) at the bench batch: for A/B of library builds (OFL_HIP_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
for n in (64, 8):
    f = bench.smooth_flow(n, 1080, 1920, 8.0, 1000, dev)
    m = bench.hole_mask(n, 1080, 1920, dev)
    for _ in range(5):
        _native.flow_flags(f, m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        _native.flow_flags(f, m)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print("B=%d  %.4f ms  %.0f GB/s" % (n, ms, 9 * n * 1080 * 1920 / ms / 1e6), end="   ")
print()
