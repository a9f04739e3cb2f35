#!/usr/bin/env python3
"""Forward warp (ofl_warp_bwd_f32, C = 3, masks, valid area) and gradient-wrt-flow kernels over a list of small shapes -- 200 calls back to
back -- for A/B of builds (thresholds between 1 / 2 / 4 tiles per block of the row-table kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
def t(fn, k=200):
    for _ in range(20): fn()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / k * 1e3)
    return sorted(ts)[1]
for (n, h, w) in ((8, 256, 256), (32, 256, 256), (4, 512, 512), (16, 384, 512), (16, 512, 512), (1, 1080, 1920), (2, 720, 1280), (4, 720, 1280), (2, 1080, 1920)):
    f = bench.smooth_flow(n, h, w, 4.0, 1, dev)
    img = torch.rand(n, 3, h, w, device=dev)
    sm = torch.rand(n, h, w, device=dev) > 0.1
    go = torch.randn(n, 3, h, w, device=dev)
    tf = t(lambda: _native.warp_bwd(f, img, src_mask=sm, flow_mask=sm, want_valid=True))
    tg = t(lambda: _native.warp_bwd_grad(f, img, go, want_src=False, want_flow=True))
    print("B=%2d %4dx%-4d (%5d tiles of 32x16)  forward %6.1f us   grad wrt flow %6.1f us" % (n, h, w, n * ((w + 31) // 32) * ((h + 15) // 16), tf, tg))
