#!/usr/bin/env python3
"""Gradient-with-respect-to-the-flow kernel (ofl_warp_bwd_grad_f32, grad_flow only) at B x 1080p: row-table kernel against the column kernel
(OFL_OPT_WARP_PATH 6), identity and time, 1-3 channels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
n, h, w = 16, 1080, 1920
f1, f2, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
def t(fn, k=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for c in (1, 2, 3):
    src = img[:, :c].contiguous() if c != 2 else f1
    go = torch.randn(n, c, h, w, device=dev)
    res = {}
    for path in (6, 0, 6, 0):
        _native.set_warp_path(path)
        fn = lambda: _native.warp_bwd_grad(f2, src, go, want_src=False, want_flow=True)[1]
        r = fn()
        res.setdefault(path, r)
        print("C=%d path %d  %.4f ms" % (c, path, t(fn)))
    _native.set_warp_path(0)
    print("   identical:", bool(torch.equal(res[0], res[6])))
