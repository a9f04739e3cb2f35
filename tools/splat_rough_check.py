#!/usr/bin/env python3
"""The gather splat on ROUGH flows at the bench's frame size against the C oracle (B = 2, 1080p, sigma 12 ... 32, 3 channels + mask channel and
2 channels): masks bit for bit, values bit for bit wherever no band folded (statistics of the call) and within the fold tolerance elsewhere.
A by-hand check beside tests/test_gpu_fullsize.py (which pins sigma 8); run on the GPU box:  python tools/splat_rough_check.py [--sigma 12 16 24 32]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from oflibpytorch_amd import _native
from oracle import oracle
ap = argparse.ArgumentParser()
ap.add_argument("--sigma", type=float, nargs="+", default=[12.0, 16.0, 24.0, 32.0])
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = 2, 1080, 1920
_native.collect_splat_stats = True
for sigma in a.sigma:
    for c in (3, 2):
        flow = bench.smooth_flow(n, h, w, sigma, 4000 + int(sigma), dev)
        g = torch.Generator().manual_seed(77)
        data = (torch.rand(n, c, h, w, generator=g) * 100 - 20).to(dev)
        wm = bench.hole_mask(n, h, w, dev)
        ca = bench.hole_mask(n, h, w, dev).flip(1)
        out = _native.splat_fwd(flow, data, weight_mask=wm, chan_mask_a=ca, want_mask_chan=True, want_density=True, want_warped=True)
        st = _native._last_splat_stats.cpu().tolist()
        dd = np.concatenate([data.cpu().numpy(), ca.cpu().numpy()[:, None].astype(np.float32)], 1)
        ref, rwarped, rden = oracle.apply_s_flow(flow.cpu().numpy(), dd, wm.cpu().numpy(), True, return_density=True)
        o0, o1, o2, o3 = (x.cpu().numpy() for x in out[:4])
        assert np.array_equal(o3, rwarped), "warped mask differs"
        diff = (o0 != ref[:, :c]).any(1) | (o2 != rden) | (o1 != ref[:, c])
        np.testing.assert_allclose(o0, ref[:, :c], rtol=3e-5, atol=3e-3)
        np.testing.assert_allclose(o2, rden, rtol=3e-5, atol=1e-4)
        rows = int(diff.any(2).sum())
        print("sigma %4.0f C=%d: two-pass images %d, fold units %d, band units %d; pixels that differ from the oracle in any bit: %d in %d rows (max density %.0f)%s"
              % (sigma, c, st[2], st[1], st[3], int(diff.sum()), rows, rden.max(), "" if st[1] or st[2] or not diff.any() else "  <-- NOT EXACT WITHOUT A FOLD"), flush=True)
        assert st[1] or st[2] or not diff.any()
print("ok")
