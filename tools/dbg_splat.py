import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from oflibpytorch_amd import _native
from oracle import oracle
dev = torch.device('cuda', 0)
def _smooth(n, h, w, sigma, seed, dev):
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(n, 2, max(h // 12, 2), max(w // 12, 2), generator=g) * sigma
    return torch.nn.functional.interpolate(lo, size=(h, w), mode='bicubic', align_corners=True).contiguous().to(dev)
n, c, h, w = 2, 3, 64, 96
for sigma in (0.7, 1.5, 3.0):
    flow = _smooth(n, h, w, sigma, 33, dev)
    g = torch.Generator().manual_seed(10)
    data = (torch.rand(n, c, h, w, generator=g) * 100 - 20).to(dev)
    out = _native.splat_fwd(flow, data, occlude=False, want_density=True)
    ref, rwarped, rden = oracle.apply_s_flow(flow.cpu().numpy(), data.cpu().numpy(), None, False, return_density=True)
    o = out[0].cpu().numpy(); d = out[2].cpu().numpy()
    bad = (o != ref); badd = (d != rden)
    print("sigma", sigma, "value mismatches", bad.sum(), "of", bad.size, "density mismatches", badd.sum(), "max abs", np.abs(o - ref).max())
    if badd.sum():
        idx = np.argwhere(badd)[:8]
        for b_, y_, x_ in idx:
            print("   n=%d y=%d x=%d  got %.9g want %.9g" % (b_, y_, x_, d[b_, y_, x_], rden[b_, y_, x_]))
