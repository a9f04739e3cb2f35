#!/usr/bin/env python3
"""Flow.apply 't' (C = 3, masks, valid area) at small batches, 200 calls back to back per event pair (bench.py's config-2 timing): for A/B of
builds through OFL_HIP_LIB (e.g. -DOFL_ROWS_SMALL_T=1: small launches on the row-table kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
_native.set_warp_path(int(os.environ.get('OFL_PATH', '0')))
dev = torch.device('cuda', 0)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6]:
    f1, f2, img, m1, m2, tm = bench.make_inputs(n, 1080, 1920, dev, 0)
    fl = ofl.Flow(f2, 't', m2)
    fn = lambda: fl.apply(img, target_mask=tm, return_valid_area=True)
    for _ in range(20): fn()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 200)
    t = sorted(ts)[2]
    print("B=%d  %.2f us  (%.1f %% of 8 TB/s)" % (n, t * 1e3, 35 * n * 1080 * 1920 / (t * 1e-3) / 8e10))
