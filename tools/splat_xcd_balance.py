#!/usr/bin/env python3
"""Work per XCD of the gather splat under its XCD-contiguous tile map: sum of list lengths (subtiles scanned) of the tiles each
XCD owns, and of the tiles whose lists are long (banded / fold candidates)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--sigma", type=float, default=8.0)
ap.add_argument("--seed", type=int, default=1003)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
_native.collect_splat_stats = 2
f = bench.smooth_flow(n, h, w, a.sigma, a.seed, dev)
m = bench.hole_mask(n, h, w, dev)
ofl.Flow(f, 's', m).switch_ref()
torch.cuda.synchronize()
ws = _native._last_splat_ws
tiles = n * ((w + 31) // 32) * ((h + 15) // 16)
off = 8 + ((n + 3) & ~3)
cnt = ws[off:off + tiles].float().cpu()
per = (tiles + 7) // 8
print("tiles %d, per XCD %d; list length mean %.1f max %d; stats %s" % (tiles, per, cnt.mean(), cnt.max(), ws[:3].tolist()))
tot = []
for k in range(8):
    c = cnt[k * per:(k + 1) * per]
    tot.append(float(c.sum()))
    print("XCD %d: subtiles scanned %9d   tiles > 64 entries %5d   > 128 %4d   max %d" % (k, c.sum(), (c > 64).sum(), (c > 128).sum(), c.max()))
mean = sum(tot) / 8
print("imbalance: max / mean = %.3f, min / mean = %.3f" % (max(tot) / mean, min(tot) / mean))
# per image
ti = tiles // n
im = cnt.reshape(n, ti).sum(1)
print("per image: " + " ".join("%.0f" % v for v in im.tolist()))
