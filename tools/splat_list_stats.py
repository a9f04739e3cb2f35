#!/usr/bin/env python3
"""Distribution of the per-destination-tile list lengths of the gather splat on the bench flows (how many source subtiles
a tile scans, how many half steps of 256 threads that takes)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--sigma", type=float, nargs='+', default=[2.0, 8.0])
args = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = args.batch, 1080, 1920
_native.collect_splat_stats = 2
for sigma in args.sigma:
    f = bench.smooth_flow(n, h, w, sigma, 1000, dev)
    ofl.Flow(f, 's').switch_ref()
    torch.cuda.synchronize()
    ws = _native._last_splat_ws
    tiles = n * ((w + 63) // 64) * ((h + 15) // 16)      # (64 x 16 destination tiles)
    off = 8 + ((n + 3) & ~3)
    cnt = ws[off:off + tiles].float()
    q = torch.quantile(cnt, torch.tensor([0.1, 0.5, 0.9, 0.99], device=dev)).tolist()
    print("sigma %4.1f: list length mean %.1f  p10/50/90/99 %s  max %d   <=16: %.2f  <=32: %.2f  <=64: %.2f" % (
        sigma, cnt.mean().item(), [int(v) for v in q], int(cnt.max().item()),
        (cnt <= 16).float().mean().item(), (cnt <= 32).float().mean().item(), (cnt <= 64).float().mean().item()))
