#!/usr/bin/env python3
"""Where the gather splat's SECOND launch spends its time, unit by unit.  Needs a library built with -DOFL_SP2_UNITTIME=1
(tools/build_variant.sh unittime -DOFL_SP2_UNITTIME=1; OFL_HIP_LIB=tools/microbench/var/unittime.so): the second launch's kernel then
writes, into the redo list itself, when each band unit started (100 MHz wall clock) and how long it took.
    python tools/redo_unit_times.py [--sigma 8] [--batch 16] [--op apply_s|switch_ref]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
ap = argparse.ArgumentParser()
ap.add_argument("--sigma", type=float, default=8.0)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--op", default="apply_s")
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
_native.collect_splat_stats = 2
f1 = bench.smooth_flow(n, h, w, a.sigma, 1000, dev)
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
S = ofl.Flow(f1, 's', m1)
fn = (lambda: S.apply(img, target_mask=tm, return_valid_area=True)) if a.op == "apply_s" else (lambda: S.switch_ref())
fn(); fn(); torch.cuda.synchronize()
ws = _native._last_splat_ws.cpu().numpy().view(np.uint32)
tiles = n * ((w + 63) // 64) * ((h + 15) // 16)
kBinCap = (len(ws) - 8 - ((n + 3) & ~3) - ((tiles + 3) & ~3) - 32 * tiles) // tiles
off = 8 + ((n + 3) & ~3) + ((tiles + 3) & ~3) + kBinCap * tiles
nu = int(ws[6])
lst = ws[off:off + 2 * nu].reshape(nu, 2)
tile = lst[:, 0] & 0xffff
start = ((lst[:, 0] >> 16).astype(np.int64) * 8)                 # 10 ns ticks, modulo 2^19
dur = (lst[:, 1] >> 16).astype(np.int64) * 4                     # 10 ns ticks
rows = ((lst[:, 1] >> 8) & 0xff).astype(int) - (lst[:, 1] & 0xff).astype(int)
t0 = start.min()
start = (start - t0) % (1 << 19)
end = start + dur
print("sigma %g B=%d %s: %d units (rows: %s), list cap %d" % (a.sigma, n, a.op, nu, {int(k): int(v) for k, v in zip(*np.unique(rows, return_counts=True))}, kBinCap))
print("launch span %.1f us; sum of unit times %.1f us = %.1f us per slot over 512 slots; mean %.1f us, median %.1f, p90 %.1f, p99 %.1f, max %.1f"
      % (end.max() / 100, dur.sum() / 100, dur.sum() / 100 / 512, dur.mean() / 100, np.median(dur) / 100, np.percentile(dur, 90) / 100, np.percentile(dur, 99) / 100, dur.max() / 100))
# units running at a time, in 10 us steps
step = 1000
occ = [int(((start < t + step) & (end > t)).sum()) for t in range(0, int(end.max()), step)]
print("units in flight per 10 us: " + " ".join(str(o) for o in occ))
order = np.argsort(-dur)[:12]
print("longest units: " + " | ".join("tile %d rows %d: %.1f us from %.1f" % (tile[i], rows[i], dur[i] / 100, start[i] / 100) for i in order))
late = np.argsort(-end)[:8]
print("last to finish: " + " | ".join("tile %d rows %d: %.1f us from %.1f" % (tile[i], rows[i], dur[i] / 100, start[i] / 100) for i in late))
