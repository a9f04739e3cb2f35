#!/usr/bin/env python3
"""Row-table model, final form (see box_rows_model_wide2.py): origin of the 64-row table = min of the sample y at G x G points of the
tile (scalar loads) - MARGIN; share of PIXELS that are not staged (a row outside the table, or beyond 768 chunks) and staged px / output px."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
H, W = 1080, 1920
TW, TH = 64, 16
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
MARGIN = int(sys.argv[2]) if len(sys.argv) > 2 else 8
BUDGET = int(sys.argv[3]) if len(sys.argv) > 3 else 768
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
for sigma in (2.0, 8.0, 12.0, 16.0):
    f = bench.smooth_flow(1, H, W, sigma, 1000, torch.device('cpu'))[0].numpy()
    sx, sy = xs - f[0], ys - f[1]
    tot = 0; row_px = 0; missed = 0; tiles_missing = 0
    for ty in range(0, H, TH):
        for tx in range(0, W, TW):
            tot += 1
            yy = np.minimum(np.arange(ty, ty + TH), H - 1); xx = np.minimum(np.arange(tx, tx + TW), W - 1)
            X0 = np.clip(np.floor(sx[np.ix_(yy, xx)]), -2, W).astype(int); Y0 = np.clip(np.floor(sy[np.ix_(yy, xx)]), -2, H).astype(int)
            gy = np.minimum(ty + np.round(np.linspace(0, TH - 1, 2 if G < 3 else 3)).astype(int), H - 1)
            gx = np.minimum(tx + np.round(np.linspace(0, TW - 1, G)).astype(int), W - 1)
            org = int(np.floor(np.clip(sy[np.ix_(gy, gx)].min(), -2, H))) - MARGIN
            L = X0.reshape(TH, TW // 4, 4)
            cmin = (L.min(2)) >> 2; cmax = (L.max(2) + 1) >> 2
            YL = Y0.reshape(TH, TW // 4, 4); rlo = YL.min(2); rhi = YL.max(2) + 1
            tmin = np.full(64, 1 << 20); tmax = np.full(64, -1)
            for j in range(int((rhi - rlo).max()) + 1):
                r = np.minimum(rlo + j, rhi) - org
                ok = (r >= 0) & (r < 64)
                np.minimum.at(tmin, r[ok], cmin[ok]); np.maximum.at(tmax, r[ok], cmax[ok])
            rowy = org + np.arange(64)
            c0 = np.maximum(tmin, 0); c1 = np.minimum(tmax, (W - 1) >> 2)
            cw = np.where((tmax >= tmin) & (c1 >= c0) & (rowy >= 0) & (rowy < H), c1 - c0 + 1, 0)
            incl = np.cumsum(cw); st = incl <= BUDGET
            row_px += 4 * cw[st].sum()
            yr = Y0 - org
            need0 = (Y0 >= 0) & (Y0 < H); need1 = (Y0 + 1 >= 0) & (Y0 + 1 < H)
            inr = (yr >= 0) & (yr < 63)
            yrc = np.clip(yr, 0, 62)
            good = inr & (st[yrc] | ~need0) & (st[yrc + 1] | ~need1)
            good |= ~(need0 | need1)
            missed += np.count_nonzero(~good); tiles_missing += (~good).any()
    print("budget %d " % BUDGET + "G %d margin %d sigma %4.1f: staged px / output px %.3f, pixels not staged %.4f, tiles with such pixels %.4f" % (G, MARGIN, sigma, row_px / tot / (TW * TH), missed / tot / (TW * TH), tiles_missing / tot))
