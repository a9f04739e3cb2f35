#!/usr/bin/env python3
"""As box_rows_model_wide.py, but the row table in UNSHEARED image rows (64 entries about an estimated origin), dense chunk -> thread
mapping (768 chunks, 3327 slots): lane hulls of 4 pixels posted to rows [min yi, max yi + 1]."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
H, W = 1080, 1920
TW, TH = 64, 16
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 3
MARGIN = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
for sigma in (2.0, 8.0, 12.0, 16.0):
    f = bench.smooth_flow(1, H, W, sigma, 1000, torch.device('cpu'))[0].numpy()
    sx, sy = xs - f[0], ys - f[1]
    tot = 0; row_px = 0; row_nofit = 0; why = {}; rows_hist = []; spans = np.zeros(8, int)
    for ty in range(0, H, TH):
        for tx in range(0, W, TW):
            tot += 1
            yy = np.minimum(np.arange(ty, ty + TH), H - 1); xx = np.minimum(np.arange(tx, tx + TW), W - 1)
            X0 = np.clip(np.floor(sx[np.ix_(yy, xx)]), -2, W).astype(int); Y0 = np.clip(np.floor(sy[np.ix_(yy, xx)]), -2, H).astype(int)
            mid = min(ty + 8, H - 1); xa = min(tx, W - 1); xm = min(tx + 32, W - 1)
            L = X0.reshape(TH, TW // 4, 4)
            cmin = (L.min(2)) >> 2; cmax = (L.max(2) + 1) >> 2
            YL = Y0.reshape(TH, TW // 4, 4); rlo = YL.min(2); rhi = YL.max(2) + 1
            np.add.at(spans, np.minimum(rhi - rlo, 7).ravel(), 1)
            yb = min(ty + 15, H - 1); xb_ = min(tx + 63, W - 1)
            org = int(np.floor(np.clip(min(sy[ty, xa], sy[ty, xb_], sy[yb, xa], sy[yb, xb_]), -2, H))) - MARGIN
            bad = None
            if (rhi - rlo > NR - 1).any(): bad = "lane spans > %d rows" % NR
            if (rlo - org < 0).any() or (rhi - org > 63).any(): bad = bad or "rows outside the table"
            if bad is None:
                tmin = np.full(64, 1 << 20); tmax = np.full(64, -1)
                for j in range(NR):
                    r = np.minimum(rlo + j, rhi) - org
                    np.minimum.at(tmin, r, cmin); np.maximum.at(tmax, r, cmax)
                rowy = org + np.arange(64)
                c0 = np.maximum(tmin, 0); c1 = np.minimum(tmax, (W - 1) >> 2)
                cw = np.where((tmax >= tmin) & (c1 >= c0) & (rowy >= 0) & (rowy < H), c1 - c0 + 1, 0)
                if cw.sum() > 768 or 4 * cw.sum() > 3327: bad = "too many chunks"
                rows_hist.append(np.count_nonzero(cw))
            if bad is None: row_px += 4 * cw.sum()
            else: row_nofit += 1; why[bad] = why.get(bad, 0) + 1
    print("sigma %4.1f  unsheared row table: no fit %.3f, staged px / output px %.3f, rows mean %.1f max %d  %s  lane spans %s" % (
        sigma, row_nofit / tot, row_px / max(tot - row_nofit, 1) / (TW * TH), np.mean(rows_hist), max(rows_hist), why, (spans / spans.sum()).round(4).tolist()))
