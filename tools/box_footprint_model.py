#!/usr/bin/env python3
"""CPU model of the staged footprint of the backward-warp kernel on the bench flows: plain bounding box, y-sheared box (the
product kernel), y- and x-sheared, and exact per-source-row extents -- mean staged pixels per output
pixel and the share of tiles whose box exceeds the LDS budget (1664 slots)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
def stats(sigma):
    f = bench.smooth_flow(1, 1080, 1920, sigma, 1000, torch.device('cpu'))[0].numpy()
    H, W = 1080, 1920
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    sx, sy = xs - f[0], ys - f[1]
    x0 = np.floor(sx).astype(int); y0 = np.floor(sy).astype(int)
    res = {k: [] for k in ("plain", "yshear", "xyshear", "rows")}
    for ty in range(0, H - 15, 16):
        for tx in range(0, W - 31, 32):
            X0 = x0[ty:ty+16, tx:tx+32]; Y0 = y0[ty:ty+16, tx:tx+32]
            xa, xb = X0.min(), X0.max() + 1; ya, yb = Y0.min(), Y0.max() + 1
            ca, cb = xa // 4, xb // 4
            cw = cb - ca + 1
            res["plain"].append(cw * 4 * (yb - ya + 1))
            # y-shear: slope per chunk column from flow difference at the two ends of the middle row
            mid = ty + 8
            dv = (sy[mid, min(tx+31, W-1)] - sy[mid, tx]) / 31.0 * 4.0       # rows per chunk column
            cc = (np.arange(tx, tx+32)[None, :] - f[0][ty:ty+16, tx:tx+32]) // 4   # approx chunk col of each px's taps
            chunk = (X0 // 4) - ca
            yp0 = Y0 - np.round(chunk * dv).astype(int); 
            chunk1 = ((X0 + 1) // 4) - ca
            yp1 = Y0 - np.round(chunk1 * dv).astype(int)
            lo = min(yp0.min(), yp1.min()); hi = max(yp0.max(), yp1.max()) + 1
            res["yshear"].append(cw * 4 * (hi - lo + 1))
            # + x-shear: chunk start per sheared row shifts by du/dy
            du = (sx[min(ty+15, H-1), tx+16] - sx[ty, tx+16]) / 15.0           # px per row
            rows = Y0 - ya
            xq0 = X0 - np.round(rows * du / 4).astype(int) * 4
            xq1 = X0 + 1 - np.round((rows) * du / 4).astype(int) * 4
            cw2 = (max(xq0.max(), xq1.max()) // 4) - (min(xq0.min(), xq1.min()) // 4) + 1
            res["xyshear"].append(cw2 * 4 * (hi - lo + 1))
            # exact per-source-row chunk extents
            tot = 0
            for yy in range(ya, yb + 1):
                m = (Y0 == yy) | (Y0 + 1 == yy)
                if m.any():
                    tot += ((X0[m].max() + 1) // 4 - X0[m].min() // 4 + 1) * 4
            res["rows"].append(tot)
    return {k: (np.mean(v) / 512.0, np.mean(np.array(v) > 1664)) for k, v in res.items()}
for s in (2.0, 8.0, 12.0):
    print(s, {k: ("%.2fx" % a, "%.3f over" % b) for k, (a, b) in stats(s).items()})
