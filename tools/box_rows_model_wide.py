#!/usr/bin/env python3
"""CPU model of the staged warp's box on 64 x 16 tiles (the apply kernel's shape): the product's y-sheared rectangle against PER-ROW
chunk extents in the sheared frame as the row-table variant builds them (4-pixel lane hulls posted to at most 3 sheared rows, 64-entry
row table about an estimated origin, rows handled by 16-lane groups: 32 rows x 24 chunks, or 32-lane groups: 24 rows x 32 chunks).
Staged pixels per output pixel and the share of tiles that do not fit (profiles/r5_warp_row_extents.txt)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
H, W = 1080, 1920
TW, TH = 64, 16
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
def pitch(bw):                       # lds_pitch for TWQ = 16
    r = (16 - bw) % 32
    return bw + r
for sigma in (2.0, 8.0, 12.0, 16.0):
    f = bench.smooth_flow(1, H, W, sigma, 1000, torch.device('cpu'))[0].numpy()
    sx, sy = xs - f[0], ys - f[1]
    tot = 0; box_px = 0; box_over = 0; box_clip_px = 0; row_px = 0; row_nofit = 0; why = {}
    for ty in range(0, H, TH):
        for tx in range(0, W, TW):
            tot += 1
            yy = np.minimum(np.arange(ty, ty + TH), H - 1); xx = np.minimum(np.arange(tx, tx + TW), W - 1)
            X0 = np.clip(np.floor(sx[np.ix_(yy, xx)]), -2, W).astype(int); Y0 = np.clip(np.floor(sy[np.ix_(yy, xx)]), -2, H).astype(int)
            mid = min(ty + 8, H - 1); xa, xb = min(tx, W - 1), min(tx + TW - 1, W - 1)
            dx = sx[mid, xb] - sx[mid, xa]; dy = sy[mid, xb] - sy[mid, xa]
            sq = int(np.rint(np.clip(1024.0 * dy / dx, -4096, 4096))) if dx > 4 else 0
            shy = lambda c: (c * sq) >> 8
            s0, s1 = shy(X0 >> 2), shy((X0 + 1) >> 2)
            lo_r = Y0 - np.maximum(s0, s1); hi_r = Y0 + 1 - np.minimum(s0, s1)
            # the product: one sheared rectangle
            minx, maxx = max(X0.min(), 0), min(X0.max() + 1, W - 1)
            bx0 = minx & ~3; bw = ((maxx + 4) & ~3) - bx0; bh = hi_r.max() - lo_r.min() + 1
            fit = 16 * (1 + bh * pitch(bw)) <= 53248 and bh * (bw >> 2) <= 768
            box_over += (not fit); box_px += bh * bw if fit else 0
            # row table: lane hulls (4 px), rows lo..hi (at most 3), chunk range
            L = X0.reshape(TH, TW // 4, 4); 
            cmin = (L.min(2)) >> 2; cmax = (L.max(2) + 1) >> 2
            rlo = lo_r.reshape(TH, TW // 4, 4).min(2); rhi = hi_r.reshape(TH, TW // 4, 4).max(2)
            org = int(np.floor(sy[mid, xa])) - shy(int(np.floor(sx[mid, xa])) >> 2) - 32
            bad = None
            NR = int(sys.argv[1]) if len(sys.argv) > 1 else 4
            if (rhi - rlo > NR - 1).any(): bad = "lane spans > %d rows" % NR
            if (rlo - org < 0).any() or (rhi - org > 63).any(): bad = bad or "rows outside the table"
            if bad is None:
                tmin = np.full(64, 1 << 20); tmax = np.full(64, -1)
                for j in range(NR):
                    r = np.minimum(rlo + j, rhi) - org
                    np.minimum.at(tmin, r, cmin); np.maximum.at(tmax, r, cmax)
                c0 = np.maximum(tmin, 0); c1 = np.minimum(tmax, (W - 1) >> 2)
                cw = np.where((tmax >= tmin) & (c1 >= c0), c1 - c0 + 1, 0)
                nz = np.nonzero(cw)[0]
                if len(nz) == 0: bad = "empty"
                else:
                    nrows = nz[-1] - nz[0] + 1; mx = cw.max()
                    if mx <= 24 and nrows <= 32: pass
                    elif mx <= 32 and nrows <= 24: pass
                    else: bad = "rows %s chunks %s" % (">32" if nrows > 32 else (">24" if nrows > 24 else "ok"), ">32" if mx > 32 else (">24" if mx > 24 else "ok"))
            if bad is None: row_px += 4 * cw.sum()
            else: row_nofit += 1; why[bad] = why.get(bad, 0) + 1
    print("sigma %4.1f  rectangle: oversize %.3f, staged px / output px %.3f (fitting tiles) | row table: no fit %.3f, staged %.3f  %s" % (
        sigma, box_over / tot, box_px / max(tot - box_over, 1) / (TW * TH), row_nofit / tot, row_px / max(tot - row_nofit, 1) / (TW * TH), why))
