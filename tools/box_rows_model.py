#!/usr/bin/env python3
"""CPU model of the staged warp's box on the bench flows: the product's y-sheared rectangle, the same with an additional x-shear
(chunk granularity), and exact per-row chunk extents in the sheared frame -- share of oversize tiles (box beyond the 26 KB / 384-chunk
budget) and staged pixels per output pixel (profiles/r4_warp_half_tile_retry.txt, section 3)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
def pitch(bw):
    P = bw
    while P % 16 != 8: P += 1
    return P
H, W = 1080, 1920
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
for sigma in (8.0, 12.0, 16.0):
    f = bench.smooth_flow(1, H, W, sigma, 1000, torch.device('cpu'))[0].numpy()
    sx, sy = xs - f[0], ys - f[1]
    res = {"y": [0, 0.0], "xy": [0, 0.0], "xy_exact": [0, 0.0]}
    tot = 0
    for ty in range(0, H - 15, 16):
        for tx in range(0, W - 31, 32):
            tot += 1
            X0 = np.floor(sx[ty:ty+16, tx:tx+32]).astype(int); Y0 = np.floor(sy[ty:ty+16, tx:tx+32]).astype(int)
            mid = ty + 8
            dx = sx[mid, tx+31] - sx[mid, tx]; dy = sy[mid, tx+31] - sy[mid, tx]
            sq = int(np.rint(np.clip(1024.0 * dy / dx, -4096, 4096))) if dx > 4 else 0      # rows per chunk column, Q8
            shy = lambda c: (c * sq) >> 8
            # taps: (X0, Y0), (X0+1, Y0), (X0, Y0+1), (X0+1, Y0+1)
            tx_ = np.stack([X0, X0 + 1, X0, X0 + 1]); ty_ = np.stack([Y0, Y0, Y0 + 1, Y0 + 1])
            yp = ty_ - shy(tx_ >> 2)                                   # sheared row
            # y-shear only
            minx, maxx = tx_.min(), tx_.max(); bx0 = minx & ~3; bw = ((maxx + 4) & ~3) - bx0
            bh = yp.max() - yp.min() + 1
            fit = 16 * (1 + bh * pitch(bw)) <= 26624 and bh * (bw >> 2) <= 384
            res["y"][0] += (not fit); res["y"][1] += bh * bw
            # + x-shear: chunk offset per sheared row: slope from the flow at column mid: dx per row
            cx = tx + 16
            ddx = sx[ty + 15, cx] - sx[ty, cx]; ddy = sy[ty + 15, cx] - sy[ty, cx]
            # x shift per SHEARED row, in chunks (Q8): d(x)/d(y') ~ ddx/ddy /4
            sxq = int(np.rint(np.clip(256.0 * (ddx / ddy) / 4.0, -1024, 1024))) if ddy > 2 else 0
            r = yp - yp.min()
            xq = tx_ - 4 * ((r * sxq) >> 8)
            minx, maxx = xq.min(), xq.max(); bx0 = minx & ~3; bw2 = ((maxx + 4) & ~3) - bx0
            fit = 16 * (1 + bh * pitch(bw2)) <= 26624 and bh * (bw2 >> 2) <= 384
            res["xy"][0] += (not fit); res["xy"][1] += bh * bw2
            # exact per-row chunk extents in the sheared frame
            totc = 0
            for rr in range(bh):
                m = r == rr
                if m.any():
                    totc += ((tx_[m].max()) >> 2) - ((tx_[m].min()) >> 2) + 1
            res["xy_exact"][0] += (16 * (1 + totc * 4) > 26624 or totc > 384); res["xy_exact"][1] += totc * 4
    print(sigma, {k: ("oversize %.3f" % (v[0] / tot), "staged px per output px %.2f" % (v[1] / tot / 512)) for k, v in res.items()})
