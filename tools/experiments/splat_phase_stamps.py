#!/usr/bin/env python3
"""Cycles per phase of the gather splat kernel, from the stamp build (tools/experiments/r4_splat_phase_stamps.patch ->
tools/microbench/var/stamps.so): s_memtime deltas of wave 0 of every block, summed; waits are forced at the stamps, so the
split is a picture of the dependency chain, not of the un-instrumented schedule."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

lib = _native.load_library(os.path.join(ROOT, "tools/microbench/var/stamps.so"))
dev = torch.device('cuda', 0)
n, h, w = int(os.environ.get('STAMP_BATCH', '16')), 1080, 1920
names = ["0 list head + tile's own flow (round trip 1)", "1 cell init + barrier", "2 step loads arrive (round trip 2)", "3 hit test, ranks, records",
         "4 scan loop tail", "5 barrier after scan", "6 phase S + barriers", "7 phase C", "8 finalize, stores issued", "9 store drain + flag word"]
for sigma in [float(v) for v in os.environ.get("STAMP_SIGMAS", "8,2").split(",")]:
    f1 = bench.smooth_flow(n, h, w, sigma, 1003, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    A = ofl.Flow(f1, 's', m1)
    fn = lambda: A.apply(img, target_mask=tm, return_valid_area=True)
    fn(); torch.cuda.synchronize()
    import numpy as np
    tw, th, _ = _native.splat_tile_geometry()
    blocks = ((n * ((w + tw - 1) // tw) * ((h + th - 1) // th) + 7) // 8) * 8
    buf = np.zeros(blocks * 16, dtype=np.uint32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    lib.ofl_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), blocks)
    b = buf.reshape(blocks, 16).astype(np.float64)
    b = b[b[:, 15] > 0]
    per = b[:, :10].mean(0)
    tot = per.sum()
    print("sigma %.0f: %d blocks, call %.3f ms (instrumented), mean block life %.0f ticks of s_memtime (100 MHz: %.2f us); p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
        sigma, len(b), e0.elapsed_time(e1), tot, tot / 100.0, *np.percentile(b[:, :10].sum(1), [50, 90, 99, 100])))
    for k in range(10):
        print("   %-48s %8.1f ticks/block  %5.1f %%" % (names[k], per[k], 100.0 * per[k] / tot))
    life = b[:, :10].sum(1)
    order = np.argsort(-life)[:8]
    print("   slowest blocks (ticks per phase 0..9):")
    for o in order:
        print("      life %7.0f : %s | scans %d steps %d %s" % (life[o], " ".join("%6.0f" % v for v in b[o, :10]), b[o, 10], b[o, 11], "FOLD" if b[o, 15] == 2 else ""))
    print("   blocks by number of scans: " + ", ".join("%d scans: %d blocks, mean life %.0f" % (k, int((b[:, 10] == k).sum()), life[b[:, 10] == k].mean()) for k in sorted(set(b[:, 10].astype(int)))))
    for k in sorted(set(b[:, 10].astype(int))):
        g = b[b[:, 10] == k]
        print("      %d scans: mean ticks per phase %s  | steps %.1f" % (k, " ".join("%6.0f" % v for v in g[:, :10].mean(0)), g[:, 11].mean()))
    print("   fold tiles: %d, mean life %.0f" % (int((b[:, 15] == 2).sum()), life[b[:, 15] == 2].mean() if (b[:, 15] == 2).any() else 0))
    # list length and records wanted (first scan) against the number of scans
    nl, nr = b[:, 12], b[:, 13]
    for lo, hi in ((0, 32), (32, 40), (40, 48), (48, 56), (56, 64), (64, 1000)):
        sel = (nl >= lo) & (nl < hi)
        if sel.any():
            print("      list %3d..%3d: %6d tiles, records wanted mean %.0f, > %d records: %.1f %%, mean life %.0f" % (
                lo, hi, int(sel.sum()), nr[sel].mean(), 896, 100.0 * (nr[sel] > 896).mean(), life[sel].mean()))
    srt = np.sort(life)[::-1]
    print("   sum of the slowest 1 %% of the blocks: %.1f %% of all block time; blocks over 3x the median: %d" % (
        100.0 * srt[:max(1, len(srt) // 100)].sum() / life.sum(), int((life > 3 * np.median(life)).sum())))

