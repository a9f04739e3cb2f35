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
n, h, w = 16, 1080, 1920
names = ["0 list head + tile's own flow (round trip 1)", "1 cell init + barrier", "2 step loads arrive (round trip 2)", "3 hit test, ranks, records",
         "4 scan loop tail", "5 barrier after scan", "6 phase S + barriers", "7 phase C", "8 finalize, stores issued", "9 store drain + flag word"]
for sigma in (8.0, 2.0):
    f1 = bench.smooth_flow(n, h, w, sigma, 1003, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    A = ofl.Flow(f1, 's', m1)
    fn = lambda: A.apply(img, target_mask=tm, return_valid_area=True)
    fn(); torch.cuda.synchronize()
    import numpy as np
    blocks = 65280
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
