#!/usr/bin/env python3
"""Round 6: how the gather splat's time depends on the blocks a CU holds -- the SAME kernel launched with extra dynamic LDS
(OFL_OPT_SPLAT_EXTRA_LDS), so that 3, 2 or 1 blocks of 512 threads fit a CU; plus the kernel's resources as the runtime reports
them (ofl_splat_gather_info).

    python tools/splat_occupancy.py [--batch 16] [--sigma 2 8] [--rounds 5] [--iters 10]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--sigma", type=float, nargs="+", default=[2.0, 8.0])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
for nc, what in ((3, "apply 's' (3 channels + mask channel)"), (2, "switch_ref (2 channels + mask channel)")):
    for extra in (0, 28672, 65536):
        print("%-40s extra LDS %6d: %s" % (what, extra, _native.splat_gather_info(nc, True, 0, True, extra)))
for sigma in a.sigma:
    f1 = bench.smooth_flow(n, h, w, sigma, 1000, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    S = ofl.Flow(f1, 's', m1)
    OPS = {"apply_s": lambda: S.apply(img, target_mask=tm, return_valid_area=True), "switch_ref": lambda: S.switch_ref()}
    for op, fn in OPS.items():
        times = {}
        for rnd in range(a.rounds):
            for extra in (0, 28672, 65536):
                _native.set_splat_extra_lds(extra)
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times.setdefault(extra, []).append(e0.elapsed_time(e1) / a.iters)
        _native.set_splat_extra_lds(0)
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        print("sigma %4.1f B=%d %-10s 3 blocks / CU %.4f ms   2 blocks %.4f ms (x %.2f)   1 block %.4f ms (x %.2f)" % (
            sigma, n, op, med[0], med[28672], med[28672] / med[0], med[65536], med[65536] / med[0]), flush=True)
