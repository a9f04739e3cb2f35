#!/usr/bin/env python3
"""Measurement build only (tools/build_variant.sh prebox -DOFL_WARP_PREBOX_EXPERIMENT=1): Flow.apply 't' with the staging boxes
of every tile taken from a table made by a pre-pass (option 6 = 1) against the product path (option 6 = 0), same library, one
process, interleaved rounds; outputs compared bit for bit.  Under rocprofv3 --kernel-trace --stats the two column-kernel
instantiations and the pre-pass show up as separate kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

lib = _native.load_library(os.path.join(ROOT, "tools/microbench/var/prebox.so"))
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for sigma in (2.0, 8.0, 12.0):
    f2 = bench.smooth_flow(n, 1080, 1920, sigma, 5000, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, 1080, 1920, dev, 0)
    T2 = ofl.Flow(f2, 't', m2)
    fn = lambda: T2.apply(img, target_mask=tm, return_valid_area=True)
    outs = []
    for opt in (0, 1):
        assert lib.ofl_set_option(6, opt) == 0
        w, v = fn()
        torch.cuda.synchronize()
        outs.append((w.clone(), v.clone()))
    same = torch.equal(outs[0][1], outs[1][1]) and bool(((outs[0][0] == outs[1][0]) | (outs[0][0].isnan() & outs[1][0].isnan())).all())
    times = {0: [], 1: [], 2: []}
    for rnd in range(7):
        for opt in (0, 1, 2):
            lib.ofl_set_option(6, opt)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[opt].append(e0.elapsed_time(e1) / 10)
    lib.ofl_set_option(6, 0)
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    pre = med[2] - med[0]
    print("sigma %4.1f B=%d  product %.4f ms   pre-pass %.4f ms   column kernel on the table %.4f ms (%+.1f %%)   outputs %s"
          % (sigma, n, med[0], pre, med[1] - pre, 100 * (med[1] - pre - med[0]) / med[0], "bit-identical" if same else "DIFFERENT"))
