#!/usr/bin/env python3
"""A/B of several builds of libofl_hip.so in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): every library is
dlopen'ed side by side and the package's binding is pointed at each in turn; HIP-event time per call of the chosen operations,
median / min over the rounds, on the bench flows.

    python tools/ab_libs.py --libs oflibpytorch_amd/libofl_hip.so tools/microbench/var/x.so --ops apply_s switch_ref apply_t combine3
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--ops", nargs="+", default=["apply_s", "switch_ref"])
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--sigma", type=float, default=8.0)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--check", action="store_true", help="compare every library's results with the first one's, bit for bit")
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
f1 = bench.smooth_flow(n, h, w, a.sigma, 1000, dev)
f2 = bench.smooth_flow(n, h, w, a.sigma, 5000, dev)
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)

libs = []
for path in a.libs:
    _native._lib = None
    libs.append((os.path.basename(path), _native.load_library(os.path.abspath(path))))


def use(lib):
    _native._lib = lib


use(libs[0][1])
S, T1, T2 = ofl.Flow(f1, 's', m1), ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
OPS = {
    "apply_s": (35, lambda: S.apply(img, target_mask=tm, return_valid_area=True)),
    "switch_ref": (18, lambda: S.switch_ref()),
    "apply_t": (35, lambda: T2.apply(img, target_mask=tm, return_valid_area=True)),
    "combine3": (27, lambda: T1.combine_with(T2, 3)),
    "combine1": (27, lambda: T1.combine_with(T2, 1)),
    "grad_t": (0, None),
}


def flat(res):
    if isinstance(res, ofl.Flow):
        return [res.vecs, res.mask]
    return list(res) if isinstance(res, tuple) else [res]


px = n * h * w
for op in a.ops:
    bpp, fn = OPS[op]
    times = {name: [] for name, _ in libs}
    ref = None
    for name, lib in libs:
        use(lib)
        out = flat(fn())
        torch.cuda.synchronize()
        if a.check:
            if ref is None:
                ref = [t.clone() for t in out]
            else:
                same = all(torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all()) if x.dtype.is_floating_point else torch.equal(x, y) for x, y in zip(ref, out))
                print("  %s: %s vs %s: %s" % (op, name, libs[0][0], "bit-identical" if same else "DIFFERENT"))
    for rnd in range(a.rounds):
        for name, lib in libs:
            use(lib)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / a.iters)
    for name, _ in libs:
        t = sorted(times[name])
        med, lo = t[len(t) // 2], t[0]
        print("%-12s sigma %4.1f B=%d  %-28s median %.4f ms  min %.4f ms   %7.1f GB/s = %.3f of 8 TB/s" % (op, a.sigma, n, name, med, lo, bpp * px / med / 1e6, bpp * px / med / 8e9))
