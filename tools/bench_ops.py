#!/usr/bin/env python3
"""Secondary timings of the other hot-path operations (not the headline metric): ms and Mpix/s per op on one GPU.

    python tools/bench_ops.py [--batch 16] [--iters 10]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import oflibpytorch_amd as ofl  # noqa: E402
from oflibpytorch_amd import _native  # noqa: E402

_native.collect_splat_stats = True


def timeit(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--config5", action="store_true", help="BASELINE.json configs[4]: fp16-stored 4K flows, switch_ref s->t + mode 1")
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    n, h, w = a.batch, a.height, a.width
    px = n * h * w
    rows = []
    if a.config5:
        f1 = bench.smooth_flow(n, h, w, 8.0, 1000, dev).half()
        f2 = bench.smooth_flow(n, h, w, 8.0, 5000, dev).half()
        m = torch.ones(n, h, w, dtype=torch.bool, device=dev)

        def both():
            return ofl.Flow(f1, 's', m).switch_ref().combine_with(ofl.Flow(f2, 't', m), 1)
        A = ofl.Flow(f1, 's', m)
        At = A.switch_ref()
        B = ofl.Flow(f2, 't', m)
        rows.append(("Flow(fp16) + switch_ref s->t", 10, timeit(lambda: ofl.Flow(f1, 's', m).switch_ref(), a.iters)))
        rows.append(("combine_with mode 1 't'", 15, timeit(lambda: At.combine_with(B, 1), a.iters)))
        rows.append(("both, from fp16 storage", 25, timeit(both, a.iters)))
        print("config 5: B=%d %dx%d, flows stored in fp16 (bytes per pixel counted at fp16 storage), %d iters" % (n, h, w, a.iters))
        for name, bpp, t in rows:
            print("%-30s %8.3f ms  %9.1f Mpix/s  %7.1f GB/s algorithmic (%d B/px)" % (name, t * 1e3, px / t / 1e6, bpp * px / t / 1e9, bpp))
        return
    f1, f2, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    for ref in 'ts':
        A, B = ofl.Flow(f1, ref, m1), ofl.Flow(f2, ref, m2)
        if a.only:
            if a.only == "apply_s" and ref == 's':
                rows.append(("apply 's' C=3 +valid", 35, timeit(lambda: A.apply(img, target_mask=tm, return_valid_area=True), a.iters)))
            continue
        rows.append(("apply '%s' C=3 +valid" % ref, 35, timeit(lambda: A.apply(img, target_mask=tm, return_valid_area=True), a.iters)))
        if ref == 't':
            img8 = img.to(torch.uint8)
            rows.append(("apply 't' uint8 C=3 +valid", 26, timeit(lambda: A.apply(img8, target_mask=tm, return_valid_area=True), a.iters)))
        rows.append(("switch_ref %s" % ref, 18, timeit(lambda: A.switch_ref(), a.iters)))
        for mode in (3, 2, 1):
            rows.append(("combine_with mode %d '%s'" % (mode, ref), 27, timeit(lambda: A.combine_with(B, mode), a.iters)))
    print("B=%d %dx%d fp32, %d iters" % (n, h, w, a.iters))
    if _native._last_splat_stats is not None:
        st = _native._last_splat_stats.cpu().tolist()
        print("last gather splat: images on the two-pass path %d, fold tiles on LDS atomics %d" % (st[0], st[1]))
    for name, bpp, t in rows:
        print("%-28s %8.3f ms  %9.1f Mpix/s  %7.1f GB/s algorithmic (%d B/px)" % (name, t * 1e3, px / t / 1e6, bpp * px / t / 1e9, bpp))


if __name__ == "__main__":
    main()
