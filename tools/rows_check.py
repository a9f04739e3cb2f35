#!/usr/bin/env python3
"""Row-table warp kernel (warp_bwd_rows_kernel, OFL_WARP_ROWS) against the column kernel's sheared rectangle (warp path 6), in ONE
process on the bench workload: bit-identity of Flow.apply 't' (C = 3, masks, valid area) and HIP-event times, over the roughness sweep
and a few hostile flows (pointing out of the frame, NaN, a fold)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--sigmas", type=float, nargs="*", default=[0.5, 2, 4, 8, 12, 16])
ap.add_argument("--no-hostile", action="store_true")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = args.batch, 1080, 1920
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
px = n * h * w


f1 = bench.smooth_flow(n, h, w, 8.0, 1000, dev)


def run(flow, path, timed=True, masks=True, op="apply"):
    _native.set_warp_path(path)
    fl = ofl.Flow(flow, 't', m2 if masks else None)
    if op == "apply":
        call = (lambda: fl.apply(img, target_mask=tm, return_valid_area=True)) if masks else (lambda: fl.apply(img))
    else:                                  # combine_with mode 3: the fused composition (the flow operand is the addend)
        other = ofl.Flow(f1, 't', m1 if masks else None)
        def call():
            r = other.combine_with(fl, 3)
            return (r.vecs, r.mask)
    out = call()
    t = 0.0
    if timed:
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
    _native.set_warp_path(0)
    return out, t


def same(a, b):
    if isinstance(a, tuple):
        return all(same(x, y) for x, y in zip(a, b))
    return bool(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b))


for s in args.sigmas:
    f = bench.smooth_flow(n, h, w, s, 5000, dev)
    t6s, t0s = [], []
    for _ in range(args.reps):             # alternating, the median of each
        o6, t = run(f, 6); t6s.append(t)
        o0, t = run(f, 0); t0s.append(t)
    t6, t0 = sorted(t6s)[len(t6s) // 2], sorted(t0s)[len(t0s) // 2]
    print("sigma %4.1f  rectangle %.4f ms (%.1f %%)   row tables %.4f ms (%.1f %%)   %+.1f %%   identical: %s" % (
        s, t6, 35 * px / t6 / 8e7, t0, 35 * px / t0 / 8e7, 100 * (t0 / t6 - 1), same(o0, o6)), flush=True)
    t6s, t0s = [], []
    for _ in range(args.reps):
        o6, t = run(f, 6, op="combine3"); t6s.append(t)
        o0, t = run(f, 0, op="combine3"); t0s.append(t)
    t6, t0 = sorted(t6s)[len(t6s) // 2], sorted(t0s)[len(t0s) // 2]
    print("   mode 3   rectangle %.4f ms (%.1f %%)   row tables %.4f ms (%.1f %%)   %+.1f %%   identical: %s" % (
        t6, 27 * px / t6 / 8e7, t0, 27 * px / t0 / 8e7, 100 * (t0 / t6 - 1), same(o0, o6)), flush=True)
if not args.no_hostile:
    g = torch.Generator(device='cpu').manual_seed(5)
    f = bench.smooth_flow(n, h, w, 8.0, 5000, dev)
    cases = {}
    a = f.clone(); a[:, 0] += 900.0; a[:, 1] -= 500.0; cases["pointing out of the frame"] = a
    a = f.clone(); a[:, :, ::97, ::61] = 3.0e38; a[:, :, 5::211, 3::113] = -3.0e38; cases["huge values sprinkled"] = a
    a = f * 6.0; cases["sigma 48 (folds)"] = a
    a = (torch.rand(n, 2, h, w, generator=g) * 40 - 20).to(dev); cases["white noise +-20"] = a
    a = f.clone(); a[:, 1] += torch.linspace(-300, 300, w, device=dev)[None, None, :].expand(n, h, w); cases["steep dv/dx"] = a
    a = f.clone(); a[:, 0] += torch.linspace(-900, 900, w, device=dev)[None, None, :].expand(n, h, w); cases["strong du/dx (compression / stretch)"] = a
    for name, fl in cases.items():
        for masks in (True, False):
            o6, _ = run(fl, 6, timed=False, masks=masks); o0, t0 = run(fl, 0, timed=True, masks=masks)
            o1, _ = run(fl, 1, timed=False, masks=masks)
            c6, _ = run(fl, 6, timed=False, masks=masks, op="combine3"); c0, _ = run(fl, 0, timed=False, masks=masks, op="combine3"); c1, _ = run(fl, 1, timed=False, masks=masks, op="combine3")
            print("%-40s masks %-5s identical to the rectangle: %s, to the generic kernel: %s   (%.3f ms)   mode 3: %s, %s" % (
                name, masks, same(o0, o6), same(o0, o1), t0, same(c0, c6), same(c0, c1)), flush=True)
