#!/usr/bin/env python3
"""Round 6 probe: the forward splat of a batch as ONE call against TWO half-batch calls, back to back on one stream and side by side
on two HIP streams (the second launch of one half -- few, long blocks -- then runs beside the other half's first launch).
    python tools/splat_dual_stream_probe.py [--sigma 2 8 12] [--batch 16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
ap = argparse.ArgumentParser()
ap.add_argument("--sigma", type=float, nargs="+", default=[2.0, 8.0, 12.0])
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
side = [torch.cuda.Stream(device=dev) for _ in range(a.parts - 1)]


def timeit(fn, iters=10, rounds=5):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]


for sigma in a.sigma:
    f1 = bench.smooth_flow(n, h, w, sigma, 1000, dev)
    _, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
    S = ofl.Flow(f1, 's', m1)
    bounds = [(i * n // a.parts, (i + 1) * n // a.parts) for i in range(a.parts)]
    parts = [(ofl.Flow(f1[lo:hi], 's', m1[lo:hi]), img[lo:hi], tm[lo:hi]) for lo, hi in bounds]
    for name, one, part in (("apply_s", lambda: S.apply(img, target_mask=tm, return_valid_area=True), lambda p: p[0].apply(p[1], target_mask=p[2], return_valid_area=True)),
                            ("switch_ref", lambda: S.switch_ref(), lambda p: p[0].switch_ref())):
        def seq():
            for p in parts:
                part(p)

        def dual():
            main = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event(); ev.record(main)
            done = []
            for st, p in zip(side, parts[1:]):
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    part(p)
                    d = torch.cuda.Event(); d.record(st); done.append(d)
            part(parts[0])
            for d in done:
                main.wait_event(d)
        t1, t2, t3 = timeit(one), timeit(seq), timeit(dual)
        print("sigma %4.1f B=%d %-10s one call %.4f ms | %d parts on one stream %.4f ms | on %d streams %.4f ms (%+.1f %% against one call)"
              % (sigma, n, name, t1, a.parts, t2, a.parts, t3, 100 * (t3 / t1 - 1)), flush=True)
