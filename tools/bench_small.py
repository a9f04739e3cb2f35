#!/usr/bin/env python3
"""Per-call time of the hot-path operations on SMALL frames (BASELINE configs[0] size: B = 1, 300 x 400), where launches, not
bytes, are what a call costs: ms per call back to back (one HIP-event pair round 200 calls)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl

dev = torch.device('cuda', 0)
for (n, h, w) in ((1, 300, 400), (4, 300, 400), (1, 1080, 1920)):
    f1, f2, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 1)
    S, T1, T2 = ofl.Flow(f1, 's', m1), ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
    ops = {"Flow()": lambda: ofl.Flow(f1, 't', m1), "apply t": lambda: T2.apply(img, target_mask=tm, return_valid_area=True),
           "combine3": lambda: T1.combine_with(T2, 3), "apply s": lambda: S.apply(img, target_mask=tm, return_valid_area=True),
           "switch_ref": lambda: S.switch_ref(), "combine1 t": lambda: T1.combine_with(T2, 1), "valid_target t": lambda: T1.valid_target()}
    for name, fn in ops.items():
        st = bench._loop_ms(fn, 200)
        print("B=%d %dx%d  %-14s %.4f ms per call (median; min %.4f)" % (n, h, w, name, st[1], st[2]))
