#!/usr/bin/env python3
"""Host-side (Python) cost of the hot calls on tiny frames (the GPU never back-pressures): wall time per call and a cProfile
listing -- what a step pays on the host while the GPU waits at small batch (strong scaling)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oflibpytorch_amd as ofl

dev = torch.device('cuda', 0)
n, h, w = 2, 64, 96
f1, f2 = torch.randn(n, 2, h, w, device=dev), torch.randn(n, 2, h, w, device=dev)
m1 = torch.rand(n, h, w, device=dev) > 0.1
m2 = torch.rand(n, h, w, device=dev) > 0.1
img = torch.rand(n, 3, h, w, device=dev)
tm = torch.rand(n, h, w, device=dev) > 0.1
A, B = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
S = ofl.Flow(f1, 's', m1)
calls = {"Flow()": lambda: ofl.Flow(f1, 't', m1), "apply": lambda: B.apply(img, target_mask=tm, return_valid_area=True),
         "combine3": lambda: A.combine_with(B, 3), "apply_s": lambda: S.apply(img, target_mask=tm, return_valid_area=True),
         "switch_ref": lambda: S.switch_ref()}
for name, fn in calls.items():
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-10s %.1f us per call (host)" % (name, (t1 - t0) / 2000 * 1e6))
for name in sys.argv[1:] or ("apply", "combine3", "Flow()"):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        calls[name]()
    pr.disable()
    torch.cuda.synchronize()
    print("==== %s" % name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
