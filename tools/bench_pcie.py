#!/usr/bin/env python3
"""PCIe-inclusive rate of the path when the caller hands over HOST tensors (the reference's default device is the CPU: the host
mirror stages them through the GPU and returns results where the reference would put them).  Never bench.py's `value`: DESIGN.md
section 5 quotes this beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl

n, h, w = 8, 1080, 1920
f1, f2, img, m1, m2, tm = bench.make_inputs(n, h, w, torch.device('cpu'), 5)
for pin in (False, True):
    if pin:
        f1, f2, img, m1, m2, tm = [t.pin_memory() for t in (f1, f2, img, m1, m2, tm)]

    def step():
        a, b = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
        wv = b.apply(img, target_mask=tm, return_valid_area=True)
        c = a.combine_with(b, 3)
        return wv[0].shape, c.vecs.device
    step(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    moved = n * h * w * (62 + 8 + 8 + 1 + 1)      # operands in (flows are uploaded for validation AND for the kernels), results out
    print("host tensors (%s): B=%d step %.1f ms = %.1f Mpix/s (about %.1f GB/s over PCIe)" % ("pinned" if pin else "pageable", n, t * 1e3, n * h * w / t / 1e6, moved / t / 1e9))
