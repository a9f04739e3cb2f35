#!/usr/bin/env python3
"""Where a bench step's wall time goes at a given batch (default 8: the per-GPU share of BASELINE config 4 on 8 GPUs): host time
stamps after every call of the step, medians over many steps, next to the HIP-event time of the two kernels."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device('cuda', 0)
h, w = 1080, 1920
f1, f2, img, m1, m2, tm = bench.make_inputs(a.batch, h, w, dev, 11)
rows = []
for i in range(a.steps + 10):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    A = ofl.Flow(f1, 't', m1); t.append(time.perf_counter())
    B = ofl.Flow(f2, 't', m2); t.append(time.perf_counter())
    B.apply(img, target_mask=tm, return_valid_area=True); t.append(time.perf_counter())
    A.combine_with(B, 3); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if i >= 10:
        rows.append([(t[k + 1] - t[k]) * 1e6 for k in range(5)] + [(t[5] - t[0]) * 1e6])
med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
print("B=%d  Flow(f1) %.1f us | Flow(f2) %.1f us | apply returns after %.1f us | combine_with returns after %.1f us | drain %.1f us | step %.1f us"
      % (a.batch, med(0), med(1), med(2), med(3), med(4), med(5)))
