#!/usr/bin/env python3
"""Flow.apply 't' (C = 3 + valid) on the bench flow, a few launches and nothing else: the program profilers are pointed at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f2 = bench.smooth_flow(n, 1080, 1920, 8.0, 5000, dev)
_, _, img, m1, m2, tm = bench.make_inputs(n, 1080, 1920, dev, 0)
T2 = ofl.Flow(f2, 't', m2)
for _ in range(reps):
    T2.apply(img, target_mask=tm, return_valid_area=True)
torch.cuda.synchronize()
print("done")
