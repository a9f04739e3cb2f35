#!/usr/bin/env python3
"""A few launches of the many-channel warp for rocprofv3 (tools/prof_pmc.sh): python tools/chan_once.py [sigma] [C] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import oflibpytorch_amd as ofl
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda', 0)
h, w = 1080, 1920
fl = ofl.Flow(bench.smooth_flow(n, h, w, sigma, 1003, dev), 't', bench.hole_mask(n, h, w, dev))
feat = torch.rand(n, C, h, w, device=dev)
for _ in range(4):
    out = fl.apply(feat)
torch.cuda.synchronize()
