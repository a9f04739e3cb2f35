#!/usr/bin/env python3
"""A GPU box health check that touches nothing of this repo: plain torch ops with allocator reuse (a box of the pool once faulted in
exactly this after an unrelated crash; every `gpurun` call of the round's last hours starts with it)."""
import sys, torch, torch.nn.functional as F
dev = torch.device('cuda', 0)
for sig in (2.0, 8.0, 12.0):
    for rep in range(3):
        g = torch.Generator(device='cpu').manual_seed(1000)
        lo = (torch.randn(16, 2, 27, 48, generator=g) * sig).to(dev)
        f = F.interpolate(lo, size=(1080, 1920), mode='bicubic', align_corners=True).contiguous()
        torch.cuda.synchronize()
        print("sigma", sig, "rep", rep, float(f.abs().max()), flush=True)
