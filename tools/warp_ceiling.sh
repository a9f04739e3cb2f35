#!/bin/bash
# Same-box ceiling evidence for the backward-warp kernel (run ON the GPU box from the repo root):
#   1. streaming kernels with Flow.apply's byte mix (22 B/px read, 13 B/px written, ten planes) by tile shape,
#   2. the device-to-device copy rate torch / the HIP runtime reach on this box,
#   3. the product kernel on smooth and bench flows.
out=gpurun_out/r2_ceiling; mkdir -p $out
hipcc -O3 --offload-arch=gfx950 -o /tmp/ss tools/microbench/stream_shapes.hip 2>/dev/null && /tmp/ss 64 > $out/stream_shapes.txt 2>&1
python3 - > $out/device_copy.txt 2>&1 <<'P'
import torch
a = torch.empty(1 << 28, dtype=torch.float32, device='cuda'); b = torch.empty_like(a)
for name, fn in (("copy_ 1 GiB fp32", lambda: b.copy_(a)),):
    for _ in range(3): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(20): fn()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print("%s: %.3f ms  %.1f GB/s read+written (%.1f%% of 8 TB/s)" % (name, ms, 2 * a.numel() * 4 / ms / 1e6, 2 * a.numel() * 4 / ms / 1e6 / 80))
P
for s in 0.5 2 8; do python3 tools/ab_warp.py --sigma $s --reps 2 --only 1 2>/dev/null | grep "shear on" | sed "s/^/sigma $s  /"; done > $out/product.txt
cat $out/stream_shapes.txt $out/device_copy.txt $out/product.txt
