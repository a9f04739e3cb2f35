import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
def t(fn, k=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
for (n, h, w) in ((8, 256, 256), (16, 384, 512), (4, 1080, 1920), (1, 1080, 1920), (2, 720, 1280)):
    f = bench.smooth_flow(n, h, w, 4.0, 1, dev)
    img = torch.rand(n, 3, h, w, device=dev)
    go = torch.randn(n, 3, h, w, device=dev)
    tf = t(lambda: _native.warp_bwd(f, img))
    tg = t(lambda: _native.warp_bwd_grad(f, img, go, want_src=False, want_flow=True))
    px = n * h * w
    print("B=%d %dx%d C=3: forward warp %.1f us (%.0f GB/s on 32 B/px), grad wrt flow %.1f us (%.0f GB/s on 40 B/px)" % (n, h, w, tf, 32 * px / tf / 1e3, tg, 40 * px / tg / 1e3))
