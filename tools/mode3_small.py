import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
def t(fn, k=200):
    for _ in range(20): fn()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / k * 1e3)
    return sorted(ts)[1]
for n in (1, 2, 4, 6):
    f1, f2, img, m1, m2, tm = bench.make_inputs(n, 1080, 1920, dev, 0)
    A, B = ofl.Flow(f1, 't', m1), ofl.Flow(f2, 't', m2)
    out = []
    for path in (6, 0):
        _native.set_warp_path(path)
        out.append(t(lambda: A.combine_with(B, 3)))
    _native.set_warp_path(0)
    print("B=%d combine_with mode 3: sheared rectangle %.1f us (%.1f %%)   row tables %.1f us (%.1f %% of 8 TB/s)" % (n, out[0], 27 * n * 1080 * 1920 / out[0] / 8e4, out[1], 27 * n * 1080 * 1920 / out[1] / 8e4))
