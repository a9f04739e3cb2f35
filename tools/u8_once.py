import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
n, h, w = 16, 1080, 1920
f1, f2, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
img8 = img.to(torch.uint8)
A = ofl.Flow(f2, 't', m2)
def t(fn, k=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for path in (6, 0, 6, 0):
    _native.set_warp_path(path)
    r = A.apply(img8, target_mask=tm, return_valid_area=True)
    print(path, r[0].dtype, "%.4f ms" % t(lambda: A.apply(img8, target_mask=tm, return_valid_area=True)))
_native.set_warp_path(0)
