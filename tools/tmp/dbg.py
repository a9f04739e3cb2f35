import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch, bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
n, h, w = 16, 1080, 1920
sig = float(sys.argv[1]); kern = int(sys.argv[2])
_native.collect_splat_stats = True
_native.set_splat_gather_kernel(kern)
f1 = bench.smooth_flow(n, h, w, sig, 1000, dev); torch.cuda.synchronize(); print("flow ok", float(f1.abs().max()), flush=True)
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0); torch.cuda.synchronize(); print("inputs ok", flush=True)
S = ofl.Flow(f1, 's', m1); torch.cuda.synchronize(); print("flow obj ok", flush=True)
r = S.switch_ref(); torch.cuda.synchronize(); print("switch_ref ok", _native._last_splat_stats.cpu().tolist(), flush=True)
r = S.apply(img, target_mask=tm, return_valid_area=True); torch.cuda.synchronize(); print("apply ok", _native._last_splat_stats.cpu().tolist(), flush=True)
