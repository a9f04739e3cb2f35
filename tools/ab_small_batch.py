import os, sys
sys.path.insert(0, '/root/repo')
import torch, bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
libs = []
for path in sys.argv[1:]:
    _native._lib = None
    libs.append((os.path.basename(path), _native.load_library(os.path.abspath(path))))
for n in [int(v) for v in os.environ.get("AB_BATCHES", "1,2,4,8").split(",")]:
    f1, f2, img, m1, m2, tm = bench.make_inputs(n, 1080, 1920, dev, 2)
    _native._lib = libs[0][1]
    fl = ofl.Flow(f2, 't', m2); fa = ofl.Flow(f1, 't', m1)
    for name, lib in libs:
        _native._lib = lib
        for op, fn in (("apply", lambda: fl.apply(img, target_mask=tm, return_valid_area=True)), ("comb3", lambda: fa.combine_with(fl, 3))):
            st = bench._loop_ms(fn, 200)
            print("B=%d %-6s %-8s loop ms/call mean %.4f median %.4f min %.4f" % (n, op, name, st[0], st[1], st[2]))
