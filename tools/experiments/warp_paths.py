#!/usr/bin/env python3
"""apply 't' and mode 3 through each staged warp kernel at a given batch: auto, pair (two tiles per block), single (one tile per block)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
dev = torch.device('cuda', 0)
for n in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,16,64").split(",")]:
    f1, f2, img, m1, m2, tm = bench.make_inputs(n, 1080, 1920, dev, 2)
    fl = ofl.Flow(f2, 't', m2); fa = ofl.Flow(f1, 't', m1)
    for rep in range(2):
        for path, name in ((0, "auto"), (3, "pair"), (4, "single")):
            _native.set_warp_path(path)
            try:
                for op, fn in (("apply", lambda: fl.apply(img, target_mask=tm, return_valid_area=True)), ("mode3", lambda: fa.combine_with(fl, 3))):
                    st = bench._loop_ms(fn, 50)
                    print("B=%d %-6s %-7s ms/call median %.4f min %.4f" % (n, op, name, st[1], st[2]))
            finally:
                _native.set_warp_path(0)
