#!/usr/bin/env python3
"""Phase stamps (s_memtime, wave 0 of every block) of the row-table warp kernel from a build with -DOFL_ROWS_STAMPS=1
(tools/build_variant.sh rows_stamps -DOFL_ROWS_STAMPS=1; run with OFL_HIP_LIB=tools/microbench/var/rows_stamps.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
lib = _native.load_library()
dev = torch.device('cuda', 0)
n, h, w = 16, 1080, 1920
_, _, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
names = ["prologue", "coords (incl. wait for the flow)", "posts", "write (incl. wait for the staging loads)", "barrier A", "scan", "issue", "gather", "store", "barrier B", "start .. origins (40 scalar loads)", "first barrier"]
for s in [float(a) for a in sys.argv[1:]] or [0.5, 8.0, 16.0]:
    f = bench.smooth_flow(n, h, w, s, 5000, dev)
    fl = ofl.Flow(f, 't', m2)
    fl.apply(img, target_mask=tm, return_valid_area=True)
    out = (ctypes.c_ulonglong * 16)()
    lib.ofl_debug_rows_stamps(out, 1)
    fl.apply(img, target_mask=tm, return_valid_area=True)
    lib.ofl_debug_rows_stamps(out, 1)
    blocks = out[15]
    tot = sum(out[i] for i in range(12))
    print("sigma %.1f: %d blocks, %.0f ticks per block" % (s, blocks, tot / blocks))
    for i, nm in enumerate(names):
        print("   %-44s %8.0f ticks per block  %5.1f %%" % (nm, out[i] / blocks, 100.0 * out[i] / tot))
