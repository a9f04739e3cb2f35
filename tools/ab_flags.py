#!/usr/bin/env python3
"""The validation wait, two routes side by side in ONE process (B = 64 and B = 8 1080p flows with masks):

  copy   ofl_flow_flags_f32 (memset + reduction) -> pinned copy -> polled event          (rounds 1-2)
  host   ofl_flow_flags_host: the reduction's last block writes the words to host-visible memory, the host polls one word

per call: wall time from an idle stream (what a constructor waits for), HIP-event time of the device work alone, and the wall time
of `Flow(f, 't', m)` itself.  Interleaved rounds, medians."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native, utils

dev = torch.device('cuda', 0)
LIBS = [a for a in sys.argv[1:] if a.endswith('.so')]            # optional: several builds, interleaved (host route only)
libs = []
for path in LIBS:
    _native._lib = None
    libs.append((os.path.basename(path), _native.load_library(os.path.abspath(path))))


def med(v):
    return sorted(v)[len(v) // 2]


for n in (64, 16, 8, 4, 1):
    f = bench.smooth_flow(n, 1080, 1920, 8.0, 1000, dev)
    m = bench.hole_mask(n, 1080, 1920, dev)
    routes = {"copy": lambda: utils._flags_to_host(_native.flow_flags(f, m)), "host": lambda: _native.flow_flags_host(f, m)}
    if libs:
        def with_lib(lib):
            def run():
                _native._lib = lib
                return _native.flow_flags_host(f, m)
            return run
        routes = {name: with_lib(lib) for name, lib in libs}
    for fn in routes.values():
        for _ in range(5):
            fn()
    wall = {k: [] for k in routes}
    devt = {k: [] for k in routes}
    for rnd in range(40):
        for k, fn in routes.items():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            fn()
            wall[k].append((time.perf_counter() - t0) * 1e6)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            devt[k].append(e0.elapsed_time(e1) * 1e3)
    tf = []
    for _ in range(40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ofl.Flow(f, 't', m)
        tf.append((time.perf_counter() - t0) * 1e6)
    if libs:
        print("B=%-2d  " % n + "   ".join("%s: wall %.1f us (events %.1f)" % (k, med(wall[k]), med(devt[k])) for k in routes))
        continue
    print("B=%-2d  copy route: wall %.1f us (events %.1f)   host route: wall %.1f us (events %.1f)   Flow(...) %.1f us   [9 B/px at 6.5 TB/s = %.1f us]"
          % (n, med(wall["copy"]), med(devt["copy"]), med(wall["host"]), med(devt["host"]), med(tf), 9 * n * 1080 * 1920 / 6.5e6))
