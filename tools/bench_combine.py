#!/usr/bin/env python3
"""Flow.combine over every (mode, self.ref, other.ref, ref) cell at B = 16, 1080 x 1920: the fused plan (one launch after the
optional switch_ref) against the same plan run through the public operators (`flow_class._COMBINE_FUSED = False`: apply, two
scalings, add -- what the reference's chain costs on these kernels), HIP-event time per call, results compared bit for bit.
Also Flow.valid_target / valid_source ('t' / 's' branch: ofl_warp_valid_f32) against ones-image + warp + compare + AND.

    python tools/bench_combine.py [--batch 16] [--iters 10] > profiles/r4_combine_B16.txt
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl
from oflibpytorch_amd import flow_class, utils

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
f1, f2, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)


def timed(fn):
    fn()
    ts = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.iters)
    return sorted(ts)[len(ts) // 2]


print("Flow.combine, B = %d %d x %d, smooth sigma-8 flows with hole masks; ms per call (median of %d x %d)" % (n, h, w, a.rounds, a.iters))
print("%-4s %-5s %-6s %-4s  %9s %9s %7s  %s" % ("mode", "self", "other", "ref", "chain ms", "fused ms", "saved", "same bits"))
tot_c = tot_f = 0.0
for mode in (1, 2, 3):
    for sr in 'st':
        for orf in 'st':
            for ref in 'st':
                A, B = ofl.Flow(f1, sr, m1), ofl.Flow(f2, orf, m2)
                A._flags(); B._flags()
                flow_class._COMBINE_FUSED = False
                rc = A.combine(B, mode, ref)
                tc = timed(lambda: A.combine(B, mode, ref))
                flow_class._COMBINE_FUSED = True
                rf = A.combine(B, mode, ref)
                tf = timed(lambda: A.combine(B, mode, ref))
                # (cells that splat on the sigma-8 flow cross fold tiles: float atomics there, values within tolerance, masks exact)
                same = torch.equal(rc.mask, rf.mask) and (torch.equal(rc.vecs, rf.vecs) or torch.allclose(rc.vecs, rf.vecs, rtol=2e-5, atol=2e-3))
                exact = torch.equal(rc.vecs, rf.vecs)
                tot_c += tc; tot_f += tf
                print("%-4d %-5s %-6s %-4s  %9.3f %9.3f %6.1f%%  %s" % (mode, sr, orf, ref, tc, tf, 100 * (1 - tf / tc),
                                                                        "yes" if exact else ("masks yes, values within 2e-5 (fold tiles)" if same else "NO")))
print("all 24 cells: chain %.2f ms, fused %.2f ms (%.1f %% saved)" % (tot_c, tot_f, 100 * (1 - tot_f / tot_c)))

print()
print("valid_target / valid_source through ofl_warp_valid_f32 (one launch) against ones-image + warp + compare + AND:")
for name, refc, sign in (("valid_target 't'", 't', 1.0), ("valid_source 's'", 's', -1.0)):
    F_ = ofl.Flow(f1, refc, m1)
    F_._flags()

    def unfused():
        ones = torch.ones((n, 1, h, w), device=dev)
        area = utils.apply_flow(F_.vecs * sign, ones, 't').squeeze(1)
        return (area > 0.9999) & F_.mask
    fused = F_.valid_target if refc == 't' else F_.valid_source
    same = torch.equal(unfused(), fused())
    tu, tf = timed(unfused), timed(fused)
    print("%-18s unfused %.3f ms  fused %.3f ms  (%.1f %% saved; %.1f GB/s of 10 B/px)  same bits: %s" % (
        name, tu, tf, 100 * (1 - tf / tu), 10 * n * h * w / tf / 1e6, "yes" if same else "NO"))
