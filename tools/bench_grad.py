#!/usr/bin/env python3
"""Timing of the backward passes (ofl_warp_bwd_grad_f32, ofl_splat_grad_f32) at B x 1080p: forward + backward of
apply_flow 't' / 's' through torch.autograd, ms per call and algorithmic GB/s of the backward kernel alone."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oflibpytorch_amd as ofl

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n, h, w = a.batch, 1080, 1920
f1, f2, img, m1, m2, tm = bench.make_inputs(n, h, w, dev, 0)
px = n * h * w


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters * 1e3


for ref in 'ts':
    fa = f1.clone().requires_grad_()
    ia = img.clone().requires_grad_()
    wts = torch.randn_like(img)

    def fwd():
        return ofl.apply_flow(fa, ia, ref)

    def both():
        fa.grad = None; ia.grad = None
        (fwd() * wts).sum().backward()
    with torch.no_grad():
        t_f = timeit(lambda: ofl.apply_flow(f1, img, ref))
    t_b = timeit(both)
    print("apply_flow '%s' C=3: forward %.3f ms, forward + backward (grad wrt flow and image, incl. the torch mul / sum) %.3f ms" % (ref, t_f, t_b))


# the flow-level operations: gradients wrt both flows
for name, make in (("switch_ref s->t", lambda a, b: ofl.Flow(a, 's', m1).switch_ref().vecs),
                   ("combine_with mode 3 't'", lambda a, b: ofl.Flow(a, 't', m1).combine_with(ofl.Flow(b, 't', m2), 3).vecs),
                   ("combine_with mode 1 't'", lambda a, b: ofl.Flow(a, 't', m1).combine_with(ofl.Flow(b, 't', m2), 1).vecs)):
    fa, fb = f1.clone().requires_grad_(), f2.clone().requires_grad_()
    wts = torch.randn_like(f1)

    def both():
        fa.grad = None; fb.grad = None
        (make(fa, fb) * wts).sum().backward()
    with torch.no_grad():
        t_f = timeit(lambda: make(f1, f2))
    t_b = timeit(both)
    print("%-26s forward %.3f ms, forward + backward %.3f ms" % (name, t_f, t_b))
