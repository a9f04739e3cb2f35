#!/usr/bin/env python3
"""Randomised cross-checks on the GPU (not part of the pytest tiers; run by hand: python tools/fuzz_gpu.py --seconds 120).

  warp : LDS-staged kernel == generic direct-gather kernel, bit for bit (values, valid mask, flag words), over random
         shapes (any width >= 4, 1-7 channels), flow kinds (smooth, rough, huge, shear, zero), masks, signs, addends,
         rounding modes, batch broadcasts;
  splat: in-order gather path == CPU oracle bit for bit on fold-free flows, == two-pass path within tolerance otherwise,
         identical from run to run, masks always identical;
  API  : Flow.apply(padding=...) through the kernels' flow window == through a padded copy of the flow; flows stored in
         fp16 == their fp32 conversion (switch_ref, invert); track_pts == the oracle's sampler -- all bit for bit; gradients of
         the warp and of the splat == torch's CPU autograd through the restated op sequence, within the stated tolerance.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oflibpytorch_amd import _native
_native.collect_splat_stats = True
from oracle import oracle


def rand_flow(rng, g, n, h, w, dev):
    kind = rng.choice(["smooth", "rough", "huge", "shear", "zero", "const", "stretch"])
    if kind == "stretch":                                   # the sampled region is a scaled copy of the frame: boxes beyond the LDS budget
        xs = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w) - w / 2
        ys = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1) - h / 2
        lo = torch.randn(n, 2, max(h // 12, 2), max(w // 12, 2), generator=g) * rng.uniform(0.2, 4)
        f = torch.nn.functional.interpolate(lo, size=(h, w), mode='bicubic', align_corners=True)
        f = f + torch.cat([-rng.uniform(0, 3) * xs.expand(n, 1, h, w), -rng.uniform(0, 4) * ys.expand(n, 1, h, w)], 1)
    elif kind == "zero":
        f = torch.zeros(n, 2, h, w)
    elif kind == "const":
        f = torch.zeros(n, 2, h, w) + torch.tensor([rng.uniform(-9, 9), rng.uniform(-9, 9)]).view(1, 2, 1, 1)
    else:
        lat = rng.choice([3, 6, 12, 40])
        sig = {"smooth": rng.uniform(0.2, 3), "rough": rng.uniform(3, 20), "huge": rng.uniform(50, 400), "shear": rng.uniform(0.2, 2)}[kind]
        lo = torch.randn(n, 2, max(h // lat, 2), max(w // lat, 2), generator=g) * sig
        f = torch.nn.functional.interpolate(lo, size=(h, w), mode='bicubic', align_corners=True)
        if kind == "shear":
            xs = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w) - w / 2
            ys = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1) - h / 2
            f = f + torch.cat([rng.uniform(-1, 1) * ys.expand(n, 1, h, w), rng.uniform(-8, 8) * xs.expand(n, 1, h, w)], 1)
    return kind, f.contiguous().to(dev)


def extras(rng, g, flow, sm, fm, dev):
    """Flow-level cross-checks of the round-2 paths: the padding WINDOW of the kernels == the padded copy of the flow; flows
    stored in fp16 == their fp32 conversion; track_pts == the oracle's sampler.  All bit for bit."""
    import oflibpytorch_amd as ofl
    n, _, h, w = flow.shape
    done = 0
    ref = str(rng.choice(['s', 't']))
    fl = ofl.Flow(flow, ref, fm)
    # padded apply: target larger than the flow
    pad = [int(v) for v in rng.integers(0, 9, 4)]
    if sum(pad) > 0:
        c = int(rng.integers(1, 4))
        tgt = (torch.rand(n, c, h + pad[0] + pad[1], w + pad[2] + pad[3], generator=g) * 255).to(dev)
        tm = (torch.rand(n, h + pad[0] + pad[1], w + pad[2] + pad[3], generator=g) > 0.2).to(dev)
        cut = bool(rng.random() < 0.5)
        _native._last_splat_stats = None
        a, av = fl.apply(tgt, tm.clone(), return_valid_area=True, padding=pad, cut=cut)
        flp = fl.pad(pad, mode='constant' if ref == 't' else 'replicate')
        b, bv = flp.apply(tgt, tm.clone(), return_valid_area=True)
        if cut:
            b, bv = b[..., pad[0]:pad[0] + h, pad[2]:pad[2] + w], bv[..., pad[0]:pad[0] + h, pad[2]:pad[2] + w]
        st = [0, 0] if _native._last_splat_stats is None else _native._last_splat_stats.cpu().tolist()
        exact = st[0] == 0 and st[1] == 0                    # no tile / image on float atomics (their sums are unordered)
        same = torch.equal(a, b) if exact else bool(torch.allclose(a, b, rtol=5e-5, atol=5e-5 * 255))
        if not (same and torch.equal(av, bv)):
            raise SystemExit("PADDING WINDOW MISMATCH ref %s shape %s pad %s cut %s stats %s" % (ref, tuple(flow.shape), pad, cut, st[:3]))
        done += 1
    # fp16 storage: the kernels read the halves directly
    if w % 4 == 0 and float(flow.abs().max()) < 6e4:
        hf = flow.half()
        x = ofl.Flow(hf, ref, fm)
        y = ofl.Flow(hf.float(), ref, fm)
        for op in (lambda q: q.switch_ref(), lambda q: q.invert()):
            _native._last_splat_stats = None
            p, q = op(x), op(y)
            st = [0, 0] if _native._last_splat_stats is None else _native._last_splat_stats.cpu().tolist()
            exact = st[0] == 0 and st[1] == 0
            same = torch.equal(p.vecs, q.vecs) if exact else bool(torch.allclose(p.vecs, q.vecs, rtol=5e-5, atol=5e-5 * 300))
            if not (same and torch.equal(p.mask, q.mask)):
                raise SystemExit("FP16 STORAGE MISMATCH ref %s shape %s stats %s" % (ref, tuple(flow.shape), st[:3]))
        done += 1
    # gradients against torch's CPU autograd through the restated op sequence (tests/test_autograd.py)
    if h * w <= 40000 and rng.random() < 0.5:
        for sub in ('tests', os.path.join('tests', 'golden')):
            if os.path.join(ROOT, sub) not in sys.path:
                sys.path.insert(0, os.path.join(ROOT, sub))
        import test_autograd as ta
        c = int(rng.integers(1, 4))
        img = torch.rand(1 if rng.random() < 0.3 else n, c, h, w, generator=g)
        wts = torch.randn(n, c, h, w, generator=g)
        fa, ia = flow.clone().requires_grad_(), img.to(dev).requires_grad_()
        (ofl.apply_flow(fa, ia, 't') * wts.to(dev)).sum().backward()
        fb, ib = flow.cpu().clone().requires_grad_(), img.clone().requires_grad_()
        (ta._ref_apply_t(fb, ib) * wts).sum().backward()
        ta._close(fa.grad.cpu(), fb.grad, "fuzz: warp grad wrt flow %s" % (tuple(flow.shape),), rtol=5e-4)
        ta._close(ia.grad.cpu(), ib.grad, "fuzz: warp grad wrt target %s" % (tuple(flow.shape),), rtol=5e-4)
        x = (flow[:, 0] + torch.arange(w, device=dev)[None, None, :]).contiguous()
        y = (flow[:, 1] + torch.arange(h, device=dev)[None, :, None]).contiguous()
        data = torch.rand(n, c, h, w, generator=g)
        xa, ya, da = x.clone().requires_grad_(), y.clone().requires_grad_(), data.to(dev).requires_grad_()
        od, oden = ofl.grid_from_unstructured_data(xa, ya, da, sm)
        wn = torch.randn(n, h, w, generator=g)
        ((od * wts.to(dev)).sum() + (oden * wn.to(dev)).sum()).backward()
        xb, yb, db = x.cpu().clone().requires_grad_(), y.cpu().clone().requires_grad_(), data.clone().requires_grad_()
        rd, rden = ta._ref_splat(xb, yb, db, sm.cpu())
        ((rd * wts).sum() + (rden * wn).sum()).backward()
        ta._close(da.grad.cpu(), db.grad, "fuzz: splat grad wrt data %s" % (tuple(flow.shape),), rtol=1e-3)
        ta._close(xa.grad.cpu(), xb.grad, "fuzz: splat grad wrt x %s" % (tuple(flow.shape),), rtol=2e-3)
        ta._close(ya.grad.cpu(), yb.grad, "fuzz: splat grad wrt y %s" % (tuple(flow.shape),), rtol=2e-3)
        done += 1
    # track_pts against the oracle's sampler
    m = int(rng.integers(1, 40))
    pts = torch.rand(n, m, 2, generator=g) * torch.tensor([h + 3.0, w + 3.0]) - 1.5
    got = ofl.track_pts(flow, 's', pts.to(dev))
    exp = oracle.track_pts(flow.cpu().numpy(), 's', pts.numpy())
    if not np.array_equal(got.cpu().numpy(), np.asarray(exp, dtype=np.float32)):
        raise SystemExit("TRACK_PTS MISMATCH shape %s" % (tuple(flow.shape),))
    return done + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--big", action="store_true", help="frames up to 1100 x 2000 (fewer cases)")
    ap.add_argument("--rows", action="store_true", help="as --column, on launches the ROW-TABLE kernels take (W % 4 == 0, no rounding, no flag by-product; "
                    "plain warps of 1-3 channels, mode 3, and 7-12 channels in the channel loop): auto == the sheared rectangle (path 6) == generic, bit for bit")
    ap.add_argument("--column", action="store_true", help="warp only, batches large enough for the launcher to pick the four-tile column kernel "
                    "(>= 6 912 column groups): auto == generic == two tiles per block == one tile per block, bit for bit")
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(a.seed)
    g = torch.Generator().manual_seed(a.seed)
    t0 = time.time()
    nw = ns = nx = 0
    while time.time() - t0 < a.seconds:
        n, c = int(rng.integers(1, 4)), (int(rng.integers(1, 8)) if rng.random() < 0.8 else int(rng.integers(8, 20)))   # (C >= 4, W % 4 == 0, no addend: the channel-loop kernel)
        h, w = (int(rng.integers(200, 1100)), int(rng.integers(300, 2000))) if a.big else (int(rng.integers(2, 180)), int(rng.integers(4, 300)))
        if a.big:
            n, c = int(rng.integers(1, 3)), int(rng.integers(1, 5))
        if a.column:
            h, w, c = int(rng.integers(2, 200)), int(rng.integers(4, 300)), int(rng.integers(1, 4))
            n = 6912 // (((w + 31) // 32) * ((h + 63) // 64)) + int(rng.integers(1, 9))
        if a.rows:
            a.column = True
            h, w = int(rng.integers(2, 200)), 4 * int(rng.integers(1, 76))
            c = int(rng.integers(1, 4)) if rng.random() < 0.7 else int(rng.integers(7, 13))
            n = (2 if c > 3 else 1) * 6912 // (((w + 31) // 32) * ((h + (15 if c > 3 else 63)) // (16 if c > 3 else 64))) + int(rng.integers(1, 9))
        kind, flow = rand_flow(rng, g, n, h, w, dev)
        src = (torch.rand(n, c, h, w, generator=g) * 300 - 100).to(dev)
        sm = (torch.rand(n, h, w, generator=g) > rng.uniform(0, 0.4)).to(dev)
        fm = (torch.rand(n, h, w, generator=g) > rng.uniform(0, 0.4)).to(dev)
        # ---- warp
        kw = dict(flow_sign=float(rng.choice([1.0, -1.0])))
        if rng.random() < 0.7:
            kw.update(src_mask=sm if rng.random() < 0.7 else None, flow_mask=fm if rng.random() < 0.7 else None, want_valid=True)
        if a.rows:
            if c == 2 and rng.random() < 0.6:                # mode 3 proper (the flow is the addend) / another addend (modes 1-2, Flow.combine)
                kw.update(addend=flow if rng.random() < 0.5 else torch.randn(n, c, h, w, generator=g).to(dev), a_sign=float(rng.choice([1.0, -1.0])), g_sign=float(rng.choice([1.0, -1.0])))
            elif c == 2 and kw.get("want_valid") and rng.random() < 0.4:      # mode 1 't': the staged field is src - src_b
                kw.update(src_b=(torch.rand(n, c, h, w, generator=g) * 50).to(dev))
        elif rng.random() < 0.4:
            kw.update(addend=flow if (c == 2 and rng.random() < 0.5) else torch.randn(n, c, h, w, generator=g).to(dev),
                      a_sign=float(rng.choice([1.0, -1.0])), g_sign=float(rng.choice([1.0, -1.0])))
        elif rng.random() < 0.3:
            kw.update(round_mode=int(rng.integers(1, 3)))
        if rng.random() < 0.3 and not a.rows:
            kw.update(want_flags=True, want_src_flags=(c == 2))
        outs = []
        for path in ((0, 6, 1, 5) if a.rows else (0, 1, 3, 4) if a.column else (0, 1)):
            _native.set_warp_path(path)
            try:
                outs.append(_native.warp_bwd(flow, src, **kw))
            finally:
                _native.set_warp_path(0)
        if c == 2 and kw.get("want_valid") and not kw.get("round_mode") and "src_b" not in kw:
            rf = _native.warp_bwd(flow, src, want_dst_flags=True, **kw)
            assert rf[4].cpu().tolist() == _native.flow_flags(rf[0], rf[1]).cpu().tolist(), "warp dst_flags %s %s" % ((n, c, h, w), kind)
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert (x is None) == (y is None)
                if x is not None and not torch.equal(x, y):
                    bad = (x != y) & ~(torch.isnan(x.float()) & torch.isnan(y.float()))
                    if bad.any():
                        raise SystemExit("WARP MISMATCH shape %s kind %s kw %s: %d values" % ((n, c, h, w), kind, {k: (v if not torch.is_tensor(v) else 'T') for k, v in kw.items()}, int(bad.sum())))
        nw += 1
        if a.column:
            continue
        # ---- splat
        data = src
        skw = dict(weight_mask=sm if rng.random() < 0.6 else None, occlude=bool(rng.random() < 0.6), want_density=True, want_warped=True,
                   flow_sign=float(rng.choice([1.0, -1.0])), data_sign=float(rng.choice([1.0, -1.0])))
        if rng.random() < 0.5:
            skw.update(chan_mask_a=fm, want_mask_chan=True)
        r1 = _native.splat_fwd(flow, data, **skw)
        st = _native._last_splat_stats.cpu().tolist()
        if c == 2 and "chan_mask_a" not in skw:              # the output's flag word as a by-product == a flag pass over it
            rf = _native.splat_fwd(flow, data, want_valid=True, want_dst_flags=True, **{k: v for k, v in skw.items() if k not in ("want_density", "want_warped")})
            assert rf[4].cpu().tolist() == _native.flow_flags(rf[0], rf[1]).cpu().tolist(), "splat dst_flags %s %s" % ((n, c, h, w), kind)
        r2 = _native.splat_fwd(flow, data, **skw)
        _native.set_splat_path(1)
        try:
            r3 = _native.splat_fwd(flow, data, **skw)
        finally:
            _native.set_splat_path(0)
        scale = 300.0
        for x, y, z in zip(r1, r2, r3):
            if x is None:
                continue
            if st[0] == 0 and st[1] == 0:
                assert torch.equal(x, y), "splat not deterministic %s %s" % ((n, c, h, w), kind)
            if x.dtype == torch.bool:
                assert torch.equal(x, z), "splat masks differ between the gather path and two-pass %s %s" % ((n, c, h, w), kind)
            else:
                np.testing.assert_allclose(x.cpu().numpy(), z.cpu().numpy(), rtol=5e-5, atol=5e-5 * scale,
                                           err_msg="gather path vs two-pass %s %s %s" % ((n, c, h, w), kind, st))
        if st[0] == 0 and st[1] == 0 and n * h * w < 40000:      # exact path everywhere: bit-identical to the oracle
            f = flow.cpu().numpy() * np.float32(skw["flow_sign"])
            d = data.cpu().numpy() * np.float32(skw["data_sign"])
            mc = skw["chan_mask_a"].cpu().numpy() if "chan_mask_a" in skw else np.ones((n, h, w), bool)
            dd = np.concatenate([d, mc[:, None].astype(np.float32)], 1)
            m = skw["weight_mask"].cpu().numpy() if skw.get("weight_mask") is not None else None
            ref, rwarped, rden = oracle.apply_s_flow(f, dd, m, skw["occlude"], return_density=True)
            if not np.array_equal(r1[0].cpu().numpy(), ref[:, :c]):
                raise SystemExit("SPLAT NOT EXACT shape %s kind %s" % ((n, c, h, w), kind))
            assert np.array_equal(r1[2].cpu().numpy(), rden) and np.array_equal(r1[3].cpu().numpy(), rwarped)
            if "chan_mask_a" in skw:
                assert np.array_equal(r1[1].cpu().numpy(), ref[:, c])
        ns += 1
        # ---- the rows either side of the path (API level)
        if kind in ("smooth", "rough", "shear", "const") and h >= 8 and w >= 8 and rng.random() < 0.35:
            nx += extras(rng, g, flow, sm, fm, dev)
    print("fuzz ok: %d warp cases, %d splat cases, %d API-level cases (padding window / fp16 storage / track_pts) in %.0f s"
          % (nw, ns, nx, time.time() - t0))


if __name__ == "__main__":
    main()
