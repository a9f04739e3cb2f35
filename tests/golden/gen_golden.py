#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE (oflibpytorch v2.1.1,
/root/reference, PyTorch CPU) on small deterministic inputs.

Runs only in the build container (where /root/reference exists).  Nothing of the reference
travels: the fixtures hold inputs and expected outputs only.  `cv2` is not installed and nothing
on the hot path calls it, so an empty stub module stands in for the import.

    python tests/golden/gen_golden.py

writes  tests/golden/{prims,flow_apply,flow_ops,kats}.npz  and  tests/golden/manifest.json.
Each case is  {"id", "op", "args" (json), "in": {name: key}, "out": {name: key}}  where key names an
array in the group's npz (content-addressed, so inputs shared between cases are stored once).
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault('cv2', types.ModuleType('cv2'))
sys.path.insert(0, '/root/reference/src')
sys.path.insert(0, HERE)
import codec  # noqa: E402
import oflibpytorch as of  # noqa: E402
from oflibpytorch import utils as ofu  # noqa: E402

Flow = of.Flow
torch.manual_seed(0)

GROUPS = {}
MANIFEST = []


def rec(group, cid, op, args, inputs, outputs):
    store = GROUPS.setdefault(group, {})
    cid = group + "." + cid
    entry = {"id": cid, "group": group, "op": op, "args": args, "in": {}, "out": {}}
    for kind, d in (("in", inputs), ("out", outputs)):
        for k, v in d.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            v = np.ascontiguousarray(v)
            key = "a" + hashlib.sha1(v.tobytes() + str((v.dtype, v.shape)).encode()).hexdigest()[:16]
            store[key] = v              # identical arrays (shared inputs) are stored once
            entry[kind][k] = key
    assert cid not in [m["id"] for m in MANIFEST], cid
    MANIFEST.append(entry)


# ------------------------------------------------------------------------------------------------
# synthetic inputs
# ------------------------------------------------------------------------------------------------
def smooth_flow(n, h, w, sigma, seed, cell=12):
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(n, 2, max(h // cell, 2), max(w // cell, 2), generator=g) * sigma
    return F.interpolate(lo, size=(h, w), mode='bicubic', align_corners=True).contiguous()


def hole_mask(n, h, w, seed):
    g = np.random.RandomState(seed)
    m = np.ones((n, h, w), bool)
    for i in range(n):
        y0, x0 = g.randint(0, h - 5), g.randint(0, w - 6)
        m[i, y0:y0 + g.randint(2, 6), x0:x0 + g.randint(3, 8)] = False
        m[i, g.randint(0, h - 2):, :g.randint(1, 5)] = False
    return torch.tensor(m)


def image(n, c, h, w, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, c, h, w, generator=g) * 255
    if dtype == torch.uint8:
        return img.round().to(torch.uint8)
    return img.to(dtype)


H, W = 30, 44          # numerics-heavy cases
HS, WS = 12, 16        # plumbing cases (broadcasts, ranks, dtypes, kwargs)


# ------------------------------------------------------------------------------------------------
# group: prims  (utils.py functions)
# ------------------------------------------------------------------------------------------------
def gen_prims():
    g = 'prims'
    coords = (torch.rand(3, 17, 2, generator=torch.Generator().manual_seed(1)) - 0.2) * torch.tensor([W * 1.3, H * 1.3])
    rec(g, 'norm_coords', 'normalise_coords', {"shape": [H, W]}, {"coords": coords},
        {"out": ofu.normalise_coords(coords, (H, W))})

    f = smooth_flow(2, H, W, 5.0, 11)
    for ref in 'st':
        x, y = ofu.get_flow_endpoints(f, ref)
        rec(g, 'endpoints_' + ref, 'get_flow_endpoints', {"ref": ref}, {"flow": f}, {"x": x, "y": y})

    ft = f.clone() * 1e-3
    ft[0, :, :10] = 0
    ft[1, 0, 5, 5] = 1e-3
    ft[1, 1, 6, 6] = -1e-3
    ft[1, 1, 7, 7] = 9.99e-4
    rec(g, 'threshold', 'threshold_vectors', {}, {"flow": ft}, {"out": ofu.threshold_vectors(ft)})
    fz = torch.zeros(3, 2, 8, 9)
    fz[1, 0, 2, 2] = 5e-4
    fz[2, 1, 3, 3] = 2e-3
    for th in (True, False):
        rec(g, 'is_zero_flow_%d' % th, 'is_zero_flow', {"thresholded": th}, {"flow": fz},
            {"out": ofu.is_zero_flow(fz, th)})

    # apply_flow 't': in-range and out-of-range samples
    f = smooth_flow(3, H, W, 6.0, 12)
    f[2] += torch.tensor([25.0, -17.0]).view(2, 1, 1)      # pushes samples out of the image
    img = image(3, 3, H, W, 21)
    rec(g, 'apply_t_n3', 'apply_flow', {"ref": "t"}, {"flow": f, "target": img},
        {"out": ofu.apply_flow(f, img, 't')})
    # plumbing: broadcasts, 2-D / 3-D targets, H-W-2 flows, dtypes (utils.py:510-537, 602-618)
    f = smooth_flow(3, HS, WS, 3.0, 12)
    img = image(3, 3, HS, WS, 21)
    rec(g, 'apply_t_bcast_target', 'apply_flow', {"ref": "t"}, {"flow": f, "target": img[:1]},
        {"out": ofu.apply_flow(f, img[:1], 't')})
    rec(g, 'apply_t_bcast_flow', 'apply_flow', {"ref": "t"}, {"flow": f[:1], "target": img},
        {"out": ofu.apply_flow(f[:1], img, 't')})
    rec(g, 'apply_t_2d', 'apply_flow', {"ref": "t"}, {"flow": f[0], "target": img[0, 0]},
        {"out": ofu.apply_flow(f[0], img[0, 0], 't')})
    rec(g, 'apply_t_2d_n3', 'apply_flow', {"ref": "t"}, {"flow": f, "target": img[0, 0]},
        {"out": ofu.apply_flow(f, img[0, 0], 't')})
    rec(g, 'apply_t_3d', 'apply_flow', {"ref": "t"}, {"flow": f[1], "target": img[1]},
        {"out": ofu.apply_flow(f[1], img[1], 't')})
    rec(g, 'apply_t_hw2', 'apply_flow', {"ref": "t"}, {"flow": f[1].permute(1, 2, 0).contiguous(), "target": img[1]},
        {"out": ofu.apply_flow(f[1].permute(1, 2, 0).contiguous(), img[1], 't')})
    u8 = image(3, 3, HS, WS, 22, torch.uint8)
    rec(g, 'apply_t_u8', 'apply_flow', {"ref": "t"}, {"flow": f, "target": u8},
        {"out": ofu.apply_flow(f, u8, 't')})
    i32 = (image(3, 1, HS, WS, 23) * 3 - 300).round().to(torch.int32)
    rec(g, 'apply_t_i32', 'apply_flow', {"ref": "t"}, {"flow": f, "target": i32},
        {"out": ofu.apply_flow(f, i32, 't')})
    f64 = image(2, 2, HS, WS, 24).double()
    rec(g, 'apply_t_f64', 'apply_flow', {"ref": "t"}, {"flow": f[:2], "target": f64},
        {"out": ofu.apply_flow(f[:2], f64, 't')})
    f = smooth_flow(3, H, W, 6.0, 12)
    img = image(3, 3, H, W, 21)
    # integer translation (exact shift; test_flow_class.py:1080-1091 pins equality)
    tr = torch.zeros(2, 2, H, W)
    tr[0, 0], tr[0, 1] = 10, -7
    tr[1, 0], tr[1, 1] = -3, 4
    rec(g, 'apply_t_translation', 'apply_flow', {"ref": "t"}, {"flow": tr, "target": img[:2]},
        {"out": ofu.apply_flow(tr, img[:2], 't')})
    rec(g, 'apply_s_translation', 'apply_flow', {"ref": "s"}, {"flow": tr, "target": img[:2]},
        {"out": ofu.apply_flow(tr, img[:2], 's')})
    # early exit: every |v| < 1e-3 -> the target object itself is returned (utils.py:497-498)
    tiny = torch.full((1, 2, H, W), 5e-4)
    out = ofu.apply_flow(tiny, img[:1], 't')
    rec(g, 'apply_t_tiny', 'apply_flow', {"ref": "t", "same_object": bool(out is img[:1]) or out.data_ptr() == img.data_ptr()},
        {"flow": tiny, "target": img[:1]}, {"out": out})
    # exactly at threshold: 1e-3 is NOT below threshold -> warps
    edge = torch.full((1, 2, H, W), 1e-3)
    rec(g, 'apply_t_thr_edge', 'apply_flow', {"ref": "t"}, {"flow": edge, "target": img[:1]},
        {"out": ofu.apply_flow(edge, img[:1], 't')})
    # zero flow on a non-square odd size: goes through the early exit
    # unit-size corner cases: W == 2, H == 2
    f22 = torch.tensor([[[[0.3, -0.4], [0.6, 0.2]], [[-0.2, 0.5], [0.1, -0.7]]]])
    t22 = torch.tensor([[[[1.0, 2.0], [3.0, 4.0]]]])
    rec(g, 'apply_t_2x2', 'apply_flow', {"ref": "t"}, {"flow": f22, "target": t22},
        {"out": ofu.apply_flow(f22, t22, 't')})
    rec(g, 'apply_s_2x2', 'apply_flow', {"ref": "s"}, {"flow": f22, "target": t22},
        {"out": ofu.apply_flow(f22, t22, 's')})

    # apply_flow 's' with / without mask, broadcasts
    fs = smooth_flow(3, H, W, 4.0, 13)
    fs[0, :, 10:20, 10:30] = 0                 # zero-flow block (occlusion rule)
    fs[1, :, :, :8] = 2e-4                     # below threshold counts as zero
    m = hole_mask(3, H, W, 5)
    rec(g, 'apply_s_n3', 'apply_flow', {"ref": "s"}, {"flow": fs, "target": img},
        {"out": ofu.apply_flow(fs, img, 's')})
    rec(g, 'apply_s_n3_mask', 'apply_flow', {"ref": "s"}, {"flow": fs, "target": img, "mask": m},
        {"out": ofu.apply_flow(fs, img, 's', m)})
    fss = smooth_flow(3, HS, WS, 2.5, 14)
    fss[0, :, 3:7, 4:10] = 0
    ms, imgs = hole_mask(3, HS, WS, 6), image(3, 3, HS, WS, 21)
    rec(g, 'apply_s_bcast_flow', 'apply_flow', {"ref": "s"}, {"flow": fss[:1], "target": imgs, "mask": ms[:1]},
        {"out": ofu.apply_flow(fss[:1], imgs, 's', ms[:1])})
    rec(g, 'apply_s_bcast_target', 'apply_flow', {"ref": "s"}, {"flow": fss, "target": imgs[1:2], "mask": ms},
        {"out": ofu.apply_flow(fss, imgs[1:2], 's', ms)})
    rec(g, 'apply_s_u8', 'apply_flow', {"ref": "s"}, {"flow": fss, "target": u8, "mask": ms},
        {"out": ofu.apply_flow(fss, u8, 's', ms)})
    rec(g, 'apply_s_2d', 'apply_flow', {"ref": "s"}, {"flow": fss[0], "target": imgs[0, 0], "mask": ms[0]},
        {"out": ofu.apply_flow(fss[0], imgs[0, 0], 's', ms[0])})

    # grid_from_unstructured_data on arbitrary positions (some outside the image)
    gen = torch.Generator().manual_seed(31)
    x = torch.rand(2, H, W, generator=gen) * (W + 6) - 3
    y = torch.rand(2, H, W, generator=gen) * (H + 6) - 3
    x[0, 0, :5] = torch.tensor([0.0, W - 1.0, -1.0, W + 0.0, 3.0])      # exact-integer / edge positions
    y[0, 0, :5] = torch.tensor([0.0, H - 1.0, 2.0, 5.0, -1.0])
    d = image(2, 3, H, W, 32)
    gd, den = ofu.grid_from_unstructured_data(x, y, d)
    rec(g, 'gfud_random', 'grid_from_unstructured_data', {}, {"x": x, "y": y, "data": d}, {"data": gd, "density": den})
    gd, den = ofu.grid_from_unstructured_data(x, y, d, m[:2])
    rec(g, 'gfud_random_mask', 'grid_from_unstructured_data', {}, {"x": x, "y": y, "data": d, "mask": m[:2]},
        {"data": gd, "density": den})
    xs, ys = ofu.get_flow_endpoints(fs, 's')
    gd, den = ofu.grid_from_unstructured_data(xs, ys, fs)
    rec(g, 'gfud_flow', 'grid_from_unstructured_data', {}, {"x": xs, "y": ys, "data": fs}, {"data": gd, "density": den})

    # apply_s_flow
    for occ in (True, False):
        wd, wm = ofu.apply_s_flow(fs, img, m, occ)
        rec(g, 'apply_s_flow_occ%d' % occ, 'apply_s_flow', {"occlude_zero_flow": occ},
            {"flow": fs, "data": img, "mask": m}, {"data": wd, "mask": wm})
    wd, wm = ofu.apply_s_flow(fs, img)
    rec(g, 'apply_s_flow_nomask', 'apply_s_flow', {"occlude_zero_flow": None}, {"flow": fs, "data": img},
        {"data": wd, "mask": wm})


# ------------------------------------------------------------------------------------------------
# group: flow_apply  (Flow.apply, flow_class.py:755-959)
# ------------------------------------------------------------------------------------------------
def flow_inputs(fl):
    return {"f": fl.vecs, "m": fl.mask}


def _flow_apply_cases(g, ref, h, w, tag, names):
    img = image(3, 3, h, w, 41)
    u8 = image(3, 3, h, w, 42, torch.uint8)
    tmask = hole_mask(3, h, w, 9)
    f = smooth_flow(3, h, w, 5.0 if h > 20 else 2.5, 50 + ord(ref))
    f[1, :, h // 6:h // 2, w // 3:w - 4] = 0
    f[2] += torch.tensor([0.27 * w, 0.3 * h]).view(2, 1, 1)
    m = hole_mask(3, h, w, 7)
    fl = Flow(f, ref, m)
    fl_nomask = Flow(f, ref)
    # tensor target, kwargs combinations (flow_class.py:821-959)
    for name, flow, kw, tgt in [
        ('plain', fl, {}, img),
        ('valid', fl, {"return_valid_area": True}, img),
        ('valid_tmask', fl, {"return_valid_area": True, "target_mask": tmask}, img),
        ('valid_tmask_nocons', fl, {"return_valid_area": True, "target_mask": tmask, "consider_mask": False}, img),
        ('nocons', fl, {"consider_mask": False}, img),
        ('nomask_valid', fl_nomask, {"return_valid_area": True}, img),
        ('u8_valid', fl, {"return_valid_area": True}, u8),
        ('u8_plain', fl, {}, u8),
        ('bcast_target', fl, {"return_valid_area": True, "target_mask": tmask[:1]}, img[:1]),
        ('bcast_flow', Flow(f[1:2], ref, m[1:2]), {"return_valid_area": True, "target_mask": tmask}, img),
        ('t3d', Flow(f[0], ref, m[0]), {"return_valid_area": True, "target_mask": tmask[0]}, img[0]),
        ('t2d', Flow(f[0], ref, m[0]), {"return_valid_area": True}, img[0, 0]),
        ('t2d_n3', fl, {"return_valid_area": True}, img[0, 0]),
        ('t3d_n3', fl, {}, img[0]),
    ]:
        if name not in names:
            continue
        kwj = {k: v for k, v in kw.items() if k != "target_mask"}
        tm = kw.get("target_mask")
        tm_in = None if tm is None else tm.clone()
        out = flow.apply(tgt, **{**kwj, **({"target_mask": tm_in} if tm is not None else {})})
        outs = {"warped": out[0], "valid": out[1]} if isinstance(out, tuple) else {"warped": out}
        rec(g, 'apply_%s_%s%s' % (ref, name, tag), 'Flow.apply', {"ref": ref, "kwargs": kwj},
            {**flow_inputs(flow), "target": tgt, "target_mask": tm}, outs)
    # Flow target (flow_class.py:839-842, 938)
    if 'flowtarget' in names:
        tf = Flow(smooth_flow(3, h, w, 3.0, 77), 't' if ref == 's' else 's', hole_mask(3, h, w, 13))
        out = fl.apply(tf)
        rec(g, 'apply_%s_flowtarget%s' % (ref, tag), 'Flow.apply',
            {"ref": ref, "kwargs": {}, "target_ref": tf.ref, "out_ref": out.ref},
            {**flow_inputs(fl), "tf": tf.vecs, "tm": tf.mask}, {"vecs": out.vecs, "mask": out.mask})
    if 'flowtarget_bcast' in names:
        tf = Flow(smooth_flow(3, h, w, 3.0, 77), 't' if ref == 's' else 's', hole_mask(3, h, w, 13))
        out = Flow(f[:1], ref, m[:1]).apply(tf)
        rec(g, 'apply_%s_flowtarget_bcast%s' % (ref, tag), 'Flow.apply',
            {"ref": ref, "kwargs": {}, "target_ref": tf.ref, "out_ref": out.ref},
            {"f": f[:1], "m": m[:1], "tf": tf.vecs, "tm": tf.mask}, {"vecs": out.vecs, "mask": out.mask})
    # padding: flow smaller than target (flow_class.py:830-834, 880-893, 906-934)
    if 'pad' in names:
        pad = [3, 5, 4, 2]
        hf, wf = h - pad[0] - pad[1], w - pad[2] - pad[3]
        fp = smooth_flow(2, hf, wf, 4.0, 60 + ord(ref))
        mp = hole_mask(2, hf, wf, 17)
        flp = Flow(fp, ref, mp)
        for cut in (True, False):
            tmc = tmask[:2].clone()
            out = flp.apply(img[:2], target_mask=tmc, return_valid_area=True, padding=pad, cut=cut)
            rec(g, 'apply_%s_pad_cut%d%s' % (ref, cut, tag), 'Flow.apply',
                {"ref": ref, "kwargs": {"return_valid_area": True, "padding": pad, "cut": cut}},
                {"f": fp, "m": mp, "target": img[:2], "target_mask": tmask[:2]}, {"warped": out[0], "valid": out[1]})
            out = flp.apply(img[:2], padding=pad, cut=cut)
            rec(g, 'apply_%s_pad_plain_cut%d%s' % (ref, cut, tag), 'Flow.apply',
                {"ref": ref, "kwargs": {"padding": pad, "cut": cut}},
                {"f": fp, "m": mp, "target": img[:2]}, {"warped": out})
        tfl = Flow(smooth_flow(2, h, w, 3.0, 78), ref, hole_mask(2, h, w, 14))
        out = flp.apply(tfl, padding=pad, cut=False)
        rec(g, 'apply_%s_pad_flowtarget%s' % (ref, tag), 'Flow.apply',
            {"ref": ref, "kwargs": {"padding": pad, "cut": False}, "target_ref": ref, "out_ref": out.ref},
            {"f": fp, "m": mp, "tf": tfl.vecs, "tm": tfl.mask}, {"vecs": out.vecs, "mask": out.mask})
    # zero flow: early exit path inside apply_flow (utils.py:497-498)
    if 'zero' in names:
        z = Flow(torch.zeros(2, 2, h, w), ref, m[:2])
        out = z.apply(img[:2], target_mask=tmask[:2].clone(), return_valid_area=True)
        rec(g, 'apply_%s_zero%s' % (ref, tag), 'Flow.apply', {"ref": ref, "kwargs": {"return_valid_area": True}},
            {"f": z.vecs, "m": z.mask, "target": img[:2], "target_mask": tmask[:2]}, {"warped": out[0], "valid": out[1]})


def gen_flow_apply():
    g = 'flow_apply'
    every = ['plain', 'valid', 'valid_tmask', 'valid_tmask_nocons', 'nocons', 'nomask_valid', 'u8_valid', 'u8_plain',
             'bcast_target', 'bcast_flow', 't3d', 't2d', 't2d_n3', 't3d_n3', 'flowtarget', 'flowtarget_bcast', 'zero']
    for ref in 'st':
        _flow_apply_cases(g, ref, H, W, '', ['valid_tmask', 'valid_tmask_nocons', 'flowtarget', 'pad'])
        _flow_apply_cases(g, ref, HS, WS, '_small', every)


# ------------------------------------------------------------------------------------------------
# group: flow_ops  (switch_ref, invert, combine_with, wrappers, arithmetic)
# ------------------------------------------------------------------------------------------------
def _flow_ops_cases(g, ref, h, w, tag, numerics, plumbing):
    f = smooth_flow(2, h, w, 4.0 if h > 20 else 2.0, 80 + ord(ref))
    f[1, :, h // 4:h // 2, w // 4:w // 2] = 0
    m = hole_mask(2, h, w, 19)
    fl = Flow(f, ref, m)
    f2 = smooth_flow(2, h, w, 3.0 if h > 20 else 1.5, 90 + ord(ref))
    m2 = hole_mask(2, h, w, 23)
    fl2 = Flow(f2, ref, m2)
    if numerics:
        out = fl.switch_ref()
        rec(g, 'switch_ref_' + ref + tag, 'Flow.switch_ref', {"ref": ref, "out_ref": out.ref}, flow_inputs(fl),
            {"vecs": out.vecs, "mask": out.mask})
        for oref in ('s', 't', None):
            out = fl.invert(oref)
            rec(g, 'invert_%s_%s%s' % (ref, oref, tag), 'Flow.invert', {"ref": ref, "arg_ref": oref, "out_ref": out.ref},
                flow_inputs(fl), {"vecs": out.vecs, "mask": out.mask})
        for mode in (1, 2, 3):
            out = fl.combine_with(fl2, mode)
            rec(g, 'combine_%s_mode%d%s' % (ref, mode, tag), 'Flow.combine_with',
                {"ref": ref, "mode": mode, "out_ref": out.ref},
                {"f1": f, "m1": m, "f2": f2, "m2": m2}, {"vecs": out.vecs, "mask": out.mask})
            out = of.combine_flows(f, f2, mode, ref)
            rec(g, 'combine_flows_%s_mode%d%s' % (ref, mode, tag), 'combine_flows', {"ref": ref, "mode": mode},
                {"f1": f, "f2": f2}, {"vecs": out})
        for cm in (True, False):
            rec(g, 'valid_target_%s_%d%s' % (ref, cm, tag), 'Flow.valid_target', {"ref": ref, "consider_mask": cm},
                flow_inputs(fl), {"out": fl.valid_target(cm)})
            rec(g, 'valid_source_%s_%d%s' % (ref, cm, tag), 'Flow.valid_source', {"ref": ref, "consider_mask": cm},
                flow_inputs(fl), {"out": fl.valid_source(cm)})
    if not plumbing:
        return
    out = fl.switch_ref('invalid')
    rec(g, 'switch_ref_invalid_' + ref + tag, 'Flow.switch_ref', {"ref": ref, "mode": "invalid", "out_ref": out.ref},
        flow_inputs(fl), {"vecs": out.vecs, "mask": out.mask})
    for th, ma in [(None, None), (False, True), (True, False), (False, False)]:
        rec(g, 'is_zero_%s_%s_%s%s' % (ref, th, ma, tag), 'Flow.is_zero', {"ref": ref, "thresholded": th, "masked": ma},
            flow_inputs(fl), {"out": fl.is_zero(th, ma)})
    out = of.combine_flows(f[0], f2[0], 3, ref)
    rec(g, 'combine_flows_%s_3d%s' % (ref, tag), 'combine_flows', {"ref": ref, "mode": 3}, {"f1": f[0], "f2": f2[0]},
        {"vecs": out})
    a1, a2 = f[0].permute(1, 2, 0).numpy().copy(), f2[0].permute(1, 2, 0).numpy().copy()
    out = of.combine_flows(a1, a2, 3, ref)
    rec(g, 'combine_flows_%s_hw2_numpy%s' % (ref, tag), 'combine_flows', {"ref": ref, "mode": 3, "numpy": True},
        {"f1": a1, "f2": a2}, {"vecs": out})
    # early exits of combine_with (flow_class.py:1729-1744)
    zero = Flow(torch.zeros(2, 2, h, w), ref, m)
    for mode in (1, 2, 3):
        out = zero.combine_with(fl2, mode)
        rec(g, 'combine_%s_selfzero_mode%d%s' % (ref, mode, tag), 'Flow.combine_with',
            {"ref": ref, "mode": mode, "returns": "flow" if out is fl2 else "new", "out_ref": out.ref},
            {"f1": zero.vecs, "m1": zero.mask, "f2": f2, "m2": m2}, {"vecs": out.vecs, "mask": out.mask})
        out = fl.combine_with(zero, mode)
        rec(g, 'combine_%s_flowzero_mode%d%s' % (ref, mode, tag), 'Flow.combine_with',
            {"ref": ref, "mode": mode, "returns": "self" if out is fl else "new", "out_ref": out.ref},
            {"f1": f, "m1": m, "f2": zero.vecs, "m2": zero.mask}, {"vecs": out.vecs, "mask": out.mask})
    # masked-zero: non-zero vectors only where the mask is False -> is_zero() is True (:1241-1244)
    fz = torch.zeros(2, 2, h, w)
    mz = torch.ones(2, h, w, dtype=torch.bool)
    fz[:, :, :4, :4] = 3.0
    mz[:, :4, :4] = False
    flz = Flow(fz, ref, mz)
    out = flz.combine_with(fl2, 3)
    rec(g, 'combine_%s_maskedzero%s' % (ref, tag), 'Flow.combine_with',
        {"ref": ref, "mode": 3, "returns": "flow" if out is fl2 else "new", "out_ref": out.ref},
        {"f1": fz, "m1": mz, "f2": f2, "m2": m2}, {"vecs": out.vecs, "mask": out.mask})
    # tiny vectors: thresholded=True takes the early exit, False warps ... through apply_flow's own
    # thresholded early exit (utils.py:497), i.e. G is the identity
    tiny = Flow(torch.full((2, 2, h, w), 4e-4), ref, m)
    for th in (False, True):
        out = tiny.combine_with(fl2, 3, thresholded=th)
        rec(g, 'combine_%s_tiny_th%d%s' % (ref, th, tag), 'Flow.combine_with',
            {"ref": ref, "mode": 3, "thresholded": th, "returns": "flow" if out is fl2 else "new", "out_ref": out.ref},
            {"f1": tiny.vecs, "m1": tiny.mask, "f2": f2, "m2": m2}, {"vecs": out.vecs, "mask": out.mask})
        out = fl2.combine_with(tiny, 3, thresholded=th)
        rec(g, 'combine_%s_tiny2_th%d%s' % (ref, th, tag), 'Flow.combine_with',
            {"ref": ref, "mode": 3, "thresholded": th, "returns": "self" if out is fl2 else "new", "out_ref": out.ref},
            {"f1": f2, "m1": m2, "f2": tiny.vecs, "m2": tiny.mask}, {"vecs": out.vecs, "mask": out.mask})
    # wrappers (flow_operations.py:191-228)
    out = of.switch_flow_ref(f, ref)
    rec(g, 'switch_flow_ref_' + ref + tag, 'switch_flow_ref', {"ref": ref}, {"f": f}, {"vecs": out})
    out = of.switch_flow_ref(f[0], ref)
    rec(g, 'switch_flow_ref_3d_' + ref + tag, 'switch_flow_ref', {"ref": ref}, {"f": f[0]}, {"vecs": out})
    for oref in ('s', 't', None):
        out = of.invert_flow(f, ref, oref)
        rec(g, 'invert_flow_%s_%s%s' % (ref, oref, tag), 'invert_flow', {"ref": ref, "out_ref": oref}, {"f": f}, {"vecs": out})
    # arithmetic (flow_class.py:450-549, 680-692)
    out = fl + fl2
    rec(g, 'add_' + ref + tag, 'Flow.add', {"ref": ref}, {"f1": f, "m1": m, "f2": f2, "m2": m2}, {"vecs": out.vecs, "mask": out.mask})
    out = fl - fl2
    rec(g, 'sub_' + ref + tag, 'Flow.sub', {"ref": ref}, {"f1": f, "m1": m, "f2": f2, "m2": m2}, {"vecs": out.vecs, "mask": out.mask})
    out = fl + f2
    rec(g, 'add_tensor_' + ref + tag, 'Flow.add', {"ref": ref, "tensor": True}, {"f1": f, "m1": m, "f2": f2}, {"vecs": out.vecs, "mask": out.mask})
    rec(g, 'add_bcast_' + ref + tag, 'Flow.add', {"ref": ref}, {"f1": f, "m1": m, "f2": f2[:1], "m2": m2[:1]},
        {"vecs": (fl + Flow(f2[:1], ref, m2[:1])).vecs, "mask": (fl + Flow(f2[:1], ref, m2[:1])).mask})
    out = -fl
    rec(g, 'neg_' + ref + tag, 'Flow.neg', {"ref": ref}, {"f1": f, "m1": m}, {"vecs": out.vecs, "mask": out.mask})
    out = fl * 2.5
    rec(g, 'mul_' + ref + tag, 'Flow.mul', {"ref": ref, "scalar": 2.5}, {"f1": f, "m1": m}, {"vecs": out.vecs, "mask": out.mask})


def gen_flow_ops():
    g = 'flow_ops'
    for ref in 'st':
        _flow_ops_cases(g, ref, H, W, '', True, False)
        _flow_ops_cases(g, ref, HS, WS, '_small', True, True)


# ------------------------------------------------------------------------------------------------
# group: kats  (the reference's own known-answer tests + BASELINE.md config-1 values)
# ------------------------------------------------------------------------------------------------
def _affine_inputs(prefix, vecs):
    """Encode a 1-2-H-W affine flow field compactly and exactly (tests/golden/codec.py)."""
    params, delta, esc = codec.encode_affine(vecs[0].numpy())
    return {prefix + "__params": params, prefix + "__delta": delta, prefix + "__esc": esc}


def _summary(t):
    """Compact expectation for a big fp32 output: exact values on a stride-4 lattice + float64 sums."""
    a = t.detach().numpy()
    return {"sub4": codec.subsample(a, 4)}, {"sum": float(a.astype(np.float64).sum()),
                                             "abs_sum": float(np.abs(a.astype(np.float64)).sum())}


def gen_kats():
    g = 'kats'
    # --- test_utils.py:1104-1120 TestGridFromUnstructuredData: count_nonzero(density > 0.95) == 24664 for a
    #     batch of two identical 100x150 rotation flows (stored once: 12332 per batch element)
    flow = Flow.from_transforms([['rotation', 50, 75, -20]], (100, 150), ref='s')
    flow2 = of.batch_flows((flow, flow))
    x, y = ofu.get_flow_endpoints(flow2.vecs, 's')
    data, den = ofu.grid_from_unstructured_data(x, y, flow2.vecs)
    cnt = int(np.count_nonzero(den.numpy() > 0.95))
    assert cnt == 24664, cnt
    assert torch.equal(data[0], data[1]) and torch.equal(den[0], den[1])
    sub_d, sums_d = _summary(data[:1])
    sub_n, sums_n = _summary(den[:1])
    rec(g, 'gfud_rotation', 'kat_gfud', {"transforms": [['rotation', 50, 75, -20]], "shape": [100, 150],
                                         "count_batch_of_two": cnt, "data": sums_d, "density": sums_n},
        _affine_inputs("flow", flow.vecs),
        {"density_gt_095_packed": np.packbits(den[0].numpy() > 0.95), "data_sub4": sub_d["sub4"],
         "density_sub4": sub_n["sub4"]})

    # --- test_flow_class.py:1389-1617 valid_target / valid_source 7x7 boolean KATs (pure-pytorch variants).
    #     Expected arrays are produced by the reference here; the literal `desired_area_s_pp` of the reference's
    #     test is asserted against as a cross-check.
    transforms = [['rotation', 0, 0, 45]]
    shape = (7, 7)
    mask_s = np.ones(shape, 'bool'); mask_s[4:, :3] = False
    mask_t = np.ones(shape, 'bool'); mask_t[:3, 4:] = False
    f_s_m = Flow.from_transforms(transforms, shape, 's', mask_s)
    f_t_m = Flow.from_transforms(transforms, shape, 't', mask_t)
    f_s = Flow.from_transforms(transforms, shape, 's')
    f_t = Flow.from_transforms(transforms, shape, 't')
    desired_area_s_pp = np.array([
        [1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1, 1],
        [0, 0, 0, 1, 1, 1, 0], [0, 0, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0, 0]]).astype('bool')
    assert np.array_equal(f_s.valid_target()[0].numpy(), desired_area_s_pp)
    for name, fl in (('s', f_s), ('t', f_t), ('s_masked', f_s_m), ('t_masked', f_t_m)):
        for cm in (True, False):
            rec(g, 'valid_target_%s_%d' % (name, cm), 'Flow.valid_target', {"ref": fl.ref, "consider_mask": cm},
                flow_inputs(fl), {"out": fl.valid_target(cm)})
            rec(g, 'valid_source_%s_%d' % (name, cm), 'Flow.valid_source', {"ref": fl.ref, "consider_mask": cm},
                flow_inputs(fl), {"out": fl.valid_source(cm)})

    # --- BASELINE.md / SURVEY.md 8(c): config-1 values on smudge.png[:300, :400] (BGR, C-H-W)
    from PIL import Image
    img = np.array(Image.open('/root/reference/test/smudge.png'))[..., ::-1]   # == cv2.imread (BGR)
    img = np.ascontiguousarray(np.moveaxis(img[:300, :400], -1, 0))           # 3-300-400 uint8
    timg = torch.tensor(img).float()
    assert float(timg.sum()) == 43891908.0
    fields = {
        "rot_t": Flow.from_transforms([['rotation', 200, 150, -30]], (300, 400), 't'),
        "rot_s": Flow.from_transforms([['rotation', 200, 150, -30]], (300, 400), 's'),
        "sc_t": Flow.from_transforms([['scaling', 100, 50, 0.7]], (300, 400), 't'),
        "sc_s": Flow.from_transforms([['scaling', 100, 50, 0.7]], (300, 400), 's'),
        "tr_t": Flow.from_transforms([['translation', 40, 0]], (300, 400), 't'),
    }
    inputs = {"img_u8": img}
    for k, fl in fields.items():
        inputs.update(_affine_inputs(k, fl.vecs))
    rec(g, 'cfg1_inputs', 'kat_inputs', {"fields": sorted(fields)}, inputs, {})

    def kat(cid, call, ref, vec_t, mask_t, extra):
        sub, sums = _summary(vec_t)
        rec(g, cid, 'kat_cfg1', {"call": call, "ref": ref, "mask_count": int(mask_t.sum()), **sums, **extra}, {},
            {"mask_packed": np.packbits(mask_t.numpy()), "sub4": sub["sub4"]})

    wt, mt = fields["rot_t"].apply(timg, return_valid_area=True)
    ws, ms = fields["rot_s"].apply(timg, return_valid_area=True)
    assert int(mt.sum()) == 99591 and int(ms.sum()) == 100274
    assert abs(float(wt.double().sum()) - 36876388.645) < 1 and abs(float(ws.double().sum()) - 37024340.638) < 1
    kat('cfg1_apply_t', 'apply', 't', wt, mt, {"flow": "rot_t"})
    kat('cfg1_apply_s', 'apply', 's', ws, ms, {"flow": "rot_s"})
    counts = {}
    for ref in 'ts':
        a, b = fields["rot_" + ref], fields["sc_" + ref]
        for mode in (1, 2, 3):
            out = a.combine_with(b, mode)
            counts["%s%d" % (ref, mode)] = int(out.mask.sum())
            kat('cfg1_combine_%s_mode%d' % (ref, mode), 'combine_with', ref, out.vecs, out.mask,
                {"self": "rot_" + ref, "flow": "sc_" + ref, "mode": mode})
    assert counts == {"t1": 99596, "s1": 118314, "t2": 48627, "s2": 100274, "t3": 58800, "s3": 99591}, counts
    for ref in 'st':
        out = fields["rot_" + ref].switch_ref()
        assert int(out.mask.sum()) == (100274 if ref == 's' else 100273)
        kat('cfg1_switch_ref_' + ref, 'switch_ref', ref, out.vecs, out.mask, {"flow": "rot_" + ref})
    # README example: rotation (+) translation[40, 0], mode 3, 't': mask count 108000
    out = fields["rot_t"].combine_with(fields["tr_t"], 3)
    mag = float(torch.sqrt((out.vecs ** 2).sum(1)).mean())
    assert int(out.mask.sum()) == 108000, int(out.mask.sum())
    kat('cfg1_readme_combine', 'combine_with', 't', out.vecs, out.mask,
        {"self": "rot_t", "flow": "tr_t", "mode": 3, "mean_mag": mag})


# ------------------------------------------------------------------------------------------------
# group: next  (general Flow.combine, track_pts / Flow.track, get_padding)
# ------------------------------------------------------------------------------------------------
def gen_next():
    g = 'next'
    h, w = H, W
    fa, fb = smooth_flow(2, h, w, 3.0, 301), smooth_flow(2, h, w, 2.5, 302)
    ma, mb = hole_mask(2, h, w, 31), hole_mask(2, h, w, 33)
    # Flow.combine: 3 modes x (self ref, other ref, result ref)   (flow_class.py:1812-1939; test_flow_class.py:2142-2231)
    for mode in (1, 2, 3):
        for sr in 'st':
            for orf in 'st':
                for rr in 'st':
                    out = Flow(fa, sr, ma).combine(Flow(fb, orf, mb), mode, rr)
                    rec(g, 'combine_m%d_%s%s%s' % (mode, sr, orf, rr), 'Flow.combine',
                        {"mode": mode, "self_ref": sr, "other_ref": orf, "ref": rr, "out_ref": out.ref},
                        {"f1": fa, "m1": ma, "f2": fb, "m2": mb}, {"vecs": out.vecs, "mask": out.mask})
    out = Flow(fa, 's', ma).combine(Flow(fb, 't', mb), 3)        # ref defaults to self.ref
    rec(g, 'combine_m3_st_default', 'Flow.combine', {"mode": 3, "self_ref": 's', "other_ref": 't', "ref": None,
                                                     "out_ref": out.ref},
        {"f1": fa, "m1": ma, "f2": fb, "m2": mb}, {"vecs": out.vecs, "mask": out.mask})

    # track_pts (utils.py:941-1042; KAT inputs of test_utils.py:944-1040)
    f_s = Flow.from_transforms([['rotation', 0, 0, 30]], (200, 210), 's').vecs
    f_t = Flow.from_transforms([['rotation', 0, 0, 30]], (200, 210), 't').vecs
    pts = torch.tensor([[20.5, 10.5], [8.3, 7.2], [120.4, 160.2]])
    desired = np.array([[12.5035207776, 19.343266740], [3.58801085141, 10.385382907], [24.1694586156, 198.93726969]])
    for ref, fl, rtol in (('s', f_s, 1e-6), ('t', f_t, 5e-3)):
        out = ofu.track_pts(fl, ref, pts)
        np.testing.assert_allclose(out.numpy(), desired, rtol=rtol)          # the reference's own assertion
        rec(g, 'track_pts_%s_kat' % ref, 'track_pts', {"ref": ref, "int_out": None},
            {**_affine_inputs("flow", fl), "pts": pts}, {"out": out})
        out = ofu.track_pts(fl, ref, pts.unsqueeze(0))
        rec(g, 'track_pts_%s_kat_3d' % ref, 'track_pts', {"ref": ref, "int_out": None},
            {**_affine_inputs("flow", fl), "pts": pts.unsqueeze(0)}, {"out": out})
        out = ofu.track_pts(fl, ref, pts, int_out=True)
        rec(g, 'track_pts_%s_kat_int_out' % ref, 'track_pts', {"ref": ref, "int_out": True},
            {**_affine_inputs("flow", fl), "pts": pts}, {"out": out})
    out = ofu.track_pts(torch.zeros_like(f_s), 's', pts)
    rec(g, 'track_pts_zero', 'track_pts', {"ref": 's', "int_out": None, "same_object": False},
        {"flow_raw": torch.zeros(1, 2, 20, 21), "pts": pts[:, :] * 0.1}, {"out": ofu.track_pts(torch.zeros(1, 2, 20, 21), 's', pts * 0.1)})
    # batched smooth flows, float and integer points, points out of range
    fsm = smooth_flow(3, h, w, 4.0, 311)
    gen = torch.Generator().manual_seed(312)
    p3 = torch.rand(3, 9, 2, generator=gen) * torch.tensor([h + 4.0, w + 4.0]) - 2.0
    pi = torch.stack([torch.randint(0, h, (3, 7), generator=gen), torch.randint(0, w, (3, 7), generator=gen)], dim=-1)
    for ref in 'st':
        rec(g, 'track_pts_%s_batched' % ref, 'track_pts', {"ref": ref, "int_out": None}, {"flow_raw": fsm, "pts": p3},
            {"out": ofu.track_pts(fsm, ref, p3)})
        rec(g, 'track_pts_%s_bcast_pts' % ref, 'track_pts', {"ref": ref, "int_out": None}, {"flow_raw": fsm, "pts": p3[0]},
            {"out": ofu.track_pts(fsm, ref, p3[0])})
        rec(g, 'track_pts_%s_intpts' % ref, 'track_pts', {"ref": ref, "int_out": None}, {"flow_raw": fsm, "pts": pi},
            {"out": ofu.track_pts(fsm, ref, pi)})
        rec(g, 'track_pts_%s_intpts_int_out' % ref, 'track_pts', {"ref": ref, "int_out": True}, {"flow_raw": fsm, "pts": pi},
            {"out": ofu.track_pts(fsm, ref, pi, int_out=True)})
    # Flow.track with the status of each point (flow_class.py:961-1020; test_flow_class.py:1315-1385 at 128 x 128)
    for ref in 'st':
        fl = Flow.from_transforms([['rotation', 0, 0, 30]], (128, 128), ref)
        mk = torch.ones(1, 128, 128, dtype=torch.bool)
        mk[:, :, 50:] = False
        fl = Flow(fl.vecs, ref, mk)
        tp = torch.tensor([[0, 12.0], [0, 125.0], [8.3, 7.2], [30.4, 40.2], [75.0, 50.0]])
        o, st = fl.track(tp, get_valid_status=True)
        rec(g, 'flow_track_%s_status' % ref, 'Flow.track', {"ref": ref, "int_out": None},
            {**_affine_inputs("flow", fl.vecs), "m": mk, "pts": tp}, {"out": o, "status": st})
        f3 = of.batch_flows([fl, fl, fl])
        o, st = f3.track(tp.unsqueeze(0), get_valid_status=True)
        rec(g, 'flow_track_%s_status_batched' % ref, 'Flow.track', {"ref": ref, "int_out": None, "batch": 3},
            {**_affine_inputs("flow", fl.vecs), "m": mk, "pts": tp.unsqueeze(0)}, {"out": o, "status": st})

    # get_padding (flow_class.py:1174-1224): the 7 x 7 KATs of test_flow_class.py:1619-1645 and smooth masked flows
    transforms = [['rotation', 0, 0, 45]]
    shape = (7, 7)
    mk_s = np.ones(shape, 'bool'); mk_s[:, 4:] = False
    mk_t = np.ones(shape, 'bool'); mk_t[4:] = False
    kats = {'s': (Flow.from_transforms(transforms, shape, 's'), [5, 0, 0, 3]),
            't': (Flow.from_transforms(transforms, shape, 't'), [0, 3, 5, 0]),
            's_masked': (Flow.from_transforms(transforms, shape, 's', mk_s), [3, 0, 0, 1]),
            't_masked': (Flow.from_transforms(transforms, shape, 't', mk_t), [0, 1, 3, 0])}
    for name, (fl, want) in kats.items():
        got = fl.get_padding()
        assert got[0] == want, (name, got)
        rec(g, 'get_padding_kat_' + name, 'Flow.get_padding', {"ref": fl.ref, "item": None, "padding": got},
            flow_inputs(fl), {})
    fl = Flow(smooth_flow(3, h, w, 6.0, 321), 's', hole_mask(3, h, w, 35))
    for ref in 'st':
        fl2 = Flow(fl.vecs, ref, fl.mask)
        rec(g, 'get_padding_smooth_' + ref, 'Flow.get_padding', {"ref": ref, "item": None, "padding": fl2.get_padding()},
            flow_inputs(fl2), {})
        rec(g, 'get_padding_smooth_item1_' + ref, 'Flow.get_padding', {"ref": ref, "item": 1, "padding": fl2.get_padding(1)},
            flow_inputs(fl2), {})
    tiny = Flow(torch.rand(1, 2, 7, 7, generator=torch.Generator().manual_seed(5)) * 1e-4)
    rec(g, 'get_padding_tiny', 'Flow.get_padding', {"ref": 't', "item": None, "padding": tiny.get_padding()},
        flow_inputs(tiny), {})
    # the functional wrapper (flow_operations.py:280-303): 4-D -> list of lists, 3-D -> one list; H-W-2 numpy input
    v = smooth_flow(2, h, w, 6.0, 322)
    rec(g, 'get_flow_padding_4d_s', 'get_flow_padding', {"ref": 's', "layout": 'nchw', "padding": of.get_flow_padding(v, 's')},
        {"flow_raw": v}, {})
    rec(g, 'get_flow_padding_3d_t', 'get_flow_padding', {"ref": 't', "layout": 'chw', "padding": of.get_flow_padding(v[0], 't')},
        {"flow_raw": v}, {})
    rec(g, 'get_flow_padding_hw2_numpy_t', 'get_flow_padding',
        {"ref": 't', "layout": 'hwc_np', "padding": of.get_flow_padding(v[1].permute(1, 2, 0).numpy(), 't')}, {"flow_raw": v}, {})


# ------------------------------------------------------------------------------------------------
# group: grads  (the reference's own autograd, PyTorch CPU, as the expected gradients)
#   loss = sum(out * w_out) with a stored random w_out; expected = d loss / d input for every floating input
# ------------------------------------------------------------------------------------------------
def _wts(t, seed):
    return torch.randn(t.shape, generator=torch.Generator().manual_seed(seed))


def gen_grads():
    g = 'grads'
    h, w = 20, 28
    f = smooth_flow(2, h, w, 3.0, 401)
    f[1, :, 4:9, 6:15] = 0                       # a zero-flow block: occlusion rule / un-occlude fill on the 's' side
    f2 = smooth_flow(2, h, w, 2.5, 402)
    f[1] += torch.tensor([0.2 * w, -0.25 * h]).view(2, 1, 1) * 0  # (kept in range: borders are exercised by f2 below)
    f2[0] += torch.tensor([0.3 * w, 0.2 * h]).view(2, 1, 1)       # pushes samples / end points over the border
    m, m2 = hole_mask(2, h, w, 41), hole_mask(2, h, w, 43)
    img = image(2, 3, h, w, 45) / 255

    def leaf(t):
        return t.clone().requires_grad_()

    # apply_flow 't' / 's' (utils.py:469-620)
    for ref in 'st':
        for name, fl in (('a', f), ('b', f2)):
            fv, tv = leaf(fl), leaf(img)
            out = ofu.apply_flow(fv, tv, ref, m if ref == 's' else None)
            wo = _wts(out, 411)
            (out * wo).sum().backward()
            rec(g, 'apply_flow_%s_%s' % (ref, name), 'grad_apply_flow', {"ref": ref},
                {"flow": fl, "target": img, "mask": m if ref == 's' else None, "w_out": wo},
                {"g_flow": fv.grad, "g_target": tv.grad})
        fv, tv = leaf(f[:1]), leaf(img)                        # broadcast flow: its gradient sums over the batch
        out = ofu.apply_flow(fv, tv, ref)
        wo = _wts(out, 412)
        (out * wo).sum().backward()
        rec(g, 'apply_flow_%s_bcast_flow' % ref, 'grad_apply_flow', {"ref": ref},
            {"flow": f[:1], "target": img, "w_out": wo}, {"g_flow": fv.grad, "g_target": tv.grad})
    # grid_from_unstructured_data (utils.py:1061-1154): x, y, data; data and density outputs both weighted
    gen = torch.Generator().manual_seed(421)
    x = torch.rand(2, h, w, generator=gen) * (w + 4) - 2
    y = torch.rand(2, h, w, generator=gen) * (h + 4) - 2
    xv, yv, dv = leaf(x), leaf(y), leaf(img)
    od, oden = ofu.grid_from_unstructured_data(xv, yv, dv, m)
    wd, wn = _wts(od, 422), _wts(oden, 423)
    ((od * wd).sum() + (oden * wn).sum()).backward()
    rec(g, 'gfud', 'grad_gfud', {}, {"x": x, "y": y, "data": img, "mask": m, "w_data": wd, "w_density": wn},
        {"g_x": xv.grad, "g_y": yv.grad, "g_data": dv.grad})
    # Flow-level chains: gradients wrt the vectors of both flows
    def flow_case(cid, op, args, fn):
        a, b = leaf(f), leaf(f2)
        out = fn(a, b)
        wo = _wts(out.vecs, 431)
        (out.vecs * wo).sum().backward()
        rec(g, cid, op, args, {"f1": f, "m1": m, "f2": f2, "m2": m2, "w_out": wo},
            {"g_f1": a.grad if a.grad is not None else torch.zeros_like(f),
             "g_f2": b.grad if b.grad is not None else torch.zeros_like(f2), "vecs": out.vecs.detach(), "mask": out.mask})
    for ref in 'st':
        oref = 't' if ref == 's' else 's'
        flow_case('flow_apply_%s' % ref, 'grad_Flow.apply', {"ref": ref, "target_ref": oref},
                  lambda a, b, ref=ref, oref=oref: Flow(a, ref, m).apply(Flow(b, oref, m2)))
        flow_case('switch_ref_%s' % ref, 'grad_Flow.switch_ref', {"ref": ref},
                  lambda a, b, ref=ref: Flow(a, ref, m).switch_ref())
        flow_case('invert_%s' % ref, 'grad_Flow.invert', {"ref": ref},
                  lambda a, b, ref=ref: Flow(a, ref, m).invert())
        for mode in (1, 2, 3):
            flow_case('combine_with_%s_m%d' % (ref, mode), 'grad_Flow.combine_with', {"ref": ref, "mode": mode},
                      lambda a, b, ref=ref, mode=mode: Flow(a, ref, m).combine_with(Flow(b, ref, m2), mode))
    flow_case('combine_m3_st_t', 'grad_Flow.combine', {"mode": 3, "self_ref": 's', "other_ref": 't', "ref": 't'},
              lambda a, b: Flow(a, 's', m).combine(Flow(b, 't', m2), 3, 't'))
    flow_case('combine_m1_ts_s', 'grad_Flow.combine', {"mode": 1, "self_ref": 't', "other_ref": 's', "ref": 's'},
              lambda a, b: Flow(a, 't', m).combine(Flow(b, 's', m2), 1, 's'))
    # track_pts: gradients wrt flow and points
    gen = torch.Generator().manual_seed(441)
    pts = torch.rand(2, 6, 2, generator=gen) * torch.tensor([h - 1.0, w - 1.0])
    for ref in 'st':
        fv, pv = leaf(f), leaf(pts)
        out = ofu.track_pts(fv, ref, pv)
        wo = _wts(out, 442)
        (out * wo).sum().backward()
        rec(g, 'track_pts_' + ref, 'grad_track_pts', {"ref": ref}, {"flow": f, "pts": pts, "w_out": wo},
            {"g_flow": fv.grad, "g_pts": pv.grad, "out": out.detach()})


# ------------------------------------------------------------------------------------------------
# group: gen  (the callers either side of the path: from_matrix / from_transforms utils.py:646-807,
#   resize_flow :878-916, Flow.from_matrix / from_transforms / resize / pad / unpad flow_class.py:238-328, 694-753)
# ------------------------------------------------------------------------------------------------
def gen_generators():
    g = 'gen'
    h, w = 30, 44
    rot = ofu.matrix_from_transforms([['rotation', 12, 9, -25], ['scaling', 20, 14, 1.3]])
    persp = rot.clone()
    persp[2, 0], persp[2, 1] = 4e-4, -7e-4                       # a true homography: the divide matters
    batch = torch.stack([rot, persp, torch.eye(3)])
    for name, mat, ref, inv in (('rot_s', rot, 's', None), ('rot_t', rot, 't', None), ('persp_s', persp, 's', None),
                                ('persp_t', persp, 't', False), ('persp_t_isinv', persp, 't', True),
                                ('batch_s', batch, 's', None), ('batch_t', batch, 't', None)):
        out = ofu.from_matrix(mat, [h, w], ref, inv)
        rec(g, 'from_matrix_' + name, 'from_matrix', {"shape": [h, w], "ref": ref, "matrix_is_inverse": inv}, {"matrix": mat},
            {"out": out})
    out = ofu.from_matrix(persp.numpy(), (1, h, w), 't')            # numpy matrix, 1-H-W shape
    rec(g, 'from_matrix_numpy_1hw', 'from_matrix', {"shape": [1, h, w], "ref": 't', "matrix_is_inverse": None, "numpy": True},
        {"matrix": persp}, {"out": out})
    big = ofu.from_matrix(persp, [150, 260], 's')                    # coordinates beyond 2^7: more mantissa bits in play
    rec(g, 'from_matrix_persp_150x260', 'from_matrix', {"shape": [150, 260], "ref": 's', "matrix_is_inverse": None},
        {"matrix": persp}, {"out": big})
    tlists = {'rot': [['rotation', 10, 20, -30]], 'tr': [['translation', 10, -20]], 'sc': [['scaling', 5, 7, 0.7]],
              'chain': [['translation', -3, 4.5], ['rotation', 20, 10, 33], ['scaling', 15, 12, 1.25]]}
    for name, tl in tlists.items():
        for ref in 'st':
            out = ofu.from_transforms([list(t) for t in tl], [h, w], ref)
            rec(g, 'from_transforms_%s_%s' % (name, ref), 'from_transforms', {"transforms": tl, "shape": [h, w], "ref": ref, "padding": None},
                {}, {"out": out})
    for ref in 'st':
        out = ofu.from_transforms([list(t) for t in tlists['chain']], [2, h, w], ref, padding=[3, 5, 4, 2])
        rec(g, 'from_transforms_chain_padded_batched_' + ref, 'from_transforms',
            {"transforms": tlists['chain'], "shape": [2, h, w], "ref": ref, "padding": [3, 5, 4, 2]}, {}, {"out": out})
    mk = hole_mask(1, h, w, 51)
    fl = Flow.from_matrix(persp, (h, w), 't', mk)
    rec(g, 'Flow_from_matrix_t_mask', 'Flow.from_matrix', {"shape": [h, w], "ref": 't', "matrix_is_inverse": None},
        {"matrix": persp, "m": mk}, {"vecs": fl.vecs, "mask": fl.mask})
    fl = Flow.from_transforms([list(t) for t in tlists['chain']], (h, w), 's', mk, padding=None)
    rec(g, 'Flow_from_transforms_s_mask', 'Flow.from_transforms', {"transforms": tlists['chain'], "shape": [h, w], "ref": 's', "padding": None},
        {"m": mk}, {"vecs": fl.vecs, "mask": fl.mask})
    pm = hole_mask(1, h + 8, w + 6, 52)
    fl = Flow.from_transforms([list(t) for t in tlists['rot']], (h, w), 't', pm, padding=[3, 5, 4, 2])
    rec(g, 'Flow_from_transforms_t_padded', 'Flow.from_transforms', {"transforms": tlists['rot'], "shape": [h, w], "ref": 't', "padding": [3, 5, 4, 2]},
        {"m": pm}, {"vecs": fl.vecs, "mask": fl.mask})
    # resize_flow / Flow.resize (F.interpolate bilinear, align_corners=False; vectors scaled along)
    f = smooth_flow(2, h, w, 4.0, 501)
    for name, sc in (('half', 0.5), ('x2', 2), ('aniso', [1.5, 0.7]), ('tuple', (0.8, 1.25))):
        out = ofu.resize_flow(f, sc)
        rec(g, 'resize_flow_' + name, 'resize_flow', {"scale": list(sc) if isinstance(sc, (list, tuple)) else sc, "tuple": isinstance(sc, tuple)},
            {"flow": f}, {"out": out})
    out = ofu.resize_flow(f[0], 1.5)
    rec(g, 'resize_flow_3d', 'resize_flow', {"scale": 1.5, "squeeze": True}, {"flow": f}, {"out": out})
    out = ofu.resize_flow(f[1].permute(1, 2, 0).contiguous().numpy(), [0.6, 1.4])
    rec(g, 'resize_flow_hw2_numpy', 'resize_flow', {"scale": [0.6, 1.4], "hwc_np": True}, {"flow": f}, {"out": out})
    m = hole_mask(1, h, w, 53)
    for ref in 'st':
        for name, sc in (('half', 0.5), ('aniso', [1.5, 0.7])):
            out = Flow(f[:1], ref, m).resize(sc)
            rec(g, 'Flow_resize_%s_%s' % (name, ref), 'Flow.resize', {"ref": ref, "scale": sc}, {"f": f[:1], "m": m},
                {"vecs": out.vecs, "mask": out.mask})
    try:                                                          # a batch: the reference's mask `.squeeze(0).squeeze(0)` leaves 4 dimensions
        Flow(f, 's', hole_mask(2, h, w, 54)).resize(0.5)
        raised = None
    except (ValueError, TypeError) as exc:
        raised = type(exc).__name__
    rec(g, 'Flow_resize_batched', 'Flow.resize', {"ref": 's', "scale": 0.5, "raises": raised}, {"f": f, "m": hole_mask(2, h, w, 54)}, {})
    # Flow.pad / unpad
    m2 = hole_mask(2, h, w, 55)
    for ref in 'st':
        for mode in (None, 'constant', 'reflect', 'replicate'):
            out = Flow(f, ref, m2).pad([3, 5, 4, 2], mode)
            rec(g, 'Flow_pad_%s_%s' % (mode, ref), 'Flow.pad', {"ref": ref, "padding": [3, 5, 4, 2], "mode": mode}, {"f": f, "m": m2},
                {"vecs": out.vecs, "mask": out.mask})
        out = Flow(f, ref, m2).unpad([3, 5, 4, 2])
        rec(g, 'Flow_unpad_' + ref, 'Flow.unpad', {"ref": ref, "padding": [3, 5, 4, 2]}, {"f": f, "m": m2}, {"vecs": out.vecs, "mask": out.mask})
    out = Flow(f, 't', m2).pad([2, 0, 0, 7]).unpad([2, 0, 0, 7])
    rec(g, 'Flow_pad_unpad_roundtrip', 'Flow.pad_unpad', {"ref": 't', "padding": [2, 0, 0, 7]}, {"f": f, "m": m2},
        {"vecs": out.vecs, "mask": out.mask})


# ------------------------------------------------------------------------------------------------
# group: contract  (SURVEY.md 8b "error conventions": what the reference RAISES, class and message, for a table of bad calls;
#   plus the remaining Flow operators / accessors: / ** with scalars, lists and arrays, select, __getitem__, batch_flows)
# ------------------------------------------------------------------------------------------------
BAD_CALLS = [
    # (id, function, positional args as JSON)
    ('ft_not_list', 'from_transforms', ['test', [20, 30], 't']),
    ('ft_item_not_list', 'from_transforms', [['test'], [20, 30], 't']),
    ('ft_rotation_missing', 'from_transforms', [[['translation', 20, 10], ['rotation']], [20, 30], 't']),
    ('ft_rotation_incomplete', 'from_transforms', [[['translation', 20, 10], ['rotation', 1]], [20, 30], 't']),
    ('ft_rotation_invalid', 'from_transforms', [[['translation', 20, 10], ['rotation', 1, 'test', 10]], [20, 30], 't']),
    ('ft_translation_missing', 'from_transforms', [[['translation', 20, 10], ['translation']], [20, 30], 't']),
    ('ft_translation_incomplete', 'from_transforms', [[['translation', 20, 10], ['translation', 1]], [20, 30], 't']),
    ('ft_translation_invalid', 'from_transforms', [[['translation', 20, 10], ['translation', 1, 'test']], [20, 30], 't']),
    ('ft_scaling_missing', 'from_transforms', [[['translation', 20, 10], ['scaling']], [20, 30], 't']),
    ('ft_scaling_incomplete', 'from_transforms', [[['translation', 20, 10], ['scaling', 1]], [20, 30], 't']),
    ('ft_scaling_invalid', 'from_transforms', [[['translation', 20, 10], ['scaling', 1, 'test', 2]], [20, 30], 't']),
    ('ft_unknown', 'from_transforms', [[['shear', 1, 2]], [20, 30], 't']),
    ('ft_bad_ref', 'from_transforms', [[['translation', 1, 2]], [20, 30], 'x']),
    ('ft_ref_type', 'from_transforms', [[['translation', 1, 2]], [20, 30], 3]),
    ('ft_bad_padding_type', 'from_transforms', [[['translation', 1, 2]], [20, 30], 't', 'pad']),
    ('ft_bad_padding_len', 'from_transforms', [[['translation', 1, 2]], [20, 30], 't', [1, 2, 3]]),
    ('ft_bad_padding_neg', 'from_transforms', [[['translation', 1, 2]], [20, 30], 't', [1, 2, 3, -4]]),
    ('ft_batched_shape', 'from_transforms', [[['translation', 1, 2]], [2, 20, 30], 's']),
    ('ft_bad_shape', 'from_transforms', [[['translation', 1, 2]], [20, 30, 4, 5], 's']),
    ('fm_shape_type', 'from_matrix_eye', ['shape', 't']),
    ('fm_shape_len', 'from_matrix_eye', [[10], 't']),
    ('fm_shape_neg', 'from_matrix_eye', [[10, -3], 't']),
    ('fm_shape_batched', 'from_matrix_eye', [[2, 10, 10], 't']),
    ('fm_inverse_type', 'from_matrix_eye', [[10, 10], 't', 'yes']),
    ('fm_inverse_with_s', 'from_matrix_eye', [[10, 10], 's', True]),
    ('rf_scale_type', 'resize_flow_f', ['test']),
    ('rf_scale_values', 'resize_flow_f', [['test', 0]]),
    ('rf_scale_len', 'resize_flow_f', [[1, 2, 3]]),
    ('rf_scale_zero', 'resize_flow_f', [0]),
    ('rf_scale_neg', 'resize_flow_f', [-0.1]),
    ('pad_mode', 'flow_pad', [[1, 1, 1, 1], 'wrap']),
    ('pad_type', 'flow_pad', ['pad', None]),
    ('pad_len', 'flow_pad', [[1, 2, 3], None]),
    ('pad_float', 'flow_pad', [[1, 2, 3, 4.0], None]),
    ('pad_neg', 'flow_pad', [[1, 2, 3, -4], None]),
    ('unpad_too_much', 'flow_unpad', [[10, 10, 0, 0]]),
    ('mul_list_len', 'flow_mul', [[1, 2, 3]]),
    ('mul_str', 'flow_mul', ['two']),
    ('div_list_len', 'flow_div', [[1, 2, 3]]),
    ('pow_str', 'flow_pow', ['two']),
    ('select_type', 'flow_select', ['0']),
    ('select_range', 'flow_select', [5]),
    ('apply_valid_type', 'flow_apply_kw', [{"return_valid_area": 1}]),
    ('apply_consider_type', 'flow_apply_kw', [{"consider_mask": 'yes'}]),
    ('apply_cut_type', 'flow_apply_kw', [{"cut": 0}]),
    ('apply_padding_mismatch', 'flow_apply_kw', [{"padding": [1, 1, 1, 1]}]),
    ('combine_mode', 'flow_combine_with', [4]),
    ('combine_thresholded', 'flow_combine_with', [3, 'no']),
    ('switch_ref_mode', 'flow_switch_ref', ['maybe']),
    ('invert_ref', 'flow_invert', ['x']),
    ('is_zero_masked', 'flow_is_zero', [None, 'x']),
    ('is_zero_thresholded', 'flow_is_zero', [1, None]),
    # track_pts / Flow.track (utils.py:959-981; flow_class.py:1003-1005) -- args: {pts: 'list' | shape, dtype, ref, int_out, nan}
    ('tp_pts_type', 'track_pts', [{"pts": 'list', "ref": 't'}]),
    ('tp_pts_batch', 'track_pts', [{"pts": [3, 4, 2], "ref": 's'}]),
    ('tp_pts_ndim', 'track_pts', [{"pts": [1, 2, 4, 2], "ref": 's'}]),
    ('tp_pts_ndim1', 'track_pts', [{"pts": [2], "ref": 's'}]),
    ('tp_pts_last_dim', 'track_pts', [{"pts": [4, 3], "ref": 't'}]),
    ('tp_pts_last_dim_batched', 'track_pts', [{"pts": [2, 4, 3], "ref": 's'}]),
    ('tp_int_out_type', 'track_pts', [{"pts": [4, 2], "ref": 's', "int_out": 'yes'}]),
    ('tp_bad_ref', 'track_pts', [{"pts": [4, 2], "ref": 'x'}]),
    ('tp_ref_type', 'track_pts', [{"pts": [4, 2], "ref": 1}]),
    ('tp_flow_nan', 'track_pts', [{"pts": [4, 2], "ref": 's', "nan": True}]),
    ('tp_flow_shape', 'track_pts', [{"pts": [4, 2], "ref": 's', "flow_shape": [2, 3, 12, 16]}]),
    ('track_status_type', 'flow_track', [{"pts": [4, 2], "get_valid_status": 'yes'}]),
    ('track_int_out_type', 'flow_track', [{"pts": [4, 2], "int_out": 1}]),
]


def bad_call(mod, flow_cls, utils_mod, fn, args):
    """Run one row of BAD_CALLS against a package (the reference here, the build in tests/case_runner.py)."""
    f = flow_cls(torch.ones(2, 2, 12, 16), 't')
    if fn == 'from_transforms':
        return utils_mod.from_transforms(*[([list(t) if isinstance(t, list) else t for t in a] if isinstance(a, list) and i == 0 else a)
                                           for i, a in enumerate(args)])
    if fn == 'from_matrix_eye':
        return utils_mod.from_matrix(torch.eye(3), *args)
    if fn == 'resize_flow_f':
        return utils_mod.resize_flow(f.vecs, *args)
    if fn == 'flow_pad':
        return f.pad(*args)
    if fn == 'flow_unpad':
        return f.unpad(*args)
    if fn == 'flow_mul':
        return f * args[0]
    if fn == 'flow_div':
        return f / args[0]
    if fn == 'flow_pow':
        return f ** args[0]
    if fn == 'flow_select':
        return f.select(*args)
    if fn == 'flow_apply_kw':
        return f.apply(torch.zeros(2, 1, 12, 16), **args[0])
    if fn == 'flow_combine_with':
        return f.combine_with(flow_cls(torch.ones(2, 2, 12, 16) * 2, 't'), *args)
    if fn == 'flow_switch_ref':
        return f.switch_ref(*args)
    if fn == 'flow_invert':
        return f.invert(*args)
    if fn == 'flow_is_zero':
        return f.is_zero(*args)
    if fn in ('track_pts', 'flow_track'):
        spec = args[0]
        pts = [[1, 2], [3, 4]] if spec["pts"] == 'list' else torch.ones(*spec["pts"])
        if fn == 'flow_track':
            return f.track(pts, **{k: v for k, v in spec.items() if k != 'pts'})
        vecs = torch.ones(*spec.get("flow_shape", [2, 2, 12, 16]))
        if spec.get("nan"):
            vecs[0, 0, 0, 0] = float('nan')
        return utils_mod.track_pts(vecs, spec["ref"], pts, *([spec["int_out"]] if "int_out" in spec else []))
    raise KeyError(fn)


def gen_contract():
    g = 'contract'
    for cid, fn, args in BAD_CALLS:
        try:
            bad_call(of, Flow, ofu, fn, json.loads(json.dumps(args)))
            got = None
        except Exception as exc:  # noqa: BLE001
            got = [type(exc).__name__, str(exc)]
        assert got is not None, cid
        rec(g, 'raises_' + cid, 'raises', {"fn": fn, "args": args, "exception": got[0], "message": got[1]}, {}, {})
    # operators and accessors with tensors behind them
    h, w = 12, 16
    f = smooth_flow(2, h, w, 2.0, 601)
    m = hole_mask(2, h, w, 61)
    fl = Flow(f, 't', m)
    arr = (torch.rand(2, 2, h, w, generator=torch.Generator().manual_seed(62)) + 0.5)
    for name, op, operand in (('div_scalar', 'div', 2.5), ('div_list', 'div', [2.0, -4.0]), ('div_hw', 'div', arr[0, 0]),
                              ('div_2hw', 'div', arr[0]), ('div_hw2', 'div', arr[0].permute(1, 2, 0).contiguous()), ('div_n2hw', 'div', arr),
                              ('pow_scalar', 'pow', 2), ('pow_list', 'pow', [2, 3]), ('pow_2hw', 'pow', torch.round(arr[0] * 2)),
                              ('mul_list', 'mul', [2.0, -0.5]), ('mul_hw', 'mul', arr[0, 0]), ('mul_n2hw', 'mul', arr),
                              ('mul_numpy', 'mul', arr[0].numpy())):
        out = {'div': lambda a, b: a / b, 'pow': lambda a, b: a ** b, 'mul': lambda a, b: a * b}[op](fl, operand)
        is_arr = isinstance(operand, (torch.Tensor, np.ndarray))
        rec(g, 'flow_' + name, 'Flow.binop', {"ref": 't', "op": op, "operand": None if is_arr else operand,
                                              "numpy": isinstance(operand, np.ndarray)},
            {"f": f, "m": m, **({"operand": operand} if is_arr else {})}, {"vecs": out.vecs, "mask": out.mask})
    out = fl.select(1)
    rec(g, 'flow_select_1', 'Flow.select', {"ref": 't', "item": 1}, {"f": f, "m": m}, {"vecs": out.vecs, "mask": out.mask})
    for name, item in (('rows', [[2, 9, None], None]), ('window', [[1, 10, 2], [3, 14, None]]), ('cols', [None, [4, 12, None]])):
        idx = tuple(slice(*i) if isinstance(i, list) else (slice(None) if i is None else i) for i in item)
        out = fl[idx]
        rec(g, 'flow_getitem_' + name, 'Flow.getitem', {"ref": 't', "item": item}, {"f": f, "m": m}, {"vecs": out.vecs, "mask": out.mask})
    b = of.batch_flows([Flow(f[:1], 's', m[:1]), Flow(f, 's', m), Flow(f[1:], 's')])
    rec(g, 'batch_flows', 'batch_flows', {"ref": 's'}, {"f": f, "m": m}, {"vecs": b.vecs, "mask": b.mask})
    c = fl.copy()
    rec(g, 'flow_copy_str', 'Flow.copy', {"ref": 't', "str": str(c)[:str(c).index(';')], "aliases": bool(c.vecs.data_ptr() == fl.vecs.data_ptr())},
        {"f": f, "m": m}, {"vecs": c.vecs, "mask": c.mask})


# ------------------------------------------------------------------------------------------------
# group: padbc  (round 6: Flow.apply with padding x the 1 <-> N batch broadcast -- VERDICT r5, what's weak 1)
# ------------------------------------------------------------------------------------------------
def gen_padbc():
    """The corner round 5's differential fuzz found: `padding` with `ref = 's'`, a batch-1 target and a batch-N flow.  The reference's
    loop over the TARGET's batch (flow_class.py:884-894) ANDs only element 0's flow mask into the mask channel of every element,
    and an all-zero flow hands the batch-1 target through (utils.py:497-498), so the result stays batch 1.  Recorded as the
    reference computes it: padding x {target batch 1 < n, = n} x {zero, non-zero flow} x {consider_mask} x {cut}."""
    g = 'padbc'
    h, w = HS + 8, WS + 6
    pad = [3, 5, 4, 2]
    hf, wf = h - pad[0] - pad[1], w - pad[2] - pad[3]
    img = image(3, 3, h, w, 141)
    tmask = hole_mask(3, h, w, 109)
    for ref in 'st':
        fnz = smooth_flow(3, hf, wf, 2.5, 160 + ord(ref))
        fm = hole_mask(3, hf, wf, 117)
        for zero in (False, True):
            f = torch.zeros_like(fnz) if zero else fnz
            for tn in (1, 3):
                for cons in (True, False):
                    for cut in (True, False):
                        if ref == 't' and (zero or not cut) and tn == 1:
                            continue        # (the reference itself raises there: a batch-1 mask assigned a batch-N value, flow_class.py:929-932)
                        if not cut and not cons:
                            continue
                        fl = Flow(f.clone(), ref, fm.clone())
                        tm = tmask[:tn].clone()
                        kw = {"return_valid_area": True, "padding": pad, "cut": cut, "consider_mask": cons}
                        out = fl.apply(img[:tn], target_mask=tm, **kw)
                        rec(g, 'apply_%s_pad_%s_tn%d_cons%d_cut%d' % (ref, 'zero' if zero else 'nz', tn, cons, cut), 'Flow.apply',
                            {"ref": ref, "kwargs": kw}, {"f": f, "m": fm, "target": img[:tn], "target_mask": tmask[:tn]},
                            {"warped": out[0], "valid": out[1]})
        # a Flow as the target (batch 1) under a batch-3 flow, padded
        tfl = Flow(smooth_flow(1, h, w, 2.0, 178), ref, hole_mask(1, h, w, 114))
        if ref == 's':
            out = Flow(fnz, ref, fm).apply(tfl, padding=pad, cut=True)
            rec(g, 'apply_%s_pad_flowtarget_tn1' % ref, 'Flow.apply',
                {"ref": ref, "kwargs": {"padding": pad, "cut": True}, "target_ref": ref, "out_ref": out.ref},
                {"f": fnz, "m": fm, "tf": tfl.vecs, "tm": tfl.mask}, {"vecs": out.vecs, "mask": out.mask})


def main():
    """`gen_golden.py` regenerates everything; `gen_golden.py --groups gen ...` only the named groups (the other groups' npz
    files and manifest entries stay as they are, byte for byte)."""
    of.set_pure_pytorch()
    gens = {'prims': gen_prims, 'flow_apply': gen_flow_apply, 'flow_ops': gen_flow_ops, 'kats': gen_kats, 'next': gen_next,
            'grads': gen_grads, 'gen': gen_generators, 'contract': gen_contract, 'padbc': gen_padbc}
    only = sys.argv[sys.argv.index('--groups') + 1:] if '--groups' in sys.argv else list(gens)
    for name in only:
        gens[name]()
    for grp, store in GROUPS.items():
        path = os.path.join(HERE, grp + '.npz')
        np.savez_compressed(path, **store)
        print(grp, len(store), 'arrays', os.path.getsize(path) // 1024, 'KiB')
    mpath = os.path.join(HERE, 'manifest.json')
    cases = MANIFEST
    if set(only) != set(gens) and os.path.exists(mpath):
        with open(mpath) as fh:
            old = json.load(fh)["cases"]
        cases = [c for c in old if c["group"] not in GROUPS] + MANIFEST
    with open(mpath, 'w') as fh:
        json.dump({"reference": "oflibpytorch 2.1.1", "torch": torch.__version__, "cases": cases}, fh, indent=1)
    print(len(cases), 'cases')


if __name__ == '__main__':
    main()
