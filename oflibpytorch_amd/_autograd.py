"""Autograd for the two primitives of the path.

The reference's outputs are differentiable with respect to flow vectors and warped data (README.rst:7-10; 't' branch
through ``F.grid_sample`` and ``normalise_coords``, utils.py:462-465, 549-555; 's' branch through the weights and
``scatter_add_`` of ``grid_from_unstructured_data``, utils.py:1098-1144; its tests assert ``grad_fn``, e.g.
test_utils.py:500, 1113-1114).  Here the forward passes are the fused HIP kernels, so the backward passes are HIP
kernels too (``ofl_warp_bwd_grad_f32``, ``ofl_splat_grad_f32``, ``ofl_sample_pts_grad_f32``); the fused epilogues of
the forward launch (signs, `src_b` / `data_b`, addend) are chained here.  Boolean outputs (valid masks, flag words)
are non-differentiable, as the comparisons that produce them are in the reference.  Double backward is not
implemented (`once_differentiable` raises).
"""
import torch
from torch.autograd.function import once_differentiable

from . import _native


def _reduce_to(g: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """Gradient of a batch-broadcast operand: sum over the batch axis (the reference's `expand`, utils.py:527-537)."""
    if g.shape[0] != like.shape[0]:
        g = g.sum(0, keepdim=True)
    return g.to(like.device).to(like.dtype) if like.dtype.is_floating_point else g


class WarpFn(torch.autograd.Function):
    """dst = a_sign * addend + g_sign * G(flow_sign * flow, src - src_b)"""

    @staticmethod
    def forward(ctx, flow, src, addend, src_b, kw):
        d = lambda t: None if t is None else t.detach()
        with _native._on(_native.device(flow, src)):
            res = _native._warp_bwd_raw(d(flow), d(src), addend=d(addend), src_b=d(src_b), **kw)
        ctx.save_for_backward(flow, src, src_b)
        ctx.addend_meta = None if addend is None else (addend.shape[0], addend.device, addend.dtype)
        ctx.signs = (float(kw.get("flow_sign", 1.0)), float(kw.get("a_sign", 1.0)), float(kw.get("g_sign", 1.0)))
        ctx.mark_non_differentiable(*[r for r in res[1:] if r is not None])
        return tuple(res)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, *_unused):
        flow, src, src_b = ctx.saved_tensors
        flow_sign, a_sign, g_sign = ctx.signs
        need_flow, need_src, need_add, need_b = ctx.needs_input_grad[:4]
        g = g.contiguous()
        gathered = src.detach().float()
        if src_b is not None:
            gathered = gathered.to(g.device) - src_b.detach().float().to(g.device)
        gs, gf = _native.warp_bwd_grad(flow, gathered, g, flow_sign=flow_sign, g_scale=g_sign,
                                       want_src=bool(need_src or need_b), want_flow=bool(need_flow))
        g_flow = _reduce_to(gf, flow) if need_flow else None
        g_src = _reduce_to(gs, src) if need_src else None
        g_b = _reduce_to(-gs, src_b) if (need_b and src_b is not None) else None
        g_add = None
        if need_add and ctx.addend_meta is not None:
            nb, adev, adt = ctx.addend_meta
            g_add = g * a_sign if a_sign != 1.0 else g
            if g_add.shape[0] != nb:
                g_add = g_add.sum(0, keepdim=True)
            g_add = g_add.to(adev).to(adt)
        return g_flow, g_src, g_add, g_b, None


def warp(flow, src, **kw):
    """`_native.warp_bwd` with a grad_fn.  Integer rounding modes have no gradient in the reference either
    (`torch.round`, then a cast back to the integer dtype: flow_class.py:943-951, utils.py:613-618)."""
    if int(kw.get("round_mode", 0)) != 0:
        d = lambda t: None if t is None else t.detach()
        kw = dict(kw)
        kw["addend"], kw["src_b"] = d(kw.get("addend")), d(kw.get("src_b"))
        with _native._on(_native.device(flow, src)):
            return _native._warp_bwd_raw(flow.detach(), src.detach(), **kw)
    kw = dict(kw)
    addend, src_b = kw.pop("addend", None), kw.pop("src_b", None)
    kw.pop("out_uint8", None)
    if src.dtype == torch.uint8:          # (only the flow can want a gradient: the backward kernel reads float planes)
        src = src.float()
    return WarpFn.apply(flow, src, addend, src_b, kw)


class SplatFn(torch.autograd.Function):
    """dst = P(flow_sign * flow | (xs, ys), data_sign * (data - data_b))"""

    @staticmethod
    def forward(ctx, flow, xs, ys, data, data_b, kw):
        d = lambda t: None if t is None else t.detach()
        kw = dict(kw)
        user_density = bool(kw.get("want_density", False))
        kw["want_density"] = True                      # the backward pass needs D
        with _native._on(_native.device(flow, data, xs)):
            res = list(_native._splat_fwd_raw(d(flow), d(data), xs=d(xs), ys=d(ys), data_b=d(data_b), **kw))
        dst, density = res[0], res[2]
        ctx.save_for_backward(flow, xs, ys, data, data_b, dst, density)
        ctx.kw = (float(kw.get("flow_sign", 1.0)), float(kw.get("data_sign", 1.0)), kw.get("weight_mask"),
                  bool(kw.get("occlude", True)))
        if not user_density:
            res[2] = None
        nd = [r for i, r in enumerate(res) if r is not None and i not in (0, 2)]
        ctx.mark_non_differentiable(*nd)
        ctx.user_density = user_density
        return tuple(res)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, *rest):
        flow, xs, ys, data, data_b, dst, density = ctx.saved_tensors
        flow_sign, data_sign, weight_mask, occlude = ctx.kw
        need_flow, need_x, need_y, need_data, need_b = ctx.needs_input_grad[:5]
        g_den = rest[1] if (ctx.user_density and len(rest) > 1) else None
        dev = g.device
        eff = data.detach().float().to(dev)
        if data_b is not None:
            eff = eff - data_b.detach().float().to(dev)
        if data_sign != 1.0:
            eff = eff * data_sign
        gd, gxy = _native.splat_grad(flow, eff, dst, density, g.contiguous(), xs=xs, ys=ys, flow_sign=flow_sign,
                                     weight_mask=weight_mask, occlude=occlude, grad_density=g_den,
                                     want_data=bool(need_data or need_b), want_xy=bool(need_flow or need_x or need_y))
        g_flow = g_x = g_y = g_data = g_b = None
        if need_flow and flow is not None:
            g_flow = _reduce_to(gxy * flow_sign if flow_sign != 1.0 else gxy, flow)
        if need_x and xs is not None:
            g_x = _reduce_to(gxy[:, 0], xs)
        if need_y and ys is not None:
            g_y = _reduce_to(gxy[:, 1], ys)
        if need_data:
            g_data = _reduce_to(gd * data_sign if data_sign != 1.0 else gd, data)
        if need_b and data_b is not None:
            g_b = _reduce_to(gd * (-data_sign), data_b)
        return g_flow, g_x, g_y, g_data, g_b, None


def splat(flow, data, **kw):
    """`_native.splat_fwd` with a grad_fn (see `warp` for the rounding modes)."""
    kw = dict(kw)
    xs, ys, data_b = kw.pop("xs", None), kw.pop("ys", None), kw.pop("data_b", None)
    if int(kw.get("round_mode", 0)) != 0:
        d = lambda t: None if t is None else t.detach()
        with _native._on(_native.device(flow, data, xs)):
            return _native._splat_fwd_raw(d(flow), d(data), xs=d(xs), ys=d(ys), data_b=d(data_b), **kw)
    if not data.dtype.is_floating_point:
        data = data.float()
    return SplatFn.apply(flow, xs, ys, data, data_b, kw)


class SamplePtsFn(torch.autograd.Function):
    """out = pts + bilinear(flow, pts)   (track_pts, utils.py:1004-1015)"""

    @staticmethod
    def forward(ctx, flow, pts):
        ctx.save_for_backward(flow, pts)
        return _native.sample_pts(flow.detach(), pts.detach())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        flow, pts = ctx.saved_tensors
        need_flow, need_pts = ctx.needs_input_grad
        gf, gp = _native.sample_pts_grad(flow, pts, g.contiguous(), want_flow=bool(need_flow), want_pts=bool(need_pts))
        return (_reduce_to(gf, flow) if need_flow else None), (_reduce_to(gp, pts) if need_pts else None)


def sample_pts(flow, pts):
    if _native._wants_grad(flow, pts):
        return SamplePtsFn.apply(flow, pts)
    return _native.sample_pts(flow, pts)
