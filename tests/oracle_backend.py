"""Oracle-backed stand-ins for the three primitives of `oflibpytorch_amd._native` -- TESTS ONLY.

The product has exactly one compute backend (the HIP library).  To exercise the host-side mirror of the
reference API (validation, broadcasting, early exits, padding, dtype rules, composition chains) in the
CPU-only test tier, `tests/conftest.py` monkeypatches `_native.flow_flags / warp_bwd / splat_fwd` with
the functions below, which compute the same contracts with the CPU oracle (`oracle/`).  Nothing in the
package imports this file.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402


def _np(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a if dtype is None else a.astype(dtype)


def _bcast(a, n):
    return a if a.shape[0] == n else np.broadcast_to(a, (n,) + a.shape[1:])


def _round(a, mode):
    if mode:
        a = np.rint(a)
        if mode == 2:
            a = np.clip(a, 0, 255)
    return a.astype(np.float32)


def device(*_operands):
    return torch.device('cpu')


def _no_grad(*tensors):
    """The oracle has no backward pass: a tensor that wants a gradient must not slip through silently (the HIP
    primitives route such calls through oflibpytorch_amd._autograd)."""
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise NotImplementedError("oracle-backed primitives (CPU test tier) are forward-only")


def flow_from_matrix(matrix, n, h, w, sign=1.0):
    return torch.tensor(oracle.flow_from_matrix(_np(matrix, np.float32), n, h, w, sign))


def flow_flags(vecs, mask=None):
    m = None if mask is None else _np(mask)
    return torch.tensor(oracle.flow_flags(_np(vecs, np.float32), m), dtype=torch.int32)


def warp_bwd(flow, src, *, flow_sign=1.0, src_mask=None, flow_mask=None, want_valid=False, addend=None,
             a_sign=1.0, g_sign=1.0, round_mode=0, want_flags=False, want_src_flags=False, want_dst_flags=False, src_b=None, out_uint8=False):
    _no_grad(flow, src, addend, src_b)
    f = _np(flow, np.float32) * np.float32(flow_sign)
    s = _np(src, np.float32)
    if src_b is not None:
        sb = _np(src_b, np.float32)
        nn = max(s.shape[0], sb.shape[0])
        s = _bcast(s, nn) - _bcast(sb, nn)
    n = max(f.shape[0], s.shape[0], 1 if src_mask is None else src_mask.shape[0],
            1 if flow_mask is None else flow_mask.shape[0], 1 if addend is None else addend.shape[0])
    s = _bcast(s, n)
    c = s.shape[1]
    if want_valid:
        sm = np.ones((n,) + s.shape[2:], np.float32) if src_mask is None else _bcast(_np(src_mask, np.float32), n)
        s = np.concatenate([s, sm[:, None]], axis=1)
    g = oracle.G(_bcast(f, n), s)
    valid = None
    if want_valid:
        valid = oracle.theta(g[:, c])
        if flow_mask is not None:
            valid = valid & _bcast(_np(flow_mask).astype(bool), n)
        valid = torch.tensor(valid)
    out = g[:, :c]
    if addend is not None:
        out = np.float32(a_sign) * _bcast(_np(addend, np.float32), n) + np.float32(g_sign) * out
    ff = sf = None
    if want_flags:
        ff = flow_flags(flow.expand(n, -1, -1, -1), None if flow_mask is None else flow_mask.expand(n, -1, -1))
        if want_src_flags:
            sf = flow_flags(src.expand(n, -1, -1, -1), None if src_mask is None else src_mask.expand(n, -1, -1))
    res = (torch.tensor(_round(out, round_mode)), valid, ff, sf)
    if want_dst_flags:
        res = res + (flow_flags(res[0], valid),)
    return res


def splat_fwd(flow, data, *, xs=None, ys=None, flow_sign=1.0, data_sign=1.0, weight_mask=None, chan_mask_a=None,
              chan_mask_b=None, want_valid=False, occlude=True, want_density=False, want_warped=False, round_mode=0,
              want_mask_chan=False, want_dst_flags=False, data_b=None, out_half=False):
    _no_grad(flow, data, xs, ys, data_b)
    d = _np(data, np.float32)
    if data_b is not None:
        db = _np(data_b, np.float32)
        nn = max(d.shape[0], db.shape[0])
        d = _bcast(d, nn) - _bcast(db, nn)
    d = d * np.float32(data_sign)
    n = max(d.shape[0], 1 if flow is None else flow.shape[0], 1 if xs is None else xs.shape[0],
            1 if weight_mask is None else weight_mask.shape[0], 1 if chan_mask_a is None else chan_mask_a.shape[0],
            1 if chan_mask_b is None else chan_mask_b.shape[0])
    d = _bcast(d, n)
    c, h, w = d.shape[1:]
    if want_valid or want_mask_chan:
        mc = np.ones((n, h, w), bool)
        for cm in (chan_mask_a, chan_mask_b):
            if cm is not None:
                mc = mc & _bcast(_np(cm).astype(bool), n)
        d = np.concatenate([d, mc[:, None].astype(np.float32)], axis=1)
    wm = None if weight_mask is None else _bcast(_np(weight_mask).astype(bool), n)
    if flow is not None:
        f = _bcast(_np(flow, np.float32) * np.float32(flow_sign), n)
        out, warped, den = oracle.apply_s_flow(f, d, wm, bool(occlude), return_density=True)
    else:
        out, den = oracle.grid_from_unstructured_data(_bcast(_np(xs, np.float32), n), _bcast(_np(ys, np.float32), n), d, wm)
        warped = den > 0
    valid = torch.tensor(oracle.theta(out[:, c])) if want_valid else None
    if want_mask_chan:
        valid = torch.tensor(out[:, c].copy())
    res = (torch.tensor(_round(out[:, :c], round_mode)), valid,
           torch.tensor(den) if want_density else None, torch.tensor(warped) if want_warped else None)
    if want_dst_flags:
        res = res + (flow_flags(res[0], valid if want_valid and not want_mask_chan else None),)
    return res


def warp_valid(flow, mask, flow_sign=1.0, thr=0.9999):
    f = _np(flow, np.float32) * np.float32(flow_sign)
    ones = np.ones((f.shape[0], 1) + f.shape[2:], np.float32)
    area = oracle.G(f, ones)[:, 0] > np.float32(thr)
    if mask is not None:
        area = area & _bcast(_np(mask).astype(bool), f.shape[0])
    return torch.tensor(area)


def sample_pts(flow, pts):
    _no_grad(flow, pts)
    return torch.tensor(oracle.sample_pts(_np(flow, np.float32), _np(pts, np.float32)))


def flow_extents(vecs, mask, sign):
    return torch.tensor(oracle.flow_extents(_np(vecs, np.float32), None if mask is None else _np(mask), sign))


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _pad_to(t, window, h, w, mode):
    import torch.nn.functional as F
    fh, fw = t.shape[-2:]
    lrtb = (window[1], w - fw - window[1], window[0], h - fh - window[0])
    if t.dtype == torch.bool:
        return F.pad(t.unsqueeze(1), lrtb).squeeze(1)
    return F.pad(t, lrtb, mode=mode)


def warp_bwd_win(flow, src, window, *, src_mask=None, flow_mask=None, want_valid=False, round_mode=0):
    h, w = src.shape[-2:]
    fm = torch.ones((flow.shape[0],) + tuple(flow.shape[2:]), dtype=torch.bool) if (flow_mask is None and want_valid) else flow_mask
    out = warp_bwd(_pad_to(flow, window, h, w, 'constant'), src, src_mask=src_mask,
                   flow_mask=None if fm is None else _pad_to(fm, window, h, w, None), want_valid=want_valid, round_mode=round_mode)
    return out[0], out[1]


def splat_fwd_win(flow, data, window, *, weight_mask=None, chan_mask_a=None, chan_mask_b=None, want_valid=False, occlude=True,
                  round_mode=0):
    h, w = data.shape[-2:]
    cb = torch.ones((flow.shape[0],) + tuple(flow.shape[2:]), dtype=torch.bool) if chan_mask_b is None else chan_mask_b
    out = splat_fwd(_pad_to(flow, window, h, w, 'replicate'), data,
                    weight_mask=None if weight_mask is None else _pad_to(weight_mask, window, h, w, None),
                    chan_mask_a=chan_mask_a, chan_mask_b=_pad_to(cb, window, h, w, None), want_valid=want_valid, occlude=occlude,
                    round_mode=round_mode)
    return out[0], out[1]
