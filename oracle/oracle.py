"""CPU oracle for the oflibpytorch warp / compose hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package ``oflibpytorch_amd`` never does: its compute path
is the HIP library and it fails loudly when that is missing.

Parity status: PINNED -- the C primitives in ``ofl_oracle.c`` and the closed forms below are
checked bit-for-bit against the imported reference (fixtures in ``tests/golden``) and
against the reference's own known-answer tests (``tests/test_oracle_golden.py``).

Layout: numpy, float32, NCHW, bool masks.  The two primitives are

* ``G(f, S)``      backward gather  (reference ``apply_flow`` 't' branch, utils.py:541-555)
* ``P(f, S, m)``   forward splat with the zero-flow occlusion rule (``apply_s_flow``,
                   utils.py:1157-1205, on top of ``grid_from_unstructured_data`` :1061-1154)

and everything else (``Flow.apply``, ``switch_ref``, ``invert``, ``combine_with`` modes 1-3)
is the composition the reference builds from them (flow_class.py:755-959, 1022-1086,
1648-1810), restated here as closed forms.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("OFL_ORACLE_LIB") or os.path.join(_HERE, "libofl_oracle.so")   # OFL_ORACLE_LIB: `make -C oracle asan`
_lib = None

THRESHOLD = np.float32(1e-3)        # utils.py:23 DEFAULT_THRESHOLD
VALID_THRESHOLD = np.float32(0.99999)  # flow_class.py:922


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "ofl_oracle.c")
    if os.environ.get("OFL_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "libofl_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        i32 = ctypes.c_int32
        i64 = ctypes.c_int64
        L.orc_warp_bwd_f32.argtypes = [fp, i64, fp, i64, fp, i32, i32, i32, i32]
        L.orc_normalise_coords_f32.argtypes = [fp, fp, i64, i32, i32]
        L.orc_flow_endpoints_f32.argtypes = [fp, ctypes.c_float, fp, fp, i32, i32, i32]
        L.orc_grid_from_unstructured_f32.argtypes = [fp, fp, fp, u8, fp, fp, i32, i32, i32, i32]
        L.orc_apply_s_flow_f32.argtypes = [fp, fp, u8, i32, fp, u8, fp, i32, i32, i32, i32]
        L.orc_flow_flags_f32.argtypes = [fp, u8, ctypes.c_float, ctypes.POINTER(i32), i32, i32, i32]
        L.orc_sample_pts_f32.argtypes = [fp, i64, fp, i64, fp, i32, i32, i32, i32]
        L.orc_sample_pts_f32.restype = None
        L.orc_flow_from_matrix_f32.argtypes = [fp, i64, ctypes.c_float, fp, i32, i32, i32]
        L.orc_flow_from_matrix_f32.restype = None
        L.orc_resize_bilinear_f32.argtypes = [fp, fp, i32, i32, i32, i32, i32, ctypes.c_float, ctypes.c_float]
        L.orc_resize_bilinear_f32.restype = None
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_set_threads.argtypes = [ctypes.c_int]
        for name in ("orc_warp_bwd_f32", "orc_normalise_coords_f32", "orc_flow_endpoints_f32",
                     "orc_grid_from_unstructured_f32", "orc_apply_s_flow_f32", "orc_flow_flags_f32",
                     "orc_set_threads"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(np.asarray(a) != 0, dtype=np.uint8)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _up(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def set_threads(n: int):
    lib().orc_set_threads(int(n))


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ----------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------
def G(flow, src):
    """Backward gather.  flow [N|1,2,H,W], src [N|1,C,H,W] -> [N,C,H,W]  (utils.py:541-555)."""
    flow, src = _f32(flow), _f32(src)
    n = max(flow.shape[0], src.shape[0])
    c, h, w = src.shape[1:]
    assert flow.shape[2:] == (h, w) and flow.shape[1] == 2
    assert flow.shape[0] in (1, n) and src.shape[0] in (1, n)
    out = np.empty((n, c, h, w), np.float32)
    fbs = 0 if (flow.shape[0] == 1 and n > 1) else 2 * h * w
    sbs = 0 if (src.shape[0] == 1 and n > 1) else c * h * w
    lib().orc_warp_bwd_f32(_fp(flow), fbs, _fp(src), sbs, _fp(out), n, c, h, w)
    return out


def normalise_coords(coords, shape):
    coords = _f32(coords)
    out = np.empty_like(coords)
    lib().orc_normalise_coords_f32(_fp(coords), _fp(out), coords.size // 2, int(shape[0]), int(shape[1]))
    return out


def flow_endpoints(flow, ref):
    """utils.py:1045-1058."""
    flow = _f32(flow)
    n, _, h, w = flow.shape
    x = np.empty((n, h, w), np.float32)
    y = np.empty((n, h, w), np.float32)
    lib().orc_flow_endpoints_f32(_fp(flow), 1.0 if ref == 's' else -1.0, _fp(x), _fp(y), n, h, w)
    return x, y


def grid_from_unstructured_data(x, y, data, mask=None):
    """utils.py:1061-1154 -> (grid_data [N,C,H,W], density [N,H,W])."""
    x, y, data = _f32(x), _f32(y), _f32(data)
    n, c, h, w = data.shape
    m = None if mask is None else _u8(mask)
    out = np.empty((n, c, h, w), np.float32)
    den = np.empty((n, h, w), np.float32)
    lib().orc_grid_from_unstructured_f32(_fp(x), _fp(y), _fp(data), _up(m), _fp(out), _fp(den), n, c, h, w)
    return out, den


def apply_s_flow(flow, data, mask=None, occlude_zero_flow=True, return_density=False):
    """utils.py:1157-1205 -> (warped [N,C,H,W], warped_mask [N,H,W] bool[, density])."""
    flow, data = _f32(flow), _f32(data)
    n, c, h, w = data.shape
    assert flow.shape == (n, 2, h, w)
    m = None if mask is None else _u8(mask)
    out = np.empty((n, c, h, w), np.float32)
    wm = np.empty((n, h, w), np.uint8)
    den = np.empty((n, h, w), np.float32)
    lib().orc_apply_s_flow_f32(_fp(flow), _fp(data), _up(m), int(bool(occlude_zero_flow)), _fp(out), _up(wm),
                               _fp(den), n, c, h, w)
    if return_density:
        return out, wm.astype(bool), den
    return out, wm.astype(bool)


def P(flow, src, mask=None):
    """Forward splat with occlusion rule; broadcasts 1<->N like apply_flow (utils.py:527-537)."""
    flow, src = _f32(flow), _f32(src)
    n = max(flow.shape[0], src.shape[0])
    if flow.shape[0] != n:
        flow = np.broadcast_to(flow, (n,) + flow.shape[1:])
        if mask is not None:
            mask = np.broadcast_to(mask, (n,) + mask.shape[1:])
    if src.shape[0] != n:
        src = np.broadcast_to(src, (n,) + src.shape[1:])
    return apply_s_flow(flow, src, mask, True)[0]


def sample_pts(flow, pts):
    """track_pts' sampler for floating-point points (utils.py:1004-1015, 1033-1035): flow [N|1,2,H,W], pts [N|1,M,2]
    as (y, x) -> pts + bilinearly sampled flow, NaN rows zeroed."""
    flow, pts = _f32(flow), _f32(pts)
    n = max(flow.shape[0], pts.shape[0])
    m = pts.shape[1]
    h, w = flow.shape[2:]
    out = np.empty((n, m, 2), np.float32)
    if m:
        lib().orc_sample_pts_f32(_fp(flow), 0 if (flow.shape[0] == 1 and n > 1) else 2 * h * w, _fp(pts),
                                 0 if (pts.shape[0] == 1 and n > 1) else 2 * m, _fp(out), n, m, h, w)
    return out


def track_pts(flow, ref, pts, int_out=False):
    """utils.py:941-1042 (PURE_PYTORCH), floating-point or integer points [N|1,M,2] -> [N,M,2]."""
    flow = _f32(flow)
    pts = np.asarray(pts)
    n = flow.shape[0]
    if pts.shape[0] != n:
        pts = np.broadcast_to(pts, (n,) + pts.shape[1:])
    if bool(np.all(is_zero_flow(flow, True))):                    # :988-989
        out = pts
    else:
        if ref == 't':                                            # :993-996
            x, y = flow_endpoints(neg(flow), 's')
            flow, _ = grid_from_unstructured_data(x, y, flow)
        if pts.dtype.kind in 'iu':                                # :998-1003
            h, w = flow.shape[2:]
            lin = (pts[..., 0] * w + pts[..., 1]).astype(np.int64)
            fv = np.take_along_axis(np.moveaxis(flow, 1, -1).reshape(n, h * w, 2), lin[..., None].repeat(2, -1), axis=1)
            out = pts.astype(np.float32) + fv[..., ::-1]
        else:
            out = sample_pts(flow, pts.astype(np.float32))
    if int_out:
        out = np.rint(out).astype(np.int64)
    return out


def flow_extents(flow, mask, sign):
    """The reduction of Flow.get_padding (flow_class.py:1196-1219): per batch element min y, max y, min x, max x of the
    positions -(sign * thr(v) - grid) under the mask, and whether any pixel is valid."""
    flow = _f32(flow).copy()
    n, _, h, w = flow.shape
    flow[(flow < THRESHOLD) & (flow > -THRESHOLD)] = 0            # threshold_vectors utils.py:642
    v = flow * np.float32(sign)
    gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    px, py = -(v[:, 0] - gx), -(v[:, 1] - gy)
    out = np.zeros((n, 5), np.float32)
    for b in range(n):
        mk = np.ones((h, w), bool) if mask is None else np.asarray(mask[b], bool)
        if mk.any():
            out[b] = [py[b][mk].min(), py[b][mk].max(), px[b][mk].min(), px[b][mk].max(), 1.0]
    return out


def flow_from_matrix(matrix, n, h, w, sign=1.0):
    """utils.py:339-376 (flow_from_matrix); sign = -1: the negated result of the 't' branches (utils.py:699-705, 804-807).
    matrix [n|1, 3, 3] -> [n, 2, h, w]."""
    m = _f32(np.asarray(matrix, np.float32).reshape(-1, 9))
    out = np.empty((n, 2, h, w), np.float32)
    lib().orc_flow_from_matrix_f32(_fp(m), 0 if m.shape[0] == 1 else 9, float(sign), _fp(out), n, h, w)
    return out


def resize_output_size(h, w, scale):
    """(oh, ow) of F.interpolate(scale_factor=scale): floor of the double product (torch/nn/functional.py)."""
    import math
    return int(math.floor(float(h) * float(scale[0]))), int(math.floor(float(w) * float(scale[1])))


def resize_bilinear(x, scale):
    """F.interpolate(x [N,C,H,W], scale_factor=[sh, sw], mode='bilinear', align_corners=False) as ATen's CPU kernel computes it
    (ofl_oracle.c: orc_resize_bilinear_f32)."""
    x = _f32(x)
    n, c, h, w = x.shape
    oh, ow = resize_output_size(h, w, scale)
    out = np.empty((n, c, oh, ow), np.float32)
    lib().orc_resize_bilinear_f32(_fp(x), _fp(out), n * c, h, w, oh, ow, float(np.float32(1.0 / float(scale[0]))),
                                  float(np.float32(1.0 / float(scale[1]))))
    return out


def resize_flow(flow, scale):
    """utils.py:878-916: the interpolated field with its components scaled along ([:, 0] by the width's factor, [:, 1] by the
    height's)."""
    out = resize_bilinear(flow, scale)
    out[:, 0] *= np.float32(scale[1])
    out[:, 1] *= np.float32(scale[0])
    return out


def flow_flags(flow, mask=None, thr=THRESHOLD):
    """Per-batch-element flag word (bit meanings: ofl_oracle.c, orc_flow_flags_f32)."""
    flow = _f32(flow)
    n, _, h, w = flow.shape
    m = None if mask is None else _u8(np.broadcast_to(mask, (n, h, w)))
    flags = np.zeros(n, np.int32)
    lib().orc_flow_flags_f32(_fp(flow), _up(m), float(thr), flags.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                             n, h, w)
    return flags


FLAG_NONFINITE, FLAG_NZ, FLAG_NZ_THR, FLAG_NZ_MASKED, FLAG_NZ_THR_MASKED = 1, 2, 4, 8, 16


def is_zero_flow(flow, thresholded=True):
    """utils.py:919-938 (per batch element)."""
    fl = flow_flags(flow)
    return (fl & (FLAG_NZ_THR if thresholded else FLAG_NZ)) == 0


def flow_is_zero(flow, mask, thresholded=True, masked=True):
    """Flow.is_zero flow_class.py:1226-1244 (per batch element)."""
    fl = flow_flags(flow, mask if masked else None)
    if masked:
        return (fl & (FLAG_NZ_THR_MASKED if thresholded else FLAG_NZ_MASKED)) == 0
    return (fl & (FLAG_NZ_THR if thresholded else FLAG_NZ)) == 0


def theta(ch):
    """Validity threshold on a warped mask channel (flow_class.py:922)."""
    return np.asarray(ch, np.float32) > VALID_THRESHOLD


# ----------------------------------------------------------------------------------------
# closed forms (reference flow_class.py)
# ----------------------------------------------------------------------------------------
def _cat_mask(data, m, n):
    """t = cat(t, mask.float()) with batch expansion (flow_class.py:896-898)."""
    data = _f32(data)
    if data.shape[0] != n:
        data = np.broadcast_to(data, (n,) + data.shape[1:])
    m = np.broadcast_to(np.asarray(m, bool), (n,) + data.shape[2:])
    return np.concatenate([data, m[:, None].astype(np.float32)], axis=1)


def _apply_flow(flow, target, ref, mask):
    """apply_flow (utils.py:469-620) on N-C-H-W float input incl. the zero-flow early exit (:497)."""
    if bool(np.all(is_zero_flow(flow, True))):
        return _f32(target)
    if ref == 't':
        return G(flow, target)
    return P(flow, target, mask)


def flow_apply(f, ref, m, target, target_mask=None, consider_mask=True):
    """Flow(f, ref, m).apply(target, target_mask, return_valid_area=True, consider_mask) without padding.

    f [N,2,H,W], m [N,H,W] bool, target [N|1,C,H,W] float, target_mask [N|1,H,W] bool or None.
    Returns (warped [N',C,H,W], valid [N',H,W]).   flow_class.py:755-959.
    """
    f = _f32(f)
    m = np.asarray(m, bool)
    target = _f32(target)
    tm = np.ones((target.shape[0],) + target.shape[2:], bool) if target_mask is None else np.asarray(target_mask, bool)
    if ref == 's':
        tm = tm & m                                               # flow_class.py:895
    n = max(tm.shape[0], target.shape[0])
    t = _cat_mask(target, tm, n)                                  # :896-898
    warped = _apply_flow(f, t, ref, m if consider_mask else None) # :904
    valid = theta(warped[:, -1])                                  # :922
    if ref == 't':
        valid = valid & m                                         # :934
    return warped[:, :-1], valid


def flow_apply_to_flow(f, ref, m, tf, tmask):
    """Flow(f, ref, m).apply(Flow(tf, *, tmask)) -> (vecs, mask)   (flow_class.py:839-842, 938)."""
    return flow_apply(f, ref, m, tf, tmask, True)


def neg(f):
    """Flow.__neg__ = self * -1 (flow_class.py:680-692, :549)."""
    return _f32(f) * np.float32(-1.0)


def switch_ref(f, ref, m):
    """Flow.switch_ref('valid') (flow_class.py:1022-1062) -> (vecs, mask, new_ref)."""
    f = _f32(f)
    m = np.asarray(m, bool)
    new_ref = 't' if ref == 's' else 's'
    if bool(np.all(flow_is_zero(f, m, thresholded=False))):       # :1046
        return f, m, new_ref
    if ref == 's':
        v, vm = flow_apply_to_flow(f, 's', m, f, m)               # :1050
    else:
        v, vm = flow_apply_to_flow(neg(f), 's', m, f, m)          # :1054-1055
    return v, vm, new_ref


def invert(f, ref, m, out_ref=None):
    """Flow.invert (flow_class.py:1064-1086) -> (vecs, mask, ref)."""
    f = _f32(f)
    m = np.asarray(m, bool)
    out_ref = ref if out_ref is None else out_ref
    if ref == 's':
        if out_ref == 's':
            v, vm = flow_apply_to_flow(f, 's', m, neg(f), m)      # :1079 self.apply(-self)
            return v, vm, 's'
        return neg(f), m, 't'                                     # :1081
    if out_ref == 's':
        return neg(f), m, 's'                                     # :1084
    v, vm, r = switch_ref(neg(f), 's', m)                         # :1086
    return v, vm, r


def combine_with(f_self, m_self, f_flow, m_flow, mode, ref, thresholded=False):
    """Flow(f_self, ref, m_self).combine_with(Flow(f_flow, ref, m_flow), mode) -> (vecs, mask, ref).

    flow_class.py:1648-1810 incl. the early exits (:1729-1744).
    """
    f1, f2 = _f32(f_self), _f32(f_flow)
    m1, m2 = np.asarray(m_self, bool), np.asarray(m_flow, bool)
    if bool(np.all(flow_is_zero(f1, m1, thresholded))):           # :1729
        return f2, m2, ref
    if bool(np.all(flow_is_zero(f2, m2, thresholded))):           # :1738
        if mode in (1, 2):
            return invert(f1, ref, m1)
        return f1, m1, ref
    if mode == 3:
        if ref == 's':                                            # :1804  self + self.invert('t').apply(flow)
            g, gm = flow_apply_to_flow(neg(f1), 't', m1, f2, m2)
            return f1 + g, m1 & gm, 's'
        g, gm = flow_apply_to_flow(f2, 't', m2, f1, m1)           # :1808  flow + flow.apply(self)
        return f2 + g, m2 & gm, 't'
    if mode == 2:
        if ref == 's':                                            # :1768  self.apply(flow - self)
            v, vm = flow_apply_to_flow(f1, 's', m1, f2 - f1, m2 & m1)
            return v, vm, 's'
        iv, im, _ = invert(f1, 't', m1)                           # :1773  flow - flow.apply(self.invert().apply(self))
        cv, cm = flow_apply_to_flow(iv, 't', im, f1, m1)
        dv, dm = flow_apply_to_flow(f2, 't', m2, cv, cm)
        return f2 - dv, m2 & dm, 't'
    if mode == 1:
        if ref == 's':                                            # :1759-1760
            fi, fim = neg(f2), m2                                 # flow.invert('t')
            sv, sm, _ = switch_ref(f1, 's', m1)                   # self.switch_ref() -> 't'
            av, am = flow_apply_to_flow(fi, 't', fim, sv, sm)     # flow_inv_t.apply(self.switch_ref())
            s_v, s_m = fi + av, fim & am                          # flow_inv_t + ...
            bv, bm = flow_apply_to_flow(s_v, 't', s_m, f1, m1)    # (...).apply(self)
            return f2 - bv, m2 & bm, 's'
        iv, im, _ = invert(f1, 't', m1)                           # :1763  self.invert().apply(flow - self)
        v, vm = flow_apply_to_flow(iv, 't', im, f2 - f1, m2 & m1)
        return v, vm, 't'
    raise ValueError("mode")
