"""Compact, exact encodings used by the larger golden fixtures (shared by gen_golden.py and the tests).

Affine flow fields (rotation / scaling / translation from `from_transforms`) are stored as a float64
affine model plus the per-pixel difference to the fp32 original in units-in-the-last-place, which is
a few small integers and deflates ~20:1.  Decoding is pure numpy elementwise arithmetic (IEEE,
deterministic on every host) and reproduces the original fp32 bits exactly.
"""
import numpy as np


def _key(f32: np.ndarray) -> np.ndarray:
    """Monotonic int64 key of fp32 values (an involution on the int32 bit pattern)."""
    i = f32.view(np.int32).astype(np.int64)
    return np.where(i < 0, np.int64(-2 ** 31) - i - 1 + 0 * i, i)


def _unkey(k: np.ndarray) -> np.ndarray:
    i = np.where(k < 0, np.int64(-2 ** 31) - k - 1, k).astype(np.int32)
    return i.view(np.float32)


def _predict(params: np.ndarray, h: int, w: int) -> np.ndarray:
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    out = np.empty((params.shape[0], h, w), np.float32)
    for c in range(params.shape[0]):
        out[c] = (params[c, 0] * xx + params[c, 1] * yy + params[c, 2]).astype(np.float32)
    return out


_ESC = -2 ** 15   # sentinel in `delta`: the value is stored verbatim in `escapes` (raster order)


def encode_affine(field: np.ndarray):
    """field [C,H,W] float32 -> (params [C,3] float64, delta [C,H,W] int16, escapes float32[K])"""
    field = np.ascontiguousarray(field, np.float32)
    c, h, w = field.shape
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    a = np.stack([xx.ravel(), yy.ravel(), np.ones(h * w)], axis=1)
    params = np.stack([np.linalg.lstsq(a, field[i].astype(np.float64).ravel(), rcond=None)[0] for i in range(c)])
    delta = _key(field) - _key(_predict(params, h, w))
    esc = np.abs(delta) >= 2 ** 15 - 1          # near zero crossings one ulp is tiny: keep those verbatim
    escapes = field[esc].copy()
    delta = np.where(esc, _ESC, delta).astype(np.int16)
    assert np.array_equal(decode_affine(params, delta, escapes).view(np.int32), field.view(np.int32))
    return params, delta, escapes


def decode_affine(params: np.ndarray, delta: np.ndarray, escapes: np.ndarray) -> np.ndarray:
    c, h, w = delta.shape
    esc = delta == _ESC
    out = _unkey(_key(_predict(np.asarray(params, np.float64), h, w)) + np.where(esc, 0, delta).astype(np.int64))
    out = out.reshape(c, h, w).copy()
    out[esc] = np.asarray(escapes, np.float32)
    return out


def subsample(a: np.ndarray, step: int = 4) -> np.ndarray:
    """Exact values on a coarse lattice (last two axes)."""
    return np.ascontiguousarray(a[..., ::step, ::step])
