"""Host-side mirror of the reference's functional layer for the warp / compose hot path.

Same names, argument meaning, defaults and error behaviour as ``oflibpytorch.utils`` (reference
``src/oflibpytorch/utils.py``; cited per function); the arithmetic runs in the HIP kernels of
``libofl_hip.so`` through :mod:`oflibpytorch_amd._native`.  Validation, shape plumbing and the flow
generators are plain PyTorch host code.
"""
import math
import threading
from typing import Any, Union

import numpy as np
import torch

from . import _native

DEFAULT_THRESHOLD = 1e-3   # utils.py:23
PURE_PYTORCH = True        # utils.py:24 -- the HIP path implements the PURE_PYTORCH=True behaviour


def get_pure_pytorch() -> bool:
    """utils.py:27-33"""
    return PURE_PYTORCH


def set_pure_pytorch(warn: bool = None):
    """utils.py:36-51"""
    global PURE_PYTORCH
    PURE_PYTORCH = True
    if bool(warn):
        print("Pure Pytorch mode set: no use of scipy.interpolate.griddata. Differentiable, significantly faster, "
              "but more approximate")


def unset_pure_pytorch(warn: bool = None):
    """utils.py:54-69.  The flag is kept for API compatibility; the griddata (Delaunay, CPU-only) variants of
    the 's'-reference operations are outside the accelerated path and raise NotImplementedError."""
    global PURE_PYTORCH
    PURE_PYTORCH = False
    if bool(warn):
        print("Pure Pytorch mode unset: scipy.interpolate.griddata used. Not all methods remain differentiable, "
              "significantly slower, but more accurate")


HALF_FLOW_OUTPUTS = False  # extension (off: the reference returns fp32 flows): see set_half_flow_outputs


def get_half_flow_outputs() -> bool:
    return HALF_FLOW_OUTPUTS


def set_half_flow_outputs(on: bool = True):
    """Extension for fp16-stored pipelines (BASELINE config 5): when on, an operation whose flow operands are ALL stored
    in fp16 on a HIP device and whose result is a flow (`switch_ref`, `invert`, `Flow.apply(Flow)` of 's' flows) stores
    that result in fp16 as well (fp32 arithmetic, one round-to-nearest-even at the store; `.vecs` still hands out fp32).
    Off by default: the reference up-casts on entry and returns fp32 flows (utils.py:95, 118)."""
    global HALF_FLOW_OUTPUTS
    HALF_FLOW_OUTPUTS = bool(on)


def _griddata_unavailable(what: str):
    raise NotImplementedError("oflibpytorch_amd: %s with PURE_PYTORCH unset needs scipy.interpolate.griddata, which "
                              "is outside the MI355X hot path; call set_pure_pytorch()" % what)


# ------------------------------------------------------------------------------------------------
# validators (utils.py:72-232): TypeError for wrong types, ValueError for wrong shapes / values
# ------------------------------------------------------------------------------------------------
_readback = {}
_readback_lock = threading.Lock()
_POLL_SPINS = 20000       # ~ a millisecond of polling before the blocking wait (a validation reduction takes 35 - 250 us)


def _flags_to_host(flags: torch.Tensor) -> list:
    """Read-back of DEVICE-resident flag words (a kernel's by-product): an asynchronous copy into a small pinned buffer and a
    POLLED event instead of `flags.cpu()` -- a blocking synchronisation sleeps on an interrupt, and its wake-up latency varies
    from a few microseconds to a few hundred with the host's power state.  The buffer and the event are per device and
    shared, so the whole exchange runs under a lock (two threads validating flows on one device); the spin is bounded, then
    the event is waited for the ordinary way.  (Validation of a NEW tensor does not come through here: `_host_flags`.)"""
    if flags.device.type != 'cuda' or flags.dtype != torch.int32:
        return [int(v) for v in flags.cpu().tolist()]
    n = int(flags.numel())
    with _readback_lock:
        slot = _readback.get(flags.device)
        if slot is None or slot[0].numel() < n:
            slot = (torch.empty(max(n, 64), dtype=torch.int32).pin_memory(), torch.cuda.Event())
            _readback[flags.device] = slot
        buf, done = slot
        with torch.cuda.device(flags.device):
            buf[:n].copy_(flags.reshape(-1), non_blocking=True)
            done.record()
            for _ in range(_POLL_SPINS):
                if done.query():
                    break
            else:
                done.synchronize()
        return buf[:n].tolist()


def _host_flags(vecs: torch.Tensor, mask: torch.Tensor = None) -> list:
    """Flag words of a flow tensor as host ints -- the wait every validation ends in.  One launch whose last block writes the
    words to host-visible memory (`_native.flow_flags_host`); shapes / devices that route does not take fall back to the
    reduction + copy + polled event."""
    host = _native.flow_flags_host(vecs, mask)
    if host is None:
        host = _flags_to_host(_native.flow_flags(vecs, mask))
    return host


def get_valid_vecs(vecs: Any, desired_shape: Union[tuple, list] = None, error_string: str = None,
                   _check_finite: bool = True, _keep_half: bool = False) -> torch.Tensor:
    """utils.py:72-118 -> N-2-H-W float tensor.  ``_check_finite=False`` is for internal callers that get the
    finiteness flag as a by-product of the kernel they are about to launch."""
    error_string = '' if error_string is None else error_string
    if not isinstance(vecs, (np.ndarray, torch.Tensor)):
        raise TypeError(error_string + "Input is not a numpy array or a torch tensor")
    ndim = len(vecs.shape)
    if ndim not in (3, 4):
        raise ValueError(error_string + "Input has {} dimensions, should be 3 or 4".format(ndim))
    if isinstance(vecs, np.ndarray):
        vecs = torch.tensor(vecs, dtype=torch.float, device='cpu')
    if ndim == 3:
        vecs = vecs.unsqueeze(0)
    if vecs.shape[1] != 2:
        if vecs.shape[3] == 2:                      # N-H-W-2 -> N-2-H-W
            vecs = move_axis(vecs, -1, 1)
        else:
            raise ValueError(error_string + "Input needs to be shape (N-)H-W-2 or (N-)2-H-W")
    if _keep_half and not _check_finite and vecs.dtype == torch.float16 and vecs.device.type == 'cuda':
        return vecs          # Flow.__init__ converts and validates in one fused pass once the mask is known
    vecs = vecs.float()
    if _check_finite:
        finite = not any(f & _native.FLAG_NONFINITE for f in _host_flags(vecs))   # (a HIP reduction: no host path)
        if not finite:
            raise ValueError(error_string + "Input contains NaN, Inf or -Inf values")
    if desired_shape is not None:
        d = get_valid_shape(desired_shape)
        if vecs.shape[0] != d[0] or vecs.shape[2] != d[1] or vecs.shape[3] != d[2]:
            raise ValueError(error_string + "Input shape does not match the desired shape")
    return vecs


def get_valid_shape(shape: Any) -> tuple:
    """utils.py:121-132"""
    if not isinstance(shape, (list, tuple)):
        raise TypeError("Error creating flow from matrix: Dims need to be a list or a tuple")
    if len(shape) not in (2, 3):
        raise ValueError("Error creating flow from matrix: Dims need to be a list or a tuple of length 2 or 3")
    if any((not isinstance(item, int) or item <= 0) for item in shape):
        raise ValueError("Error creating flow from matrix: Dims need to be a list or a tuple of integers above zero")
    return ((1,) + tuple(shape)) if len(shape) == 2 else tuple(shape)


def get_valid_ref(ref: Any) -> str:
    """utils.py:134-148"""
    if ref is None:
        return 't'
    if not isinstance(ref, str):
        raise TypeError("Error setting flow reference: Input is not a string")
    if ref not in ('s', 't'):
        raise ValueError("Error setting flow reference: Input is not 's' or 't', but {}".format(ref))
    return ref


def get_valid_mask(mask: Any, desired_shape: Union[tuple, list] = None, error_string: str = None) -> torch.Tensor:
    """utils.py:151-186 -> N-H-W bool tensor"""
    error_string = '' if error_string is None else error_string
    if not isinstance(mask, (np.ndarray, torch.Tensor)):
        raise TypeError(error_string + "Input is not a numpy array or a torch tensor")
    ndim = len(mask.shape)
    if ndim not in (2, 3):
        raise ValueError(error_string + "Input has {} dimensions, should be 2 or 3".format(ndim))
    if isinstance(mask, np.ndarray):
        mask = torch.tensor(mask)
    if mask.dtype != torch.bool and bool(((mask != 0) & (mask != 1)).any()):   # a bool tensor is 0/1 by construction
        raise ValueError(error_string + "Values must be 0 or 1")
    if ndim == 2:
        mask = mask.unsqueeze(0)
    if desired_shape is not None and tuple(mask.shape) != get_valid_shape(desired_shape):
        raise ValueError(error_string + "Input shape does not match the desired shape")
    return mask.to(torch.bool)


def get_valid_device(device: Any) -> torch.device:
    """utils.py:189-212"""
    if device is None:
        device = torch.device('cpu')
    elif not isinstance(device, torch.device):
        try:
            device = torch.device(device)
        except (RuntimeError, TypeError):
            raise ValueError("Error setting tensor device: Input needs to be a torch.device, or valid input to "
                             "torch.device(). Instead found {}".format(device))
    if device.type == 'cuda':
        if not torch.cuda.is_available():
            raise ValueError("Error setting tensor device: Input is 'cuda', but cuda is not available")
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
    return device


def get_valid_padding(padding: Any, error_string: str = None) -> list:
    """utils.py:215-232"""
    error_string = '' if error_string is None else error_string
    if not isinstance(padding, (list, tuple)):
        raise TypeError(error_string + "Padding needs to be a list [top, bot, left, right]")
    if len(padding) != 4:
        raise ValueError(error_string + "Padding list needs to be a list of length 4 [top, bot, left, right]")
    if not all(isinstance(item, int) for item in padding):
        raise ValueError(error_string + "Padding list [top, bot, left, right] items need to be integers")
    if not all(item >= 0 for item in padding):
        raise ValueError(error_string + "Padding list [top, bot, left, right] items need to be 0 or larger")
    return padding


# ------------------------------------------------------------------------------------------------
# tensor helpers (utils.py:235-299)
# ------------------------------------------------------------------------------------------------
def move_axis(input_tensor: torch.Tensor, source: int, destination: int) -> torch.Tensor:
    """np.moveaxis for tensors (utils.py:235-253)"""
    return torch.movedim(input_tensor, source, destination)


def to_numpy(tensor: torch.Tensor, switch_channels: bool = None) -> np.ndarray:
    """utils.py:256-275"""
    arr = tensor.detach().cpu().numpy()
    return np.moveaxis(arr, 1, -1) if switch_channels else arr


def to_tensor(array: np.ndarray, switch_channels: str = None, device=None) -> torch.Tensor:
    """utils.py:278-299"""
    device = get_valid_device(device)
    if switch_channels == 'single':
        array = np.moveaxis(array, -1, 0)
    elif switch_channels == 'batched':
        array = np.moveaxis(array, -1, 1)
    return torch.tensor(array).to(device)


# ------------------------------------------------------------------------------------------------
# flow generators (utils.py:339-442, 646-807): the 3 x 3 algebra on the host, the O(HW) field on the HIP device
# ------------------------------------------------------------------------------------------------
def matrix_from_transform(transform: str, values: list) -> torch.Tensor:
    """3x3 matrix of one transform (utils.py:396-424): translation [dx, dy]; rotation [cx, cy, deg ccw];
    scaling [cx, cy, factor].  Image convention: y points down."""
    m = torch.eye(3)
    if transform == 'translation':
        m[0, 2], m[1, 2] = values[0], values[1]
    elif transform in ('rotation', 'scaling'):
        to_origin = matrix_from_transform('translation', [-values[0], -values[1]])
        back = matrix_from_transform('translation', [values[0], values[1]])
        if transform == 'scaling':
            m[0, 0] = m[1, 1] = values[2]
        else:
            a = math.radians(values[2])
            m[0:2, 0:2] = torch.tensor([[math.cos(a), math.sin(a)], [-math.sin(a), math.cos(a)]])
        m = back @ m @ to_origin
    return m


def matrix_from_transforms(transform_list: list) -> torch.Tensor:
    """utils.py:379-393: product of the transforms, first list entry applied first"""
    m = torch.eye(3)
    for t in reversed(transform_list):
        m = m @ matrix_from_transform(t[0], t[1:])
    return m


def reverse_transform_values(transform_list: list) -> list:
    """utils.py:427-442"""
    out = []
    for t in transform_list:
        name, v = t[0], t[1:]
        if name == 'translation':
            out.append([name, -v[0], -v[1]])
        elif name == 'scaling':
            out.append([name, v[0], v[1], 1 / v[2]])
        elif name == 'rotation':
            out.append([name, v[0], v[1], -v[2]])
    return out


def flow_from_matrix(matrix: torch.Tensor, shape: list, _sign: float = 1.0, _device=None) -> torch.Tensor:
    """'s'-reference flow of a homography: M [x, y, 1]^T dehomogenised, minus [x, y] (utils.py:339-376).
    matrix N-3-3, shape [N, H, W] -> N-2-H-W on the matrix's device.  The O(HW) field is generated by `ofl_flow_from_matrix_f32`
    on the HIP device (bit-identical to the reference's PyTorch-CPU result; `_sign` = -1: the exact negation of the reference's
    't' branches).  A matrix that wants a gradient takes the torch expression (the reference's flow is differentiable wrt it).
    `_device` (internal: Flow.from_matrix / from_transforms with a HIP `device`): leave the field there instead of taking it
    to the matrix's device and back."""
    n, h, w = shape
    if n == 1 and matrix.dim() == 3:
        n = matrix.shape[0]          # (the reference ignores shape[0]: its matmul broadcasts the N matrices over the one grid, utils.py:364-370; a 2-D 3 x 3 matrix is ONE matrix)
    dev = matrix.device if _device is None else _device
    if _native._wants_grad(matrix):
        # differentiable wrt the matrix (utils.py:342): the torch expression, on the device the field is wanted on -- the
        # matrix is MOVED there (`.to` keeps the graph; it may live on the host or on another device than `_device`)
        gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
        hom = torch.stack((gx.float().to(dev), gy.float().to(dev), torch.ones((h, w), device=dev)), dim=-1)   # H-W-3
        moved = torch.matmul(matrix.float().to(dev).expand(n, -1, -1).unsqueeze(1).unsqueeze(1), hom.unsqueeze(-1)).squeeze(-1)   # N-H-W-3
        pts = moved[..., 0:2] / moved[..., 2:3]
        out = move_axis(pts - hom[..., 0:2], -1, 1)
        return -out if _sign < 0 else out
    if dev.type == 'cuda' and matrix.device != dev:
        matrix = matrix.to(dev)                              # (9 floats per matrix: the field is generated where it is wanted)
    return _native.flow_from_matrix(matrix, n, h, w, _sign).to(dev)


def from_matrix(matrix, shape, ref: str = None, matrix_is_inverse: bool = None, _device=None) -> torch.Tensor:
    """Flow vectors N-2-H-W from a transformation matrix (utils.py:646-705): for 's' the matrix is applied directly, for 't'
    its (pseudo-)inverse gives the backward flow (negated).  The batch size comes from the matrix (3-3 or N-3-3); the
    shape is H-W or 1-H-W.  Checks in the reference's order: shape, matrix, ref, matrix_is_inverse."""
    dims = get_valid_shape(shape)
    if dims[0] != 1:
        raise ValueError("Error creating flow from matrix: Given shape has batch dimension larger than 1")
    if not isinstance(matrix, (np.ndarray, torch.Tensor)):
        raise TypeError("Error creating flow from matrix: Matrix needs to be a numpy array or a torch tensor")
    if isinstance(matrix, np.ndarray):
        matrix = torch.tensor(matrix)
    ndim = len(matrix.shape)
    if ndim != 2 and ndim != 3:
        raise ValueError("Error creating flow from matrix: Matrix has {} dimensions, should be 2 or 3".format(ndim))
    if tuple(matrix.shape[-2:]) != (3, 3):
        raise ValueError("Error creating flow from matrix: Matrix needs to be of shape (3, 3)")
    if ndim == 2:
        matrix = matrix.unsqueeze(0)
    matrix = matrix.to(torch.float)
    ref = get_valid_ref(ref)
    matrix_is_inverse = False if matrix_is_inverse is None else matrix_is_inverse
    if not isinstance(matrix_is_inverse, bool):
        raise TypeError("Error creating flow from matrix: Matrix_is_inverse needs to be None or a Boolean")
    dims = (matrix.shape[0],) + dims[1:]
    if ref == 's':
        if matrix_is_inverse:
            raise ValueError("Error creating flow from matrix: Matrix_is_inverse cannot be True when ref is 's'")
        return flow_from_matrix(matrix, list(dims), _device=_device)
    if not matrix_is_inverse:
        # the 3 x 3 pseudo-inverse is formed by the HOST's LAPACK path whatever device the matrix lives on (unless it wants a
        # gradient): nine numbers, and the same bits as the reference's PyTorch-CPU result (a GPU SVD rounds differently)
        if matrix.device.type != 'cpu' and not _native._wants_grad(matrix):
            matrix = torch.pinverse(matrix.cpu()).to(matrix.device)
        else:
            matrix = torch.pinverse(matrix)
    return flow_from_matrix(matrix, list(dims), _sign=-1.0, _device=_device)


def from_transforms(transform_list: list, shape, ref: str = None, padding: list = None, _device=None) -> torch.Tensor:
    """Flow vectors 1-2-H-W from a list of transforms (utils.py:707-807), checks in the reference's order and with its messages:
    's' uses the matrix of the transforms, 't' the matrix of the reversed transforms in reverse order (no numerical inverse),
    both through `from_matrix` -- so `shape` is H-W (or 1-H-W; without padding).  `padding` [top, bot, left, right] grows the
    field and shifts rotation / scaling centres along (the reference computes the padded shape from shape[0], shape[1]:
    it, and so this, means H-W there).  Unlike the reference, the caller's transform lists are not shifted in place."""
    ref = get_valid_ref(ref)
    if padding is not None:
        padding = get_valid_padding(padding, "Error padding flow: ")
        shape = [shape[0] + sum(padding[0:2]), shape[1] + sum(padding[2:4])]
    if not isinstance(transform_list, list):
        raise TypeError("Error creating flow from transforms: Transform_list needs to be a list")
    if not all(isinstance(item, list) for item in transform_list):
        raise TypeError("Error creating flow from transforms: Transform_list needs to be a list of lists")
    if not all(len(item) > 1 for item in transform_list):
        raise ValueError("Error creating flow from transforms: Invalid transforms passed")
    transforms = []
    for t in transform_list:
        t = list(t)
        if t[0] == 'translation':
            if not len(t) == 3:
                raise ValueError("Error creating flow from transforms: Not enough transform values passed for "
                                 "'translation' - expected 2, got {}".format(len(t) - 1))
        elif t[0] in ('rotation', 'scaling'):
            if not len(t) == 4:
                raise ValueError("Error creating flow from transforms: Not enough transform values passed for "
                                 "'{}' - expected 3, got {}".format(t[0], len(t) - 1))
            if padding is not None:
                t[1] += padding[2]
                t[2] += padding[0]
        else:
            raise ValueError("Error creating flow from transforms: Transform '{}' not recognised".format(t[0]))
        if not all(isinstance(item, (float, int)) for item in t[1:]):
            raise ValueError("Error creating flow from transforms: "
                             "Transform values for '{}' need to be integers or floats".format(t[0]))
        transforms.append(t)
    if ref == 's':
        return from_matrix(matrix_from_transforms(transforms), shape, ref, matrix_is_inverse=False, _device=_device)
    matrix = matrix_from_transforms(list(reversed(reverse_transform_values(transforms))))
    return from_matrix(matrix, shape, ref, matrix_is_inverse=True, _device=_device)


# ------------------------------------------------------------------------------------------------
# hot path
# ------------------------------------------------------------------------------------------------
def normalise_coords(coords: torch.Tensor, shape: Union[tuple, list]) -> torch.Tensor:
    """Pixel coordinates (x, y) -> [-1, 1] (utils.py:445-466).  API helper; inside the warp kernel the same
    four fp32 operations are applied in the same order."""
    if len(shape) != 2:
        raise ValueError("Error normalising coords: Given shape needs to be list or tuple of length 2")
    out = coords.float() * 2
    out[..., 0] /= (shape[1] - 1)
    out[..., 1] /= (shape[0] - 1)
    out -= 1
    return out


def _round_mode(dtype: torch.dtype) -> int:
    if dtype.is_floating_point:
        return _native.ROUND_NONE
    return _native.ROUND_U8 if dtype == torch.uint8 else _native.ROUND_RINT


def apply_flow(flow, target: torch.Tensor, ref: str, mask=None) -> torch.Tensor:
    """Warp `target` with `flow` (utils.py:469-620).  't': backward bilinear gather; 's': forward splat with the
    zero-flow occlusion rule.  Returns a tensor of the target's shape and dtype on the flow's device."""
    ref = get_valid_ref(ref)
    flow = get_valid_vecs(flow, error_string="Error applying flow to a target: ", _check_finite=False)
    flags = _host_flags(flow)
    if any(f & _native.FLAG_NONFINITE for f in flags):                          # utils.py:98
        raise ValueError("Error applying flow to a target: Input contains NaN, Inf or -Inf values")
    if not any(f & _native.FLAG_NZ_THR for f in flags):                         # utils.py:497-498
        return target
    if not isinstance(target, torch.Tensor):
        raise TypeError("Error applying flow to a target: Target needs to be a torch tensor")
    if target.dim() not in (2, 3, 4):
        raise ValueError("Error applying flow to a target: Target tensor needs to have shape H-W, C-H-W, or N-C-H-W")
    if target.shape[-2:] != flow.shape[-2:]:
        raise ValueError("Error applying flow to a target: Target height and width needs to match flow field array")
    if mask is not None:
        mask = get_valid_mask(mask, desired_shape=(flow.shape[0],) + tuple(flow.shape[2:]))
    dims, dtype = target.dim(), target.dtype
    t = target if dims == 4 else (target.unsqueeze(0) if dims == 3 else target.unsqueeze(0).unsqueeze(0))
    if t.shape[0] != flow.shape[0] and t.shape[0] != 1 and flow.shape[0] != 1:
        raise ValueError("Error applying flow to target: Batch dimensions for flow ({}) and target ({}) don't match"
                         .format(flow.shape[0], t.shape[0]))
    rm = _round_mode(dtype)
    if ref == 't':
        out = _native.warp_bwd(flow, t, round_mode=rm, out_uint8=True)[0]
    else:
        if not get_pure_pytorch():
            _griddata_unavailable("apply_flow(ref='s')")
        out = _native.splat_fwd(flow, t, weight_mask=mask, occlude=True, round_mode=rm)[0]
    out = out.to(flow.device)
    if out.shape[0] == 1:
        if dims == 2:
            out = out[0, 0]
        elif dims == 3:
            out = out[0]
    elif dims == 2:
        out = out[:, 0]
    return out.to(dtype)


# The device resize restates the arithmetic of ATen's CPU bilinear kernels AS PROBED ON torch 2.10.0 (x86-64, AVX-512 build): which of
# its two kernels runs (output height + width <= 128 or above) and where it contracts to FMAs are properties of that build, not of an
# API -- on another torch version or CPU ISA the host result may differ from this restatement in the last bit (the GPU tests compare
# exactly on the probed version and within 1e-6 of the scale otherwise).  False: resize on a HIP device goes through ATen's own GPU
# kernel, as the reference's does on the same device (last-bit differences from the reference's CPU values).
RESIZE_MATCHES_ATEN_CPU = True
RESIZE_PROBED_TORCH = "2.10"


def interpolate_bilinear(x: torch.Tensor, scale) -> torch.Tensor:
    """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False) (utils.py:908, flow_class.py:710).  On the host
    that is ATen's CPU kernel, as in the reference.  On a HIP device `ofl_resize_bilinear_f32` restates the arithmetic of ATen's
    CPU kernels, so the result equals the reference's PyTorch-CPU values bit for bit (ATen's GPU kernel would differ in the last
    bits); a tensor that wants a gradient keeps ATen's differentiable op."""
    if RESIZE_MATCHES_ATEN_CPU and x.device.type == 'cuda' and x.dtype == torch.float32 and not _native._wants_grad(x):
        return _native.resize_bilinear(x, scale)
    import torch.nn.functional as F
    return F.interpolate(x, scale_factor=[float(v) for v in scale], mode='bilinear', align_corners=False)


def resize_flow(flow, scale) -> torch.Tensor:
    """Bilinear resize of a flow field with the vectors scaled along (utils.py:878-916)."""
    valid_flow = get_valid_vecs(flow, error_string="Error resizing flow: ")
    if isinstance(scale, (float, int)):
        scale = [scale, scale]
    elif isinstance(scale, (tuple, list)):
        if len(scale) != 2:
            raise ValueError("Error resizing flow: Scale {} must have a length of 2".format(type(scale)))
        if not all(isinstance(item, (float, int)) for item in scale):
            raise ValueError("Error resizing flow: Scale {} items must be integers or floats".format(type(scale)))
    else:
        raise TypeError("Error resizing flow: Scale must be an integer, float, or list or tuple of integers or floats")
    if any(s <= 0 for s in scale):
        raise ValueError("Error resizing flow: Scale values must be larger than 0")
    resized = interpolate_bilinear(valid_flow, scale)
    resized[:, 0] *= scale[1]
    resized[:, 1] *= scale[0]
    return resized.squeeze(0) if len(flow.shape) == 3 else resized


def threshold_vectors(vecs: torch.Tensor, threshold: Union[float, int] = None, use_mag: bool = None) -> torch.Tensor:
    """Zero every component strictly inside (-threshold, threshold) (utils.py:623-643).  API helper (elementwise
    host-side PyTorch); the kernels apply the same strict test in-register."""
    threshold = DEFAULT_THRESHOLD if threshold is None else threshold
    out = vecs.clone()
    if use_mag:
        out[torch.norm(vecs, dim=1, keepdim=True).expand(-1, 2, -1, -1) < threshold] = 0
    else:
        out[(vecs < threshold) & (vecs > -threshold)] = 0
    return out


def is_zero_flow(flow, thresholded: bool = None) -> torch.Tensor:
    """Per batch element: are all vectors zero (utils.py:919-938)?  One fused reduction kernel."""
    flow = get_valid_vecs(flow, error_string="Error checking whether flow is zero: ", _check_finite=False)
    thresholded = True if thresholded is None else thresholded
    if not isinstance(thresholded, bool):
        raise TypeError("Error checking whether flow is zero: Thresholded needs to be a boolean")
    host = _host_flags(flow)
    if any(f & _native.FLAG_NONFINITE for f in host):
        raise ValueError("Error checking whether flow is zero: Input contains NaN, Inf or -Inf values")
    bit = _native.FLAG_NZ_THR if thresholded else _native.FLAG_NZ
    return torch.tensor([(f & bit) == 0 for f in host], dtype=torch.bool, device=flow.device)


def _points_per_flow(pts, n_flows: int, err: str):
    """Points handed to `track_pts` as one [N, M, 2] (y, x) tensor, one set per flow, and whether the caller passed a
    single un-batched M-2 set (utils.py:961-978: same checks, same messages -- pinned by the `contract` fixtures)."""
    if not isinstance(pts, torch.Tensor):
        raise TypeError(err + "Pts needs to be a numpy array or a torch tensor")
    unbatched = pts.dim() == 2
    if pts.dim() not in (2, 3):
        raise ValueError(err + "Pts needs to have shape M-2 or N-M-2")
    sets = pts[None] if unbatched else pts
    if sets.shape[0] not in (1, n_flows):
        raise ValueError(err + "If used, pts batch size needs to be equal to the flow batch size")
    if sets.shape[-1] != 2:
        raise ValueError(err + "Pts needs to have shape M-2 or N-M-2")
    return sets.expand(n_flows, -1, -1), unbatched


def track_pts(flow, ref: str, pts: torch.Tensor, int_out: bool = None) -> torch.Tensor:
    """Where the points (y, x) of shape M-2 or N-M-2 end up under a flow field (utils.py:941-1042).  Differentiable wrt the
    flow and the points.  The displacement of a point is the flow read AT the point's own position, so the field must live on
    the points' grid: an 's' flow does; a 't' flow (PURE_PYTORCH) is first carried to its start points -- the reference's
    `grid_from_unstructured_data` at `get_flow_endpoints(-flow, 's')` (:993-996), here one forward-splat launch with the end
    points formed in the kernel.  Float points sample the field bilinearly (`ofl_sample_pts_f32`: the reference's flip ->
    normalise_coords -> grid_sample -> flip, :1004-1014); integer points read their own pixel (:998-1003)."""
    err = "Error tracking points: "
    flow = get_valid_vecs(flow, error_string=err, _check_finite=False)
    words = _host_flags(flow)
    if any(wd & _native.FLAG_NONFINITE for wd in words):                        # utils.py:98
        raise ValueError(err + "Input contains NaN, Inf or -Inf values")
    ref = get_valid_ref(ref)
    points, unbatched = _points_per_flow(pts, flow.shape[0], err)
    int_out = False if int_out is None else int_out
    if not isinstance(int_out, bool):
        raise TypeError(err + "Int_out needs to be a boolean")
    points = points.to(flow.device)

    moved = points
    if any(wd & _native.FLAG_NZ_THR for wd in words):                           # (all below the threshold: nothing moves, :988-989)
        field = flow
        if ref == 't':
            if not get_pure_pytorch():
                _griddata_unavailable("track_pts(ref='t')")
            field = _native.splat_fwd(flow, flow, flow_sign=-1.0, occlude=False)[0].to(flow.device)
        if points.dtype.is_floating_point:
            from . import _autograd
            moved = _autograd.sample_pts(field, points.float()).to(flow.device)  # (a point whose sample is NaN comes back as 0)
        else:
            # row-major pixel index of every point, then both planes read at once; (u, v) -> (dy, dx)
            n, _, h, w = field.shape
            pixel = (points[..., 0] * w + points[..., 1]).long()
            uv = torch.gather(field.reshape(n, 2, h * w), 2, pixel[:, None, :].expand(-1, 2, -1))
            moved = points.float() + uv.flip(1).transpose(1, 2)
            moved = torch.where(torch.isnan(moved).any(dim=-1, keepdim=True), torch.zeros_like(moved), moved)   # :1033-1035
    if int_out:
        moved = torch.round(moved).long()
    return moved.squeeze(0) if unbatched else moved


def get_flow_endpoints(flow: torch.Tensor, ref: str) -> tuple:
    """End ('s') / start ('t') point grids x, y of shape N-H-W (utils.py:1045-1058).  API helper; the splat kernel
    computes the same `s * flow + arange` in-register."""
    n, _, h, w = flow.shape
    s = +1 if ref == 's' else -1
    x = s * flow[:, 0] + torch.arange(w, device=flow.device)[None, None, :]
    y = s * flow[:, 1] + torch.arange(h, device=flow.device)[None, :, None]
    return x, y


def grid_from_unstructured_data(x: torch.Tensor, y: torch.Tensor, data: torch.Tensor, mask: torch.Tensor = None) -> tuple:
    """Inverse-bilinear splat of `data` at positions (x, y) onto the regular grid (utils.py:1061-1154).
    Returns (grid_data N-C-H-W, density N-H-W)."""
    out, _, density, _ = _native.splat_fwd(None, data, xs=x, ys=y, weight_mask=mask, occlude=False, want_density=True)
    return out.to(data.device), density.to(data.device)


def apply_s_flow(flow: torch.Tensor, data: torch.Tensor, mask: torch.Tensor = None, occlude_zero_flow: bool = None) -> tuple:
    """Forward warp with an 's'-reference flow (utils.py:1157-1205).  Returns (warped N-C-H-W, mask N-H-W bool of
    the positions data was warped to)."""
    occlude_zero_flow = True if occlude_zero_flow is None else occlude_zero_flow
    out, _, _, warped = _native.splat_fwd(flow, data, weight_mask=mask, occlude=bool(occlude_zero_flow),
                                          want_warped=True)
    return out.to(flow.device), warped.to(flow.device)
