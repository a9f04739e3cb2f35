"""`Flow`: host-side mirror of the reference's flow object for the warp / compose hot path.

Same constructor, properties, operators and method signatures as ``oflibpytorch.Flow`` (reference
``src/oflibpytorch/flow_class.py``; cited per method).  The object is a thin container -- `vecs`
N-2-H-W fp32, `mask` N-H-W bool, `ref` 's'/'t', `device` -- and every warp / splat / composition is
one fused launch of the HIP kernels in ``libofl_hip.so`` (see :mod:`oflibpytorch_amd._native`).

Differences from the reference that do not change results:
  * validation is one fused reduction (finiteness + the four zero tests) per tensor version,
    cached on the object, instead of a pass per predicate per call;
  * an all-True default mask is kept implicit (`None`) until somebody asks for `.mask`;
  * intermediates produced by the kernels are not re-validated eagerly.
"""
import warnings
import weakref
from typing import Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import _native, distributed
from .utils import (get_valid_vecs, get_valid_ref, get_valid_mask, get_valid_device, get_valid_padding,
                    get_valid_shape, get_pure_pytorch, move_axis, from_matrix, from_transforms, resize_flow,
                    apply_flow, _flags_to_host, _host_flags, _griddata_unavailable, track_pts, get_half_flow_outputs,
                    interpolate_bilinear)

FlowAlias = 'Flow'
_VALID_THR = 0.99999   # flow_class.py:922
_COMBINE_FUSED = True  # False (tests / tools/bench_combine.py): Flow.combine runs its plan through the public operators, one launch per operator


class _NeverEqual(object):
    """Key component that matches nothing, not even itself: a flag word cached under it is never taken as current."""
    __slots__ = ()

    def __eq__(self, other):
        return False

    def __ne__(self, other):
        return True

    __hash__ = None


# -- the reference's per-call zero / finiteness tests, on demand ---------------------------------------------------------------
# The reference re-computes `is_zero` / `isfinite().all()` inside EVERY apply / combine_with (utils.py:497-498, flow_class.py:
# 1226-1244, 1729-1744).  Here the flag word of a tensor is cached under its version counter, which sees every in-place edit torch
# itself makes -- but not a write through a NumPy array that shares the memory (`torch.from_numpy`), through `.data`, by a foreign
# kernel, or by a c10d in-place collective on the tensor (those leave `_version` alone; `distributed.broadcast_operand` bumps it
# itself).  Two ways to tell the library that such a write happened:
#   * `Flow.invalidate()` drops the cached word of one flow;
#   * `set_revalidate_every_call(True)` makes every PUBLIC call (apply, combine_with, switch_ref, is_zero, ...) start a new epoch
#     that no cached word belongs to: each call looks at its operands again, exactly as the reference does (within a call the words
#     are still shared between the steps of that call).
_reval_on = False
_reval_epoch = 0
_reval_depth = 0


def set_revalidate_every_call(on: bool = True):
    """True: every public `Flow` method re-runs the fused validation / zero-test reduction on the flows it reads (the reference's
    behaviour, utils.py:497-498); False (default): the flag word is cached per tensor version.  Turn it on when flow storage is
    written behind torch's back (NumPy views, `.data`, custom kernels, in-place collectives), or call `Flow.invalidate()`."""
    global _reval_on, _reval_epoch
    _reval_on = bool(on)
    _reval_epoch += 1


def get_revalidate_every_call() -> bool:
    return _reval_on


def _public(fn):
    """Entry point of the public API: with `set_revalidate_every_call(True)` the outermost public call starts a new epoch."""
    import functools

    @functools.wraps(fn)
    def entry(*args, **kwargs):
        global _reval_epoch, _reval_depth
        if not _reval_on:
            return fn(*args, **kwargs)
        if _reval_depth == 0:
            _reval_epoch += 1
        _reval_depth += 1
        try:
            return fn(*args, **kwargs)
        finally:
            _reval_depth -= 1
    return entry


def _ver(t: torch.Tensor, private: bool = False):
    """Version counter of a tensor (see below; the common case is one attribute read).  Inference tensors have none (reading `_version` raises) yet CAN be edited in place inside
    `torch.inference_mode()`: nothing tells an edited one from an untouched one, so their flag words are never cached
    across calls (a FRESH key component that compares unequal to everything -- fresh because tuple comparison short-cuts on
    identity) -- UNLESS the tensor is `private`: a kernel output that only this Flow object has ever held (nobody
    else has a reference through which to edit it; ADVICE r3: inference_mode is the usual deployment mode, and re-running the
    reduction for every kernel output cost a pass and a host wait per call there)."""
    try:
        return t._version
    except RuntimeError:          # "Inference tensors do not track version counter."
        return 0 if private else _NeverEqual()


def _storage_ptr(t):
    try:
        return t.untyped_storage().data_ptr()
    except Exception:  # noqa: BLE001
        return id(t)


_private_flows = weakref.WeakSet()   # the flows whose `_private` is set: kernel outputs that are inference tensors (empty outside inference mode)


def _sharing_ends_privacy(vecs, mask):
    """`vecs` / `mask` are about to be wrapped by another flow object (a copy, a view, a relabelled or negated flow that keeps
    the mask): whoever held their storage privately no longer does -- an in-place edit through the new object must not meet
    the old one's cached flag word.  Costs nothing unless private flows exist (inference mode only)."""
    if not _private_flows:
        return
    ptrs = set(_storage_ptr(t) for t in (vecs, mask) if t is not None)
    for f in list(_private_flows):
        if _storage_ptr(f._fv) in ptrs or (f._mask is not None and _storage_ptr(f._mask) in ptrs):
            f._release_private()


class Flow(object):
    # ------------------------------------------------------------------------------------------
    # construction / properties (flow_class.py:37-236)
    # ------------------------------------------------------------------------------------------
    # -- vector storage -------------------------------------------------------------------------
    # `_vecs` is what the reference holds: N-2-H-W fp32.  A flow handed over as an fp16 tensor on a HIP device (BASELINE
    # config 5) STAYS in fp16 (`_half`): the kernels that take fp16 operands read it directly (ofl_splat_tiled_f16,
    # ofl_warp_bwd_h_f32: exact up-conversion in registers, fp32 arithmetic, the reference's `vecs.float()` of utils.py:95,118
    # without the fp32 copy), and the fp32 tensor is only made when somebody asks for it (`.vecs`, arithmetic, ...).
    @property
    def _vecs(self) -> torch.Tensor:
        if self._v32 is None:
            self._v32 = self._half.float()
        return self._v32

    @_vecs.setter
    def _vecs(self, t: torch.Tensor):
        if t.dtype == torch.float16 and t.device.type == 'cuda':
            self._half, self._v32 = t, None
        else:
            self._half, self._v32 = None, t

    @property
    def _fv(self) -> torch.Tensor:
        """The vectors as stored: fp16 if the flow was handed over (or, by option, produced) in fp16, else fp32."""
        return self._half if self._half is not None else self._v32

    def __init__(self, flow_vectors, ref: str = None, mask=None, device=None):
        self._flag_cache = None
        self._pending_flags = None
        self._mask = None
        self._vecs = get_valid_vecs(flow_vectors, error_string="Error setting flow vectors: ", _check_finite=False,
                                    _keep_half=True)
        self._device = self._fv.device
        try:
            self.ref = ref
            if mask is not None:
                m = get_valid_mask(mask, desired_shape=self.shape, error_string="Error setting flow mask: ")
                self._mask = m.to(self._fv.device)
        except (TypeError, ValueError):
            self._mask = None
            self._from_half()
            self._require_finite("Error setting flow vectors: ")   # the vecs error comes first in the reference
            raise
        self._from_half()
        _sharing_ends_privacy(self._fv, self._mask)
        self._require_finite("Error setting flow vectors: ")
        self.device = device

    def _from_half(self):
        """fp16-stored vectors on a HIP device (BASELINE config 5): the reference's `.float()` (utils.py:95,118) and the
        validation reduction run as one kernel, now that the mask is known."""
        if self._half is not None:
            self._set_pending_flags(_native.flow_flags(self._half, self._mask))     # (flags only: 5 B/px, the flow stays fp16)

    _private = False     # the vectors and the mask are kernel outputs no one else holds (see _ver); cleared when they are handed out

    @classmethod
    def _wrap(cls, vecs: torch.Tensor, ref: str, mask, device=None, flags: torch.Tensor = None,
              like: FlowAlias = None, fresh: bool = False, made_from: tuple = None, borrowed: bool = False) -> FlowAlias:
        """Internal: wrap tensors that are valid by construction (kernel outputs, views of validated flows).
        `flags` is the device-side flag word a kernel produced as a by-product, read lazily.  `fresh`: vecs and mask were
        allocated by the call that produced them; `made_from`: ... unless they share storage with one of these tensors (the
        early exits hand their inputs through).  `borrowed`: a temporary that never leaves the method that makes it (the same
        tensors under another reference): whoever holds them privately still does."""
        obj = cls.__new__(cls)
        obj._vecs, obj._ref, obj._mask = vecs, ref, mask
        obj._device = vecs.device if device is None else device
        obj._flag_cache, obj._pending_flags = None, None
        if made_from is not None:
            theirs = set(_storage_ptr(t) for t in made_from if t is not None)
            fresh = _storage_ptr(vecs) not in theirs and (mask is None or _storage_ptr(mask) not in theirs)
        if not fresh and not borrowed:
            _sharing_ends_privacy(vecs, mask)
        elif vecs.is_inference():                # (only inference tensors need it: the others carry version counters)
            obj._private = True
            _private_flows.add(obj)
        if obj._fv.device != obj._device:
            obj._vecs = obj._fv.to(obj._device)
        if obj._mask is not None and obj._mask.device != obj._device:
            obj._mask = obj._mask.to(obj._device)
        if flags is not None:
            obj._set_pending_flags(flags)
        if like is not None:
            obj._inherit_flags(like)
        return obj

    def _inherit_flags(self, src: FlowAlias):
        """`self` holds the vectors of `src` or their exact negation, under the same mask: every flag (finiteness, the
        symmetric zero / threshold tests) carries over, no reduction needed."""
        if src._flags_known() and (self._mask is src._mask):
            key = self._key()
            self._flag_cache = (key, src._flag_cache[1])

    def _negated(self, ref: str) -> FlowAlias:
        return Flow._wrap(self._vecs * -1.0, ref, self._mask, self._device, like=self)

    @classmethod
    def _deferred(cls, flow_vectors, ref: str = None, mask=None, device=None) -> FlowAlias:
        """Internal: like the public constructor (same structural checks and errors) but the finiteness test is not
        run yet -- the first kernel that reads the vectors produces the flag word as a by-product and the same
        ValueError is raised then (tensor-level wrappers: the error surfaces inside the same call)."""
        obj = cls.__new__(cls)
        obj._flag_cache, obj._pending_flags, obj._mask = None, None, None
        obj._vecs = get_valid_vecs(flow_vectors, error_string="Error setting flow vectors: ", _check_finite=False)
        obj._device = obj._vecs.device
        obj.ref = ref
        if mask is not None:
            m = get_valid_mask(mask, desired_shape=obj.shape, error_string="Error setting flow mask: ")
            obj._mask = m.to(obj._vecs.device)
        obj.device = device
        return obj

    def _key(self) -> tuple:
        """Cache key of the flag word: tensor versions (in-place edits invalidate it).  Tensors created under
        torch.inference_mode() carry no version counter: their key never matches, i.e. every call that needs the flags of
        such a flow runs the (cheap, fused) reduction again -- an in-place edit inside inference mode must not meet a
        stale 'all zero' / 'finite' word."""
        return (_ver(self._fv, self._private), None if self._mask is None else (id(self._mask), _ver(self._mask, self._private)), _reval_epoch)

    def _flags_known(self) -> bool:
        key = self._key()
        return self._flag_cache is not None and self._flag_cache[0] == key

    # -- flags: finiteness + zero tests, one fused reduction per tensor version ------------------
    def _flags(self) -> list:
        key = self._key()
        if self._flag_cache is None or self._flag_cache[0] != key:
            dev_flags = None
            if self._pending_flags is not None and self._pending_flags[0] == key:
                dev_flags = self._pending_flags[1]
            self._pending_flags = None
            sharded = distributed.is_enabled()
            if dev_flags is None and (not sharded or distributed.host_exchange_active()):
                # a tensor nobody has looked at yet (the constructor's validation): reduction + read-back in one launch.
                # Batch sharding on one node: the same launch, then the OR of the batch word over the ranks through
                # shared memory (distributed._HostExchange) -- the sharded wait is the unsharded one plus microseconds
                words = _host_flags(self._fv, self._mask)
                self._flag_cache = (key, words)
                if sharded:
                    local = 0
                    for f in words:
                        local |= f
                    self._flag_cache = (key, words, (True, distributed.reduce_flags(local, self._fv.device)))
                return self._flag_cache[1]
            if dev_flags is None:
                dev_flags = _native.flow_flags(self._fv, self._mask)
            if sharded and distributed.host_exchange_active():
                words = _flags_to_host(dev_flags)                     # (a kernel's by-product words: copy + polled event)
                local = 0
                for f in words:
                    local |= f
                self._flag_cache = (key, words, (True, distributed.reduce_flags(local, self._fv.device)))
            elif sharded:
                # batch sharding: the OR over every rank's shard is formed on the device (one small all-reduce) and read
                # together with the local words -- one host sync per tensor version, as without sharding
                words, glob = distributed.split_global_or(_flags_to_host(distributed.with_global_or(dev_flags)))
                self._flag_cache = (key, words, (True, glob))
            else:
                self._flag_cache = (key, _flags_to_host(dev_flags))
        return self._flag_cache[1]

    def _set_pending_flags(self, dev_flags):
        if dev_flags is not None:
            key = self._key()
            self._pending_flags = (key, dev_flags)

    def _batch_flags(self) -> int:
        """OR of the flag words over the whole batch -- over every rank's shard when batch sharding is on
        (the reference's early exits are batch-global, distributed.py)."""
        flags = self._flags()
        if len(self._flag_cache) < 3 or self._flag_cache[2][0] != distributed.is_enabled():
            local = 0
            for f in flags:
                local |= f
            glob = distributed.reduce_flags(local, self._fv.device)
            self._flag_cache = (self._flag_cache[0], self._flag_cache[1], (distributed.is_enabled(), glob))
        return self._flag_cache[2][1]

    def invalidate(self) -> FlowAlias:
        """Forget what is known about the vectors (finite / all zero / below the threshold, plain and under the mask): the next call
        that needs it looks again.  For writes torch's version counter does not see -- a NumPy array sharing the memory
        (`torch.from_numpy`), `tensor.data`, a custom kernel, an in-place collective on the tensor.  The reference needs no such call
        because it runs the tests inside every apply / combine_with (utils.py:497-498, flow_class.py:1226-1244); see also
        `set_revalidate_every_call`.  Returns the flow itself."""
        self._flag_cache, self._pending_flags = None, None
        return self

    def _require_finite(self, error_string: str):
        if self._batch_flags() & _native.FLAG_NONFINITE:                              # utils.py:98
            raise ValueError(error_string + "Input contains NaN, Inf or -Inf values")

    def _all_zero(self, bit: int) -> bool:
        return not (self._batch_flags() & bit)

    @property
    def vecs(self) -> torch.Tensor:
        """Flow vectors N-2-H-W, fp32 (flow_class.py:68-80).  As in the reference the tensor handed out IS the flow's
        storage: an in-place edit of it edits the flow.  A flow kept in fp16 (`_half`) therefore gives up its fp16 planes
        the moment its fp32 tensor is handed out -- from then on there is one store, the fp32 one, and the kernels read
        that (the flag word carries over: the up-conversion is exact; a later in-place edit bumps the tensor's version and
        invalidates it like any other)."""
        v = self._vecs
        self._release_private()
        if self._half is not None:
            flags = self._flag_cache[1:] if self._flags_known() else None
            pending = self._pending_flags[1] if (self._pending_flags is not None and self._pending_flags[0] == self._key()) else None
            self._half = None
            self._flag_cache = None if flags is None else (self._key(),) + tuple(flags)
            self._pending_flags = None if pending is None else (self._key(), pending)
        return v

    @vecs.setter
    def vecs(self, input_vecs):
        v = get_valid_vecs(input_vecs, error_string="Error setting flow vectors: ", _check_finite=False)
        old = self._vecs
        # the caller's tensor may be stored as it is (an fp32 tensor on the device passes through get_valid_vecs): a flow
        # that was private to its object is not any more, so an inference tensor's key must stop matching
        self._release_private()
        self._vecs, self._flag_cache, self._pending_flags = v, None, None
        try:
            self._require_finite("Error setting flow vectors: ")
        except ValueError:
            self._vecs, self._flag_cache = old, None
            raise

    @property
    def vecs_numpy(self) -> np.ndarray:
        """N-H-W-2 float32 numpy view of the vectors (flow_class.py:95-110)"""
        self._release_private()               # (a CPU-resident flow hands out shared memory here)
        return np.moveaxis(self._vecs.detach().cpu().numpy(), 1, -1)

    @property
    def ref(self) -> str:
        """'s' (source) or 't' (target) reference (flow_class.py:112-148)"""
        return self._ref

    @ref.setter
    def ref(self, input_ref: str = None):
        self._ref = get_valid_ref(input_ref)

    @property
    def mask(self) -> torch.Tensor:
        """Validity mask N-H-W bool (flow_class.py:159-172).  An all-True default is materialised on first use."""
        if self._mask is None:
            self._mask = torch.ones(self.shape, dtype=torch.bool, device=self._fv.device)
            self._flag_cache = None if self._flag_cache is None else (self._key(), self._flag_cache[1])
        self._release_private()
        return self._mask

    def _release_private(self):
        """The storage is about to be handed out: from now on somebody else can edit it in place.  Tensors that track versions
        keep their key (an edit bumps it); an inference tensor's key stops matching, so its next use runs the reduction again."""
        if self._private:
            self._private = False
            _private_flows.discard(self)

    @mask.setter
    def mask(self, input_mask=None):
        self._release_private()               # (the caller's mask may be stored as it is: see the vecs setter)
        if input_mask is None:
            self._mask = None
        else:
            m = get_valid_mask(input_mask, desired_shape=self.shape, error_string="Error setting flow mask: ")
            self._mask = m.to(self._fv.device)
        self._flag_cache, self._pending_flags = None, None

    @property
    def mask_numpy(self) -> np.ndarray:
        """flow_class.py:188-205"""
        return self.mask.detach().cpu().numpy()

    @property
    def device(self) -> torch.device:
        """flow_class.py:207-213"""
        return self._device

    @device.setter
    def device(self, input_device=None):
        device = self._fv.device if input_device is None else get_valid_device(input_device)
        self._device = device
        if self._fv.device != device or (self._mask is not None and self._mask.device != device):
            self._release_private()           # storage is replaced: the privacy of the old tensors says nothing about the new ones
        if self._fv.device != device:
            self._vecs = self._fv.to(device) if device.type == 'cuda' else self._vecs.to(device)
            self._flag_cache = None if self._flag_cache is None else \
                ((_ver(self._fv),) + tuple(self._flag_cache[0][1:]), self._flag_cache[1])
        if self._mask is not None and self._mask.device != device:
            flags = None if self._flag_cache is None else self._flag_cache[1]
            self._mask = self._mask.to(device)
            if flags is not None:
                self._flag_cache = (self._key(), flags)

    @property
    def shape(self) -> tuple:
        """(N, H, W) (flow_class.py:228-236)"""
        return (self._fv.shape[0],) + tuple(self._fv.shape[2:])

    @classmethod
    def zero(cls, shape, ref: str = None, mask=None, device=None) -> FlowAlias:
        """All-zero flow of shape (H, W) or (N, H, W) (flow_class.py:238-258)"""
        dims = get_valid_shape(shape)
        return cls(torch.zeros(dims[0], 2, dims[1], dims[2]), ref, mask, device)

    @classmethod
    def from_matrix(cls, matrix, shape, ref: str = None, mask=None, device=None, matrix_is_inverse: bool = None) -> FlowAlias:
        """flow_class.py:260-292"""
        device = get_valid_device(device) if device is not None else None
        on = device if (device is not None and device.type == 'cuda') else None        # generate the field where it is wanted
        return cls(from_matrix(matrix, shape, ref, matrix_is_inverse, _device=on), ref, mask, device)

    @classmethod
    def from_transforms(cls, transform_list: list, shape, ref: str = None, mask=None, device=None,
                        padding: list = None) -> FlowAlias:
        """flow_class.py:294-328"""
        device = get_valid_device(device) if device is not None else None
        on = device if (device is not None and device.type == 'cuda') else None
        return cls(from_transforms(transform_list, shape, ref, padding, _device=on), ref, mask, device)

    # ------------------------------------------------------------------------------------------
    # copies, indexing (flow_class.py:376-448)
    # ------------------------------------------------------------------------------------------
    def copy(self) -> FlowAlias:
        """flow_class.py:376-383.  The copy aliases the same (already validated) tensors: its flag word is inherited
        instead of recomputed (no launch, no host sync)."""
        return Flow._wrap(self._fv, self._ref, self._mask, self._device, like=self)

    def to_device(self, device) -> FlowAlias:
        device = get_valid_device(device)
        return Flow(self._vecs.to(device), self._ref, None if self._mask is None else self._mask.to(device), device)

    def __str__(self) -> str:
        return "Flow object, reference {}, batch size {}, shape {}*{}, device {}; ".format(
            self._ref, *self.shape, self._device) + self.__repr__()

    def select(self, item: int = None) -> FlowAlias:
        if item is None:
            return self
        if not isinstance(item, int):
            raise TypeError("Error selecting from flow object: item needs to be an integer")
        try:
            return self._subset(self._vecs[item], self.mask[item])
        except IndexError:
            raise IndexError("Error selecting from flow object: item {} out of bounds for flow with batch size {}"
                             .format(item, self.shape[0]))

    def __getitem__(self, item) -> FlowAlias:
        # index H, W (then N, 2) the way a tensor of shape H-W-N-2 would be indexed (flow_class.py:433-448)
        vecs = self._vecs.permute(2, 3, 0, 1).__getitem__(item).permute(2, 3, 0, 1)
        mask = self.mask.permute(1, 2, 0).__getitem__(item).permute(2, 0, 1)
        return self._subset(vecs, mask)

    def _subset(self, vecs: torch.Tensor, mask: torch.Tensor) -> FlowAlias:
        """A slice of this (validated) flow as a flow object: the constructor's shape checks (flow_class.py:60-99) without its
        finiteness reduction -- a subset of finite values is finite (no kernel launch, no host sync)."""
        v = get_valid_vecs(vecs, error_string="Error setting flow vectors: ", _check_finite=False)
        m = get_valid_mask(mask, desired_shape=(v.shape[0], v.shape[2], v.shape[3]), error_string="Error setting flow mask: ")
        return Flow._wrap(v, self._ref, m, self._device)

    # ------------------------------------------------------------------------------------------
    # arithmetic (flow_class.py:450-692)
    # ------------------------------------------------------------------------------------------
    def _binary_operand(self, other, verb_ing: str, names: tuple):
        if isinstance(other, (np.ndarray, torch.Tensor)):
            ov = get_valid_vecs(other, desired_shape=self.shape, error_string="Error adding to flow: ")
            om = None
        elif isinstance(other, Flow):
            ov, om = other._vecs, other._mask
        else:
            raise TypeError("Error {}: {} is not a flow object, numpy array, or torch tensor".format(verb_ing, names[0]))
        if self.shape[0] != ov.shape[0] and self.shape[0] != 1 and ov.shape[0] != 1:
            raise ValueError("Error {}: {} batch dimensions don't match, and neither is 1".format(verb_ing, names[1]))
        if self.shape[1:] != tuple(ov.shape[2:]):
            raise ValueError("Error {}: {} flow objects are not the same shape".format(verb_ing, names[1]))
        return ov.to(self._device), (None if om is None else om.to(self._device))

    def _and_masks(self, other_mask):
        if self._mask is None and other_mask is None:
            return None
        if self._mask is None:
            return other_mask if other_mask.shape[0] >= self.shape[0] else other_mask.expand(self.shape).clone()
        if other_mask is None:
            return self._mask
        return self._mask & other_mask

    def __add__(self, other) -> FlowAlias:
        """Vector sum, masks ANDed -- not a composition (flow_class.py:450-488)"""
        ov, om = self._binary_operand(other, "adding to flow", ("Addend", "Augend and addend"))
        vecs = self._vecs + ov
        mask = self._and_masks(om)
        if mask is not None and mask.shape[0] != vecs.shape[0]:
            mask = mask.expand(vecs.shape[0], -1, -1)
        return Flow._wrap(vecs, self._ref, mask, self._device)

    def __sub__(self, other) -> FlowAlias:
        """flow_class.py:490-531"""
        ov, om = self._binary_operand(other, "subtracting from flow", ("Subtrahend", "Minuend and subtrahend"))
        vecs = self._vecs - ov
        mask = self._and_masks(om)
        if mask is not None and mask.shape[0] != vecs.shape[0]:
            mask = mask.expand(vecs.shape[0], -1, -1)
        return Flow._wrap(vecs, self._ref, mask, self._device)

    def _scalar_or_field(self, other, verb_ing: str, noun: str):
        """Operand of * / **: number, list of 2, or array of shape 2, H-W, 2-H-W, H-W-2, N-2-H-W."""
        try:
            return float(other)
        except (TypeError, ValueError):
            pass
        if isinstance(other, list):
            if len(other) != 2:
                raise ValueError("Error {} flow: {} list not length 2".format(verb_ing, noun))
            other = torch.tensor(other)
        elif isinstance(other, np.ndarray):
            other = torch.tensor(other)
        if not isinstance(other, torch.Tensor):
            raise TypeError("Error {} flow: {} cannot be converted to float, or isn't a list, numpy array, or torch "
                            "tensor".format(verb_ing, noun))
        hw = self.shape[1:]
        if other.dim() == 1 and other.shape[0] == 2:
            other = other.view(1, 2, 1, 1)
        elif other.dim() == 2 and tuple(other.shape) == hw:
            other = other.view(1, 1, *hw)
        elif other.dim() == 3 and tuple(other.shape) == (2,) + hw:
            other = other.unsqueeze(0)
        elif other.dim() == 3 and tuple(other.shape) == hw + (2,):
            other = move_axis(other, -1, 0).unsqueeze(0)
        elif other.dim() == 4 and tuple(other.shape[2:]) == hw and other.shape[1] == 2 and \
                (self.shape[0] == 1 or other.shape[0] == 1 or self.shape[0] == other.shape[0]):
            pass
        else:
            raise ValueError("Error {} flow: {} array or tensor needs to be of size 2, of the shape of the flow object "
                             "(H-W), or 2-H-W or H-W-2, or N-2-H-W".format(verb_ing, noun))
        return other.to(self._device)

    def _result_of(self, vecs) -> FlowAlias:
        mask = self._mask
        if mask is not None and vecs.shape[0] != self.shape[0]:
            mask = mask.repeat(vecs.shape[0], 1, 1)
        return Flow(vecs, self._ref, mask, self._device)

    def __mul__(self, other) -> FlowAlias:
        """flow_class.py:533-580"""
        o = self._scalar_or_field(other, "multiplying", "Multiplier")
        if isinstance(o, float):
            if o == -1.0 or o == 1.0:
                return Flow._wrap(self._vecs * o, self._ref, self._mask, self._device, like=self)
            return Flow._wrap(self._vecs * o, self._ref, self._mask, self._device) if np.isfinite(o) \
                else Flow(self._vecs * o, self._ref, self._mask, self._device)
        return self._result_of(self._vecs * o)

    def __truediv__(self, other) -> FlowAlias:
        """flow_class.py:582-629.  A Python-number divisor is handed to ATen as a DEVICE scalar tensor: with a host scalar
        ATen's GPU kernel multiplies by the reciprocal instead of dividing, which differs from the reference's (CPU) quotient
        in the last bit for a third of the values; tensor / tensor is the correctly rounded division on either device."""
        o = self._scalar_or_field(other, "dividing", "Divisor")
        if isinstance(o, float) and self._vecs.device.type == 'cuda':
            o = torch.tensor(o, dtype=self._vecs.dtype, device=self._vecs.device)
        return self._result_of(self._vecs / o)

    def __pow__(self, other) -> FlowAlias:
        """flow_class.py:631-678"""
        return self._result_of(self._vecs ** self._scalar_or_field(other, "exponentiating", "Exponent"))

    def __neg__(self) -> FlowAlias:
        """flow_class.py:680-692"""
        return self * -1

    # ------------------------------------------------------------------------------------------
    # resize / pad (flow_class.py:694-753) -- thin PyTorch wrappers, not on the kernel path
    # ------------------------------------------------------------------------------------------
    def resize(self, scale) -> FlowAlias:
        """flow_class.py:694-714: bilinear interpolation (`align_corners=False`) of the vectors (scaled along) and of the mask
        (then rounded).  On a HIP device `ofl_resize_bilinear_f32` restates the arithmetic of ATen's CPU kernels, so the result is
        the reference's PyTorch-CPU result bit for bit; on the host it is ATen itself.  The mask goes through the reference's own
        `.squeeze(0).squeeze(0)`, so -- exactly like the reference -- a batch of more than one flow raises the mask's shape error."""
        resized = resize_flow(self._vecs, scale)
        sc = [scale, scale] if isinstance(scale, (float, int)) else scale
        m = interpolate_bilinear(self.mask.float().unsqueeze(1), sc).squeeze(0).squeeze(0)
        return Flow(resized, self._ref, torch.round(m), device=self._device)

    def pad(self, padding: list = None, mode: str = None) -> FlowAlias:
        mode = 'constant' if mode is None else mode
        if mode not in ('constant', 'reflect', 'replicate'):
            raise ValueError("Error padding flow: Mode should be one of "
                             "'constant', 'reflect', or 'replicate', but instead got '{}'".format(mode))
        padding = get_valid_padding(padding, "Error padding flow: ")
        lrtb = (padding[2], padding[3], padding[0], padding[1])
        return Flow._wrap(F.pad(self._vecs, lrtb, mode=mode), self._ref,
                          F.pad(self.mask.unsqueeze(1), lrtb).squeeze(1), self._device)

    def unpad(self, padding: list = None) -> FlowAlias:
        padding = get_valid_padding(padding, "Error padding flow: ")
        h, w = self.shape[1:3]
        if sum(padding[0:2]) > h - 1 or sum(padding[2:4]) > w - 1:
            raise ValueError("Error unpadding flow: one or more dimensions cut to zero or less")
        return self[padding[0]:h - padding[1], padding[2]:w - padding[3]]

    # ------------------------------------------------------------------------------------------
    # apply (flow_class.py:755-959)
    # ------------------------------------------------------------------------------------------
    def apply(self, target, target_mask: torch.Tensor = None, return_valid_area: bool = None,
              consider_mask: bool = None, padding: list = None, cut: bool = None):
        """Warp `target` (tensor H-W / C-H-W / N-C-H-W, or a Flow) with this flow.  One fused kernel launch: the
        target's mask rides along as an extra channel, is thresholded at 0.99999 and ANDed with the flow mask in
        the same pass (reference: cat + apply_flow + gt + and, flow_class.py:896-934)."""
        return_valid_area = False if return_valid_area is None else return_valid_area
        if not isinstance(return_valid_area, bool):
            raise TypeError("Error applying flow: Return_valid_area needs to be a boolean")
        consider_mask = True if consider_mask is None else consider_mask
        if not isinstance(consider_mask, bool):
            raise TypeError("Error applying flow: Consider_mask needs to be a boolean")
        cut = True if cut is None else cut
        if not isinstance(cut, bool):
            raise TypeError("Error applying flow: Cut needs to be a boolean")
        if padding is not None:
            padding = get_valid_padding(padding, "Error applying flow: ")
            if self.shape[1] + padding[0] + padding[1] != target.shape[-2] or \
                    self.shape[2] + padding[2] + padding[3] != target.shape[-1]:
                raise ValueError("Error applying flow: Padding values do not match flow and target shape difference")

        return_dtype, return_2d, return_3d = torch.float, False, False
        if isinstance(target, Flow):
            return_flow = True
            t, tmask = target._fv, target._mask
        elif isinstance(target, torch.Tensor):
            return_flow = False
            if target.dim() == 4:
                t = target
            elif target.dim() == 3:
                t, return_3d = target.unsqueeze(0), True
            elif target.dim() == 2:
                t, return_2d = target.unsqueeze(0).unsqueeze(0), True
            else:
                raise ValueError("Error applying flow: Target needs to have the shape H-W (2 dimensions)"
                                 ", C-H-W (3 dimensions), or N-C-H-W (4 dimensions)")
            tmask = None
            if target_mask is not None:
                if not isinstance(target_mask, torch.Tensor):
                    raise TypeError("Error applying flow: Target_mask needs to be a torch tensor")
                if target_mask.dim() == 2:
                    target_mask = target_mask.unsqueeze(0)
                if tuple(target_mask.shape) != (t.shape[0],) + tuple(t.shape[2:]):
                    raise ValueError("Error applying flow: Target_mask needs to match the target shape")
                if target_mask.dtype != torch.bool:
                    raise TypeError("Error applying flow: Target_mask needs to have dtype 'bool'")
                if not return_valid_area:
                    warnings.warn("Warning applying flow: a mask is passed, but return_valid_area is False - so the "
                                  "mask passed will not affect the output, but possibly make the function slower.")
                tmask = target_mask
            return_dtype = target.dtype
        else:
            raise TypeError("Error applying flow: Target needs to be either a flow object or a torch tensor")
        need_valid = return_flow or return_valid_area

        rm = _native.ROUND_NONE
        if not return_flow and not return_dtype.is_floating_point:
            rm = _native.ROUND_U8 if return_dtype == torch.uint8 else _native.ROUND_RINT
        if padding is None:
            if tuple(target.shape[-2:]) != self.shape[-2:]:
                raise ValueError("Error applying flow: Flow and target have to have the same shape")
            warped, valid, dflags = self._warp(t, tmask, need_valid, consider_mask, rm)
        else:
            warped, valid, dflags = self._warp_padded(t, tmask, need_valid, consider_mask, rm, padding)

        if padding is not None and not cut and need_valid and self._ref == 't' and warped.shape[0] < self.shape[0]:
            # an all-zero 't' flow of batch N, padded and NOT cut, over a batch-1 target: the reference's mask is still batch 1 (the target
            # went through apply_flow untouched, utils.py:497-498) when it assigns the batch-N `tmp & self._mask` into its flow-area
            # window (flow_class.py:929-932) -- torch refuses, with this message (found by tests/golden/fuzz_vs_reference.py)
            raise RuntimeError("The expanded size of the tensor (%d) must match the existing size (%d) at non-singleton dimension 0.  "
                               "Target sizes: [%d, %d, %d].  Tensor sizes: [%d, %d, %d]"
                               % (warped.shape[0], self.shape[0], warped.shape[0], self.shape[1], self.shape[2], self.shape[0], self.shape[1], self.shape[2]))
        if padding is not None and cut:
            win = (slice(padding[0], padding[0] + self.shape[1]), slice(padding[2], padding[2] + self.shape[2]))
            warped = warped[..., win[0], win[1]]
            if valid is not None:
                valid = valid[..., win[0], win[1]]
            dflags = None                                # (they describe the whole padded frame)

        if return_flow:
            if valid is not None and valid.shape[0] != warped.shape[0]:
                # an all-zero 't' flow of batch N over a batch-1 Flow target: apply_flow hands the batch-1 target through
                # (utils.py:497-498) while the mask is ANDed with this flow's N masks (:934) -- the reference's Flow(...) of the two
                # (:938) then fails in its mask setter, with this message (found by tests/golden/fuzz_vs_reference.py)
                raise ValueError("Error setting flow mask: Input shape does not match the desired shape")
            return Flow._wrap(warped, target._ref, valid, self._device, flags=dflags, made_from=(t, tmask, self._mask))
        if not return_dtype.is_floating_point:
            if not get_pure_pytorch():
                warped = warped.to(return_dtype)         # PURE_PYTORCH keeps the rounded values as floats (:943-949)
        else:
            warped = warped.to(return_dtype)
        if (return_2d or return_3d) and warped.shape[0] == 1:
            warped = warped[0, 0] if return_2d else warped[0]
        return (warped, valid) if return_valid_area else warped

    def _warp_padded(self, t: torch.Tensor, tmask, need_valid: bool, consider_mask: bool, round_mode: int, padding: list):
        """`apply` with a target larger than the flow by `padding` (flow_class.py:830-834, 880-895, 901-913, 924-932): the
        reference pads the flow to the target's size first -- zeros for 't', replicated border for 's', mask False -- and
        warps the whole frame.  Here the kernels read the un-padded flow through a window (`ofl_warp_bwd_win_f32`,
        `ofl_splat_tiled_win_f32`): no padded copy of the flow or its mask is made.  The all-zero early exit, narrow frames
        the gather kernels do not take, and tensors that want a gradient go through the padded copy (rare / plumbing)."""
        window = (padding[0], padding[2])
        # The reference's padded 's' branch walks the TARGET's batch (flow_class.py:884-894: `for i in range(mask.shape[0])`): with a
        # batch-1 target under a batch-N flow it ANDs element 0's flow mask -- and only that one -- into the mask channel, which every
        # element of the flow then warps; and because `mask` keeps batch 1 the target is not expanded (:896-897), so an all-zero flow
        # hands the batch-1 target (and that batch-1 mask) straight through (utils.py:497-498).  Reproduced: results are what the
        # reference returns on the same inputs, quirk included (fixtures tests/golden/padbc.npz).
        quirk = self._ref == 's' and need_valid and t.shape[0] < self.shape[0]
        chan_b = self._mask if not quirk else self.mask[:1]
        direct = not self._all_zero(_native.FLAG_NZ_THR) and \
            not _native._wants_grad(self._vecs, t) and (self._ref == 't' or get_pure_pytorch())
        if direct:
            self._require_finite("Error applying flow to a target: ")
            if self._ref == 't':
                warped, valid = _native.warp_bwd_win(self._vecs, t, window, src_mask=tmask,
                                                     flow_mask=self._mask if need_valid else None, want_valid=need_valid,
                                                     round_mode=round_mode)
                return warped.to(self._device), (None if valid is None else valid.to(self._device)), None
            res = _native.splat_fwd_win(self._vecs, t, window, weight_mask=self.mask if consider_mask else None,
                                        chan_mask_a=tmask, chan_mask_b=chan_b, want_valid=need_valid, occlude=True,
                                        round_mode=round_mode)
            if res is not None:
                return res[0].to(self._device), (None if res[1] is None else res[1].to(self._device)), None
        # 't': zero padding is irrelevant outside the flow area; 's': replicate avoids artefacts at the border of the flow area
        # (flow_class.py:906-913).  The padded mask is False, so the un-padded formulas hold for the padded flow as they stand.
        flow = self.pad(padding, mode='constant' if self._ref == 't' else 'replicate')
        if quirk:
            return flow._warp(t, tmask, need_valid, consider_mask, round_mode, mask_chan_b=flow.mask[:1])
        return flow._warp(t, tmask, need_valid, consider_mask, round_mode)

    def _warp(self, t: torch.Tensor, tmask, need_valid: bool, consider_mask: bool, round_mode: int = 0,
              flow_sign: float = 1.0, data_sign: float = 1.0, t_minus: torch.Tensor = None, mask_chan_b: torch.Tensor = None):
        """Core of `apply`: t [Nt,C,H,W] any dtype, tmask [Nt,H,W] bool or None (all True).
        Returns (warped fp32 [N,C,H,W], valid bool [N,H,W] | None) on self.device.
        `flow_sign` / `data_sign` = -1 (forward flows only) restate `(-self)` as the warper / `-t` as the target inside the
        kernel (exact negations; finiteness and zero tests do not depend on the sign), so that switch_ref / invert need
        no negated copy and no second flag reduction.  `t_minus` (past the early exit only): the target is t - t_minus,
        subtracted inside the kernel (modes 1 't' / 2 's': target `flow - self`; `tmask` / the two channel masks carry
        m_flow & m_self).  `mask_chan_b` ('s' only; the padded batch-1-target case of `_warp_padded`): the batch-1 flow mask
        that goes into the mask channel instead of this flow's own, and keeps an all-zero flow's result at batch 1."""
        if self._ref == 's' and not get_pure_pytorch():
            _griddata_unavailable("Flow.apply(ref='s')")
        batch_flags = self._batch_flags()                                             # (one look at the cached word for both tests)
        if batch_flags & _native.FLAG_NONFINITE:                                      # utils.py:98
            raise ValueError("Error applying flow to a target: Input contains NaN, Inf or -Inf values")
        if not (batch_flags & _native.FLAG_NZ_THR):
            # apply_flow's early exit (utils.py:497-498): every |component| < 1e-3 -> the target (and its mask
            # channel) pass through unchanged; batch broadcasting as in flow_class.py:895-898, 922-934
            warped = t.to(torch.float).to(self._device)
            if data_sign != 1.0:
                warped = warped * data_sign
            valid = None
            if need_valid:
                valid = torch.ones((t.shape[0],) + tuple(t.shape[2:]), dtype=torch.bool, device=self._device) \
                    if tmask is None else tmask.to(self._device)
                if self._ref == 's' and mask_chan_b is not None:      # padded, batch-1 target: element 0's mask only, batch stays 1 (:884-897)
                    valid = valid & mask_chan_b.to(self._device)
                elif self._ref == 's':                                # mask & self._mask before the warp (:895)
                    valid = valid & self.mask if self._mask is not None or valid.shape[0] < self.shape[0] else valid
                if valid.shape[0] != warped.shape[0]:                 # :896-897
                    warped = warped.expand(valid.shape[0], -1, -1, -1)
                if self._ref == 't':                                  # ... or after it (:934)
                    valid = valid & self.mask if self._mask is not None or valid.shape[0] < self.shape[0] else valid
            if round_mode:
                warped = torch.round(warped)
                if round_mode == _native.ROUND_U8:
                    warped = torch.clamp(warped, 0, 255)
            return warped, valid, None
        dflags = None
        if self._ref == 't':
            warped, valid, _, _ = _native.warp_bwd(self._fv, t, src_mask=tmask,
                                                   flow_mask=self._mask if need_valid else None,
                                                   want_valid=need_valid, round_mode=round_mode, src_b=t_minus,
                                                   out_uint8=not get_pure_pytorch())   # (:943-949: only then is it cast back)
        else:
            # a warped FLOW (2 channels, with its valid mask) brings its flag word along: the splat produces it as a
            # by-product, so that using the result as a warper needs no validation pass of its own
            want_f = need_valid and t.shape[1] == 2 and round_mode == 0
            res = _native.splat_fwd(self._fv, t, weight_mask=self._mask if consider_mask else None,
                                    chan_mask_a=tmask, chan_mask_b=self._mask if mask_chan_b is None else mask_chan_b,
                                    want_valid=need_valid, occlude=True, round_mode=round_mode,
                                    flow_sign=flow_sign, data_sign=data_sign, want_dst_flags=want_f, data_b=t_minus,
                                    out_half=get_half_flow_outputs())
            warped, valid = res[0], res[1]
            dflags = res[4] if want_f else None
        return warped.to(self._device), (None if valid is None else valid.to(self._device)), dflags

    # ------------------------------------------------------------------------------------------
    # track (flow_class.py:961-1020)
    # ------------------------------------------------------------------------------------------
    def track(self, pts: torch.Tensor, int_out: bool = None, get_valid_status: bool = None):
        """Warp points (y, x) of shape M-2 or N-M-2; optionally the status of each point = `valid_source()` at its
        (rounded) position.  Differentiable wrt the flow vectors and the points."""
        input_2d = pts.dim() == 2
        get_valid_status = False if get_valid_status is None else get_valid_status
        if not isinstance(get_valid_status, bool):
            raise TypeError("Error tracking points: Get_tracked needs to be a boolean")
        warped_pts = track_pts(flow=self._vecs, ref=self._ref, pts=pts, int_out=int_out)
        if input_2d and self.shape[0] == 1:
            warped_pts = warped_pts.squeeze(0)
        if not get_valid_status:
            return warped_pts
        if pts.dtype.is_floating_point:
            pts = torch.round(pts)
        pts2 = pts.unsqueeze(0) if input_2d else pts
        if pts2.shape[0] != self.shape[0]:
            pts2 = pts2.expand(self.shape[0], -1, -1)
        valid_source = self.valid_source().view(self.shape[0], -1)
        pts2 = (pts2[..., 0] * self.shape[-1] + pts2[..., 1]).to(valid_source.device)
        status_array = torch.gather(valid_source, 1, pts2.long())
        if warped_pts.dim() == 2:
            status_array = status_array.squeeze(0)
        return warped_pts, status_array

    # ------------------------------------------------------------------------------------------
    # switch_ref / invert (flow_class.py:1022-1086)
    # ------------------------------------------------------------------------------------------
    def switch_ref(self, mode: str = None) -> FlowAlias:
        mode = 'valid' if mode is None else mode
        if mode == 'invalid':
            return Flow._wrap(self._fv, 't' if self._ref == 's' else 's', self._mask, self._device, like=self)
        if mode != 'valid':
            raise ValueError("Error switching flow reference: Mode not recognised, should be 'valid' or 'invalid'")
        if self._all_zero(_native.FLAG_NZ_MASKED):                                   # flow_class.py:1046
            return self.switch_ref(mode='invalid')
        if self._ref == 's':
            out = self.apply(self)                                                    # one splat: P(f, f||[m], m)
            out._ref = 't'
            return out
        # (-as_s).apply(as_s) with as_s = this flow read as 's' (flow_class.py:1060-1062): one splat P(-f, f||[m], m),
        # the negation folded into the kernel's end points
        as_s = Flow._wrap(self._fv, 's', self._mask, self._device, like=self, borrowed=True)
        warped, valid, dflags = as_s._warp(self._fv, self._mask, True, True, flow_sign=-1.0)
        return Flow._wrap(warped, 's', valid, self._device, flags=dflags, made_from=(self._fv, self._mask))

    def invert(self, ref: str = None) -> FlowAlias:
        ref = self._ref if ref is None else get_valid_ref(ref)
        if self._ref == 's':
            if ref == 's':                                      # self.apply(-self): P(f, -f||[m], m)
                warped, valid, dflags = self._warp(self._fv, self._mask, True, True, data_sign=-1.0)
                return Flow._wrap(warped, 's', valid, self._device, flags=dflags, made_from=(self._fv, self._mask))
            return self._negated('t')
        if ref == 's':
            return self._negated('s')
        # self.invert('s').switch_ref(): with g = -f read as 's', g.apply(g) = P(-f, -f||[m], m)   (flow_class.py:1084-1086)
        if self._all_zero(_native.FLAG_NZ_MASKED):              # switch_ref's early exit (:1046) on g
            return self._negated('t')
        warped, valid, dflags = Flow._wrap(self._fv, 's', self._mask, self._device, like=self, borrowed=True)._warp(
            self._fv, self._mask, True, True, flow_sign=-1.0, data_sign=-1.0)
        return Flow._wrap(warped, 't', valid, self._device, flags=dflags, made_from=(self._fv, self._mask))

    # ------------------------------------------------------------------------------------------
    # valid areas (flow_class.py:1088-1172)
    # ------------------------------------------------------------------------------------------
    def valid_target(self, consider_mask: bool = None) -> torch.Tensor:
        consider_mask = True if consider_mask is None else consider_mask
        if not isinstance(consider_mask, bool):
            raise TypeError("Error applying flow: Consider_mask needs to be a boolean")
        if self._ref == 's':
            return self._splat_mask_is_one(1.0, consider_mask)
        return self._warped_ones_valid(1.0)

    def valid_source(self, consider_mask: bool = None) -> torch.Tensor:
        consider_mask = True if consider_mask is None else consider_mask
        if not isinstance(consider_mask, bool):
            raise TypeError("Error applying flow: Consider_mask needs to be a boolean")
        if self._ref == 's':
            return self._warped_ones_valid(-1.0)
        return self._splat_mask_is_one(-1.0, consider_mask)

    def _warped_ones_valid(self, sign: float) -> torch.Tensor:
        """(apply_flow(sign * vecs, ones, 't') > 0.9999) & mask (flow_class.py:1119-1122, 1151-1157) as ONE launch that reads the
        flow and its mask and writes the area (`ofl_warp_valid_f32`): no all-ones image, no warped copy of it, no compare and
        AND passes."""
        self._require_finite("Error applying flow to a target: ")
        if self._all_zero(_native.FLAG_NZ_THR):                      # apply_flow's early exit (utils.py:497-498): ones > 0.9999
            return self.mask.clone()
        return _native.warp_valid(self._vecs, self._mask, sign, 0.9999).to(self._device)

    def _splat_mask_is_one(self, sign: float, consider_mask: bool) -> torch.Tensor:
        """apply_flow(sign * vecs, mask.float(), 's', mask if consider_mask else None) == 1 (flow_class.py:1116-1118,
        1165-1169): the flow mask splatted as data, through the kernels' mask channel so that "every contributor
        valid" is exactly 1 whatever order the accumulation ran in."""
        if not get_pure_pytorch():
            _griddata_unavailable("valid_target / valid_source")
        self._require_finite("Error applying flow to a target: ")
        if self._all_zero(_native.FLAG_NZ_THR):                      # apply_flow's early exit: the mask itself
            return self.mask.clone()
        dummy = self._vecs[:, :1]
        _, mch, _, _ = _native.splat_fwd(self._vecs, dummy, flow_sign=sign, chan_mask_a=self._mask,
                                         weight_mask=self._mask if consider_mask else None, occlude=True,
                                         want_mask_chan=True)
        return (mch == 1).to(self._device)

    # ------------------------------------------------------------------------------------------
    # padding needed (flow_class.py:1174-1224)
    # ------------------------------------------------------------------------------------------
    def get_padding(self, item: int = None) -> list:
        """[top, bottom, left, right] (per batch element, or of element `item`) by which a source image ('t') or the
        flow itself ('s') must be padded so that no valid vector reaches outside.  One masked min / max reduction
        kernel (`ofl_flow_extents_f32`) over the thresholded positions instead of clone + 4 elementwise passes + 4
        boolean-indexed reductions per batch element."""
        flow = self.select(item=item)
        ext = _native.flow_extents(flow._vecs, flow._mask, 1.0 if flow._ref == 't' else -1.0).cpu().tolist()
        h, w = flow.shape[1:]
        padding = []
        for lo_y, hi_y, lo_x, hi_x, any_valid in ext:
            if not any_valid:        # torch.min of an empty selection raises in the reference as well
                raise RuntimeError("Error getting padding: the flow mask is False everywhere")
            pad = [max(-lo_y, 0), max(hi_y - (h - 1), 0), max(-lo_x, 0), max(hi_x - (w - 1), 0)]
            padding.append([int(np.ceil(p)) for p in pad])
        return padding[0] if item is not None else padding

    # ------------------------------------------------------------------------------------------
    # zero test (flow_class.py:1226-1244)
    # ------------------------------------------------------------------------------------------
    def is_zero(self, thresholded: bool = None, masked: bool = None) -> torch.Tensor:
        masked = True if masked is None else masked
        if not isinstance(masked, bool):
            raise TypeError("Error checking whether flow is zero: Masked needs to be a boolean")
        thresholded = True if thresholded is None else thresholded
        if not isinstance(thresholded, bool):
            raise TypeError("Error checking whether flow is zero: Thresholded needs to be a boolean")
        if masked:
            bit = _native.FLAG_NZ_THR_MASKED if thresholded else _native.FLAG_NZ_MASKED
        else:
            bit = _native.FLAG_NZ_THR if thresholded else _native.FLAG_NZ
        return torch.tensor([(f & bit) == 0 for f in self._flags()], dtype=torch.bool, device=self._device)

    # ------------------------------------------------------------------------------------------
    # composition (flow_class.py:1648-1810)
    # ------------------------------------------------------------------------------------------
    def combine_with(self, flow: FlowAlias, mode: int, thresholded: bool = None) -> FlowAlias:
        """flow_1 (+) flow_2 = flow_3 for two flows of equal shape and reference; `mode` names the unknown.
        Mode 3 is ONE fused launch (gather of (u, v, mask) + vector add + mask AND)."""
        if not isinstance(flow, Flow):
            raise TypeError("Error combining flows: Flow need to be of type 'Flow'")
        if self.shape != flow.shape:
            raise ValueError("Error combining flows: Flow fields need to have the same shape, including batch size")
        if self.ref != flow.ref:
            raise ValueError("Error combining flows: Flow fields need to have the same reference")
        if self._device != flow._device:
            flow = flow.to_device(self._device)
        if mode not in [1, 2, 3]:
            raise ValueError("Error combining flows: Mode needs to be 1, 2 or 3")
        thresholded = False if thresholded is None else thresholded
        if not isinstance(thresholded, bool):
            raise TypeError("Error combining flows: Thresholded needs to be a boolean")
        speculative = None
        if mode == 3 and not (self._flags_known() and flow._flags_known()):
            # flags not known yet: launch the fused composition right away; it returns both operands' flag words as a
            # by-product, and the reference's validation / early-exit decisions are taken afterwards (result discarded
            # if one of them fires).  One pass over the data instead of validation passes + the composition.
            speculative = self._combine3(flow, speculative=True)
        self._require_finite("Error setting flow vectors: " if speculative is not None else "Error combining flows: ")
        flow._require_finite("Error setting flow vectors: " if speculative is not None else "Error combining flows: ")

        bit = _native.FLAG_NZ_THR_MASKED if thresholded else _native.FLAG_NZ_MASKED
        if self._all_zero(bit):                                                      # flow_class.py:1729-1737
            return flow
        if flow._all_zero(bit):                                                      # :1738-1744
            return self if mode == 3 else self.invert()

        ref = self._ref
        if mode == 3:
            if speculative is not None and not (flow if ref == 't' else self)._all_zero(_native.FLAG_NZ_THR):
                return speculative
            return self._combine3(flow)
        if mode == 1:
            if ref == 's':                                                           # :1759-1760
                # flow - (flow_inv_t + flow_inv_t.apply(self.switch_ref())).apply(self) with flow_inv_t = flow.invert('t') =
                # Flow(-flow.vecs, 't'): the inner sum is the fused mode-3 launch (the negation folded into its signs), the
                # outer difference the epilogue of the second gather
                return flow._minus_applied(self.switch_ref()._combine3(flow, result_is_warper=True, negated_warper=True), self)
            # self.invert().apply(flow - self)   (:1763): the difference is formed while the source box is staged
            inv = self.invert()
            if inv._all_zero(_native.FLAG_NZ_THR):
                return inv.apply(flow - self)
            warped, valid, _ = inv._warp(flow._fv, flow._and_masks(self._mask), True, True, t_minus=self._vecs)
            return Flow._wrap(warped, 't', valid, self._device, made_from=(flow._fv, flow._mask, self._mask, inv._mask))
        if ref == 's':                                                               # mode 2, :1768
            # self.apply(flow - self): the difference (and the AND of the two masks) is formed inside the splat
            if self._all_zero(_native.FLAG_NZ_THR) or not get_pure_pytorch():        # (apply's early exit / griddata gate)
                return self.apply(flow - self)
            warped, valid, dflags = self._warp(flow._vecs, flow._mask, True, True, t_minus=self._vecs)
            return Flow._wrap(warped, 's', valid, self._device, flags=dflags, made_from=(flow._fv, flow._mask, self._mask))
        if not get_pure_pytorch():
            _griddata_unavailable("combine_with(mode=2, ref='t')")
        return flow._minus_applied(flow, self.invert().apply(self))                  # :1773  flow - flow.apply(...)

    def _minus_applied(self, warper: FlowAlias, target: FlowAlias) -> FlowAlias:
        """self - warper.apply(target) for a 't'-referenced `warper` and a flow `target`: one launch, the subtraction is
        the gather's epilogue (1 * self + (-1) * G: the same fp32 operation as the reference's `-`); masks AND as in
        flow_class.py:490-531 / 921-934.  The thresholded early exit of `apply` takes the plain path."""
        if warper._ref != 't' or warper._all_zero(_native.FLAG_NZ_THR) or self.shape[0] != warper.shape[0] \
                or target.shape[0] != warper.shape[0]:
            return self - warper.apply(target)
        warper._require_finite("Error applying flow to a target: ")
        vecs, valid, _, _ = _native.warp_bwd(warper._vecs, target._vecs, src_mask=target._mask, flow_mask=warper._mask,
                                             want_valid=True, addend=self._vecs, a_sign=1.0, g_sign=-1.0)
        if self._mask is not None and self._mask is not warper._mask:
            valid = valid & self._mask
        return Flow._wrap(vecs, self._ref, valid, self._device, fresh=True)

    def _combine3(self, flow: FlowAlias, speculative: bool = False, result_is_warper: bool = False,
                  negated_warper: bool = False) -> FlowAlias:
        """mode 3: 't'  f3 = f2 + G(f2, f1),  m3 = m2 & theta(G(f2, [m1]))           (flow_class.py:1808)
                   's'  f3 = f1 + G(-f1, f2), m3 = m1 & theta(G(-f1, [m2]))          (flow_class.py:1804)
        `negated_warper` ('t' only): `flow` stands for Flow(-flow.vecs, 't', flow.mask) -- `flow.invert('t')` of a
        forward flow -- without the negated copy: positions and addend change sign in the kernel, flags do not."""
        a_sign = 1.0
        if self._ref == 't':
            warper, sign, src = flow, 1.0, self
            if negated_warper:
                sign, a_sign = -1.0, -1.0
        else:
            warper, sign, src = self, -1.0, flow
        if not speculative and warper._all_zero(_native.FLAG_NZ_THR):
            # apply_flow's thresholded early exit inside .apply (utils.py:497): the gather is the identity
            w_ = warper._negated('t') if a_sign < 0 else warper
            return w_ + Flow._wrap(src._vecs, src._ref, src._and_masks(warper._mask), self._device)
        # (`result_is_warper`: the result will warp something next, so its own flag word is worth the ~5 % it costs here)
        res = _native.warp_bwd(warper._vecs, src._vecs, flow_sign=sign, src_mask=src._mask,
                               flow_mask=warper._mask, want_valid=True, addend=warper._vecs, a_sign=a_sign,
                               want_flags=speculative, want_src_flags=speculative, want_dst_flags=result_is_warper)
        vecs, valid, wf, sf = res[:4]
        if speculative:
            if not warper._flags_known():
                warper._set_pending_flags(wf)
            if not src._flags_known():
                src._set_pending_flags(sf)
        return Flow._wrap(vecs, self._ref, valid, self._device, flags=res[4] if result_is_warper else None, fresh=True)

    # ------------------------------------------------------------------------------------------
    # general composition (flow_class.py:1812-1939)
    # ------------------------------------------------------------------------------------------
    def combine(self, other: FlowAlias, mode: int, ref: str = None) -> FlowAlias:
        """flow_1 (+) flow_2 = flow_3 for two flows of equal shape and ANY references, result in reference `ref`
        (default: that of self); `mode` names the unknown.  Same results as the reference (flow_class.py:1812-1939, every
        (mode, self.ref, other.ref, ref) cell pinned by fixtures), computed from a per-cell PLAN (`_combine_plan`): at most
        one `switch_ref`, then ONE fused launch -- the linear combination of the two fields rides the epilogue of the
        backward gather (`a_sign * close + g_sign * G(+-close, far)`) or is formed inside the forward splat (`data - data_b`),
        and the `invert` of the carrying flow is a sign of the kernel's sample positions / end points."""
        if not isinstance(other, Flow):
            raise TypeError("Error combining flows: Flow need to be of type 'Flow'")
        if self.shape != other.shape:
            raise ValueError("Error combining flows: Flow fields need to have the same shape, including batch size")
        if self._device != other._device:
            other = other.to_device(self._device)
        if mode not in [1, 2, 3]:
            raise ValueError("Error combining flows: Mode needs to be 1, 2 or 3")
        ref = self._ref if ref is None else get_valid_ref(ref)
        plan = _combine_plan(mode, self._ref, other._ref, ref)
        close, far = (self, other) if plan.close_is_self else (other, self)
        if plan.switch_far:                                  # the far field onto the pivot grid (a splat, or a relabel)
            far = far.switch_ref()
        if plan.close_on_goal:
            result = close._carry_back_and_add(far, plan.carry_sign, plan.sign_close, plan.sign_far)
        else:
            result = close._add_and_carry_forward(far, plan.carry_sign, plan.sign_close, plan.sign_far)
        result._ref = ref
        return result

    def _carry_back_and_add(self, field: FlowAlias, carry_sign: float, sign_self: float, sign_field: float) -> FlowAlias:
        """sign_self * self + sign_field * W(field), with W the backward warp of `field` (a flow on another grid) onto this
        flow's grid along `self` (carry_sign +1: self is 't'-referenced) or along Flow(-self.vecs, 't') (carry_sign -1: self
        is 's'-referenced and `invert('t')`-ed, flow_class.py:1081).  Vectors `s_self * self + s_field * G`, mask
        `theta(G(mask channel)) & self.mask` -- the reference's apply + two scalings + add (flow_class.py:921-934, 450-488,
        533-549), as the addend epilogue of one gather."""
        warper_zero = self._all_zero(_native.FLAG_NZ_THR)      # apply_flow's early exit (utils.py:497-498): W is the identity
        if warper_zero or self.shape[0] != field.shape[0] or not _COMBINE_FUSED:
            carrier = self if carry_sign > 0 else self._negated('t')
            return carrier.apply(field) * sign_field + self * sign_self
        self._require_finite("Error applying flow to a target: ")
        res = _native.warp_bwd(self._vecs, field._vecs, flow_sign=carry_sign, src_mask=field._mask, flow_mask=self._mask,
                               want_valid=True, addend=self._vecs, a_sign=sign_self, g_sign=sign_field)
        return Flow._wrap(res[0], self._ref, res[1], self._device, fresh=True)

    def _add_and_carry_forward(self, field: FlowAlias, carry_sign: float, sign_self: float, sign_field: float) -> FlowAlias:
        """P(sign_self * self + sign_field * field): the combination of two fields on one grid, carried to the other end of
        `self` by a forward splat along `self` (carry_sign +1: self is 's'-referenced) or along Flow(-self.vecs, 's')
        (carry_sign -1: self is 't'-referenced and `invert('s')`-ed, flow_class.py:1084).  A difference is formed inside the
        splat (`data - data_b`, the same fp32 subtraction: a + (-b) == a - b); a sum takes one elementwise pass first."""
        if sign_self > 0 and sign_field > 0:
            total, minus = self + field, None
        elif sign_self > 0:
            total, minus = self, field                       # self - field
        else:
            total, minus = field, self                       # field - self
        warper = self if carry_sign > 0 else Flow._wrap(self._fv, 's', self._mask, self._device, like=self, borrowed=True)
        if warper._all_zero(_native.FLAG_NZ_THR) or not get_pure_pytorch() or self.shape[0] != field.shape[0] or not _COMBINE_FUSED:
            carrier = self if carry_sign > 0 else self._negated('s')
            return carrier.apply(total if minus is None else total - minus)
        tmask = total._mask if minus is None else total._and_masks(minus._mask)
        warped, valid, dflags = warper._warp(total._fv if minus is None else total._vecs, tmask, True, True, flow_sign=carry_sign,
                                             t_minus=None if minus is None else minus._vecs)
        return Flow._wrap(warped, self._ref, valid, self._device, flags=dflags, made_from=(total._fv, tmask, self._mask))


class _CombinePlan(object):
    """How `Flow.combine` computes one (mode, self.ref, other.ref, ref) cell."""
    __slots__ = ("close_is_self", "switch_far", "close_on_goal", "carry_sign", "sign_close", "sign_far")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


_PLANS = {}


def _combine_plan(mode: int, self_ref: str, other_ref: str, out_ref: str) -> _CombinePlan:
    """The 24 plans of `Flow.combine`, derived from the geometry rather than tabulated by hand.

    Three instants 0, 1, 2; flow_1 spans (0, 1), flow_2 (1, 2), flow_3 (0, 2).  A flow lives on the pixel grid of its
    earlier instant when referenced 's', of its later one when 't'.  The unknown flow spans (a, b); the third instant is the
    PIVOT: both known flows touch it, and the unknown is the vector sum of the hop a -> pivot and the hop pivot -> b once
    both hops sit on one grid.  The result is wanted on the GOAL grid (a for 's', b for 't').  The known flow that touches
    the goal is CLOSE, the other FAR.  A known flow counts +1 for a hop that runs with it (earlier -> later), -1 against it.
      1. FAR must sit on the pivot grid: `switch_ref` if it sits on the unknown's far end instead.
      2. CLOSE on the goal grid: FAR is carried pivot -> goal by a backward warp along CLOSE (negated when CLOSE runs
         goal -> pivot), then the two hops are added there.   CLOSE on the pivot grid: the hops are added there, and the
         sum is carried pivot -> goal by a forward warp along CLOSE (negated when CLOSE runs goal -> pivot)."""
    key = (mode, self_ref, other_ref, out_ref)
    plan = _PLANS.get(key)
    if plan is not None:
        return plan
    spans = {1: (0, 1), 2: (1, 2), 3: (0, 2)}
    a, b = spans[mode]
    pivot = 3 - a - b
    goal = a if out_ref == 's' else b
    known = [k for k in (1, 2, 3) if k != mode]            # self is the lower-numbered known flow, other the higher
    grid = lambda span, r: span[0] if r == 's' else span[1]
    self_span, other_span = spans[known[0]], spans[known[1]]
    close_is_self = goal in self_span
    close_span, close_ref = (self_span, self_ref) if close_is_self else (other_span, other_ref)
    far_span, far_ref = (other_span, other_ref) if close_is_self else (self_span, self_ref)
    along = lambda u, v: 1.0 if u < v else -1.0           # a hop u -> v measured against a flow that runs earlier -> later
    hop_in, hop_out = along(a, pivot), along(pivot, b)     # a -> pivot, pivot -> b
    plan = _CombinePlan(
        close_is_self=close_is_self,
        switch_far=grid(far_span, far_ref) != pivot,
        close_on_goal=grid(close_span, close_ref) == goal,
        carry_sign=along(pivot, goal),
        sign_close=hop_out if goal == b else hop_in,       # CLOSE spans pivot-goal: the hop that ends (starts) at the goal
        sign_far=hop_in if goal == b else hop_out)
    _PLANS[key] = plan
    return plan


# the public methods that read a flow's flag word: each starts a new validation epoch under set_revalidate_every_call(True)
for _name in ('apply', 'track', 'switch_ref', 'invert', 'valid_target', 'valid_source', 'get_padding', 'is_zero', 'combine_with',
              'combine'):
    setattr(Flow, _name, _public(getattr(Flow, _name)))
del _name
