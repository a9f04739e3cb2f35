"""ctypes binding of libofl_hip.so (C ABI: include/oflib_hip.h) and the tensor-level primitives the
host-side mirror of the reference API is written against.

This is the ONLY compute backend of the package.  There is no CPU fallback: if the HIP library
cannot be loaded, or no HIP device is visible, every primitive raises ``NativeUnavailable``.
Tensors that live on the CPU are staged to the current HIP device for the launch (the reference
accepts CPU tensors; the arithmetic still runs in the HIP kernels) and results are returned on the
HIP device -- callers move them where the reference would have put them.
"""
import ctypes
import os
import threading
import time

import numpy as np
import torch

from . import _build

FLAG_NONFINITE, FLAG_NZ, FLAG_NZ_THR, FLAG_NZ_MASKED, FLAG_NZ_THR_MASKED = 1, 2, 4, 8, 16
ABI_VERSION = 33              # ofl_version() of the library this file's argtypes describe
ROUND_NONE, ROUND_RINT, ROUND_U8 = 0, 1, 2
THRESHOLD = 1e-3

_SYMBOLS = ("ofl_version", "ofl_set_option", "ofl_warp_bwd_f32", "ofl_splat_fwd_f32", "ofl_splat_finalize_f32", "ofl_splat_tiled_workspace_ints", "ofl_splat_tiled_pass_images",
            "ofl_splat_tiled_f32", "ofl_flow_flags_f32", "ofl_warp_bwd_u8", "ofl_flow_from_f16",
            "ofl_warp_bwd_grad_f32", "ofl_splat_grad_f32", "ofl_sample_pts_f32", "ofl_sample_pts_grad_f32",
            "ofl_flow_extents_f32", "ofl_flag_words_or_i32", "ofl_splat_sum_f32", "ofl_warp_bwd_win_f32", "ofl_splat_tiled_win_f32", "ofl_splat_tiled_f16",
            "ofl_warp_bwd_h_f32", "ofl_flow_flags_host", "ofl_host_words_alloc", "ofl_host_words_free", "ofl_flow_from_matrix_f32", "ofl_splat_tiled_fallback_images", "ofl_warp_valid_f32", "ofl_resize_bilinear_f32", "ofl_splat_tile_geometry", "ofl_splat_gather_info", "ofl_last_kernel_name")
_lib = None


class NativeUnavailable(RuntimeError):
    """The HIP backend (libofl_hip.so + a visible HIP device) is required and missing."""


def load_library(path: str = None):
    """dlopen libofl_hip.so and declare the C ABI.  Never touches the GPU (usable in CPU-only checks)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("OFL_HIP_LIB") or _build.LIB_PATH   # OFL_HIP_LIB: A/B of two builds (tools/ab_warp.py)
    stale = False
    if os.path.abspath(path) == os.path.abspath(_build.LIB_PATH):
        try:                                    # the in-tree library is rebuilt when a source or the header is newer
            stale = _build.needs_build()
        except OSError:
            stale = False
    if not os.path.exists(path) or stale:
        try:
            _build.build()
        except Exception as exc:  # noqa: BLE001
            if not os.path.exists(path):
                raise NativeUnavailable("oflibpytorch_amd: %s is missing and could not be built: %s" % (path, exc))
            import warnings                      # (no compiler here: run what is there, the ABI check below still applies)
            warnings.warn("oflibpytorch_amd: %s is older than its sources and could not be rebuilt: %s" % (path, exc))
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:
        raise NativeUnavailable("oflibpytorch_amd: cannot load %s: %s" % (path, exc))
    p, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    lib.ofl_version.argtypes = []
    lib.ofl_version.restype = ctypes.c_int
    if lib.ofl_version() != ABI_VERSION:     # a stale build would take the arguments below in the wrong places
        raise NativeUnavailable("oflibpytorch_amd: %s has ABI version %d, this package needs %d -- rebuild it "
                                "(python -m oflibpytorch_amd._build)" % (path, lib.ofl_version(), ABI_VERSION))
    lib.ofl_set_option.argtypes = [i32, i32]
    lib.ofl_warp_bwd_f32.argtypes = [p, i64, f32, p, i64, p, i64, p, i64, p, i64, p, i64, f32, f32, p, p, p, p, p,
                                     i32, i32, i32, i32, i32, p]
    lib.ofl_flow_from_f16.argtypes = [p, i64, p, i64, p, p, i32, i32, i32, p]
    lib.ofl_warp_bwd_u8.argtypes = [p, i64, f32, p, i64, p, i64, p, i64, p, i32, p, i32, i32, i32, i32, i32, p]
    lib.ofl_splat_fwd_f32.argtypes = [p, i64, f32, p, p, i64, p, i64, f32, p, i64, p, i64, p, i64, i32, i32, p,
                                      i32, i32, i32, i32, p]
    lib.ofl_splat_finalize_f32.argtypes = [p, p, i64, p, i64, f32, p, i64, p, i64, p, i64, i32, i32, p, p, p, p, p,
                                           i32, i32, i32, i32, i32, p]
    lib.ofl_flow_flags_f32.argtypes = [p, i64, p, i64, f32, p, i32, i32, i32, p]
    lib.ofl_splat_tiled_workspace_ints.argtypes = [i32, i32, i32]
    lib.ofl_splat_tiled_pass_images.argtypes = [i32, i32, i32]
    lib.ofl_splat_tiled_fallback_images.argtypes = [i32, i32, i32, i32]
    lib.ofl_splat_tile_geometry.argtypes = [p, p, p]
    lib.ofl_splat_gather_info.argtypes = [i32, i32, i32, i32, i32, p]
    lib.ofl_splat_tiled_f32.argtypes = [p, i64, f32, p, p, i64, p, i64, f32, p, i64, p, i64, p, i64, p, i64, i32, i32, p, p, p, p, p,
                                        p, p, i64, p, i32, i32, i32, i32, i32, p]
    lib.ofl_warp_bwd_grad_f32.argtypes = [p, i64, f32, p, i64, p, f32, p, i64, p, i32, i32, i32, i32, p]
    lib.ofl_splat_grad_f32.argtypes = [p, i64, f32, p, p, i64, p, i64, p, i64, i32, p, p, p, p, p, p, p, i32, i32, i32, i32, p]
    lib.ofl_sample_pts_f32.argtypes = [p, i64, p, i64, p, i32, i32, i32, i32, p]
    lib.ofl_sample_pts_grad_f32.argtypes = [p, i64, p, i64, p, p, p, i32, i32, i32, i32, p]
    lib.ofl_flow_extents_f32.argtypes = [p, i64, p, i64, f32, p, p, i32, i32, i32, p]
    lib.ofl_warp_valid_f32.argtypes = [p, i64, f32, p, i64, f32, p, i32, i32, i32, p]
    lib.ofl_resize_bilinear_f32.argtypes = [p, p, i32, i32, i32, i32, i32, f32, f32, p]
    lib.ofl_flag_words_or_i32.argtypes = [p, i32, p, p]
    lib.ofl_splat_sum_f32.argtypes = [p, i64, f32, p, i64, f32, p, p, i64, p, i32, i32, i32, i32, p]
    lib.ofl_splat_tiled_f16.argtypes = [p, i64, f32, p, i64, f32, p, i64, p, i64, p, i64, i32, i32, p, i32, p, p, p, i64, p, i32, i32, i32, p]
    lib.ofl_warp_bwd_h_f32.argtypes = [p, i64, f32, p, i64, p, i64, p, i64, p, i64, p, p, i32, i32, i32, p]
    lib.ofl_warp_bwd_win_f32.argtypes = [p, i64, f32, i32, i32, i32, i32, p, i64, p, i64, p, i64, p, p, i32, i32, i32, i32, i32, p]
    lib.ofl_splat_tiled_win_f32.argtypes = [p, i64, f32, i32, i32, i32, i32, p, i64, f32, p, i64, p, i64, p, i64, i32, i32, p, p, p, p, p,
                                            p, i64, p, i32, i32, i32, i32, i32, p]
    lib.ofl_flow_flags_host.argtypes = [p, i32, i64, p, i64, f32, p, p, i32, i32, i32, i32, p]
    lib.ofl_host_words_alloc.argtypes = [i64, ctypes.POINTER(ctypes.c_void_p)]
    lib.ofl_host_words_free.argtypes = [p]
    lib.ofl_flow_from_matrix_f32.argtypes = [p, i64, f32, p, i32, i32, i32, p]
    for name in _SYMBOLS:
        getattr(lib, name).restype = ctypes.c_int
    lib.ofl_splat_tiled_workspace_ints.restype = ctypes.c_int64
    lib.ofl_last_kernel_name.restype = ctypes.c_char_p
    lib.ofl_last_kernel_name.argtypes = []
    lib.ofl_splat_tiled_pass_images.restype = ctypes.c_int64
    lib.ofl_splat_tiled_fallback_images.restype = ctypes.c_int64
    _lib = lib
    return lib


_splat_path = 0


def set_splat_path(mode: int):
    """0 = auto (fused tiled kernel when eligible), 1 = general two-pass atomics path only (tests compare the two)."""
    global _splat_path
    _splat_path = int(mode)


collect_splat_stats = False   # tests / tools set this to read _last_splat_stats after a gather (in-order) splat
_last_splat_stats = None      # device int32[4] of the most recent gather splat: how exact was it
_last_splat_ws = None


def set_warp_shear(on: bool):
    """LDS-staged warp kernel: True = y-sheared staging box (default), False = plain bounding box (speed only)."""
    _check(load_library().ofl_set_option(3, 1 if on else 0), "ofl_set_option")


def set_splat_pass_images(k: int):
    """Gather splat: at most k images per pass (0 = automatic); tests use it to force several passes."""
    _check(load_library().ofl_set_option(4, int(k)), "ofl_set_option")


def set_warp_path(mode: int):
    """0 = auto (LDS-staged kernel when eligible), 1 = generic direct-gather kernel only, 3 / 4 = staged with two tiles / one tile
    per block whatever the launch size, 5 = auto but more than 3 channels as separate launches of 3 instead of the channel-loop
    kernel, 6 = auto but the sheared rectangle instead of per-row extents, 7 = auto but four-tile row-table columns whatever the
    launch size (tests compare them all)."""
    _check(load_library().ofl_set_option(1, int(mode)), "ofl_set_option")


def exported_symbols():
    return _SYMBOLS


_hip_seen = False


def device(*operands) -> torch.device:
    """The HIP device the kernels run on: the device of the first operand that already lives on one (the flow comes
    first in every primitive), else torch's current device."""
    global _hip_seen
    if not _hip_seen:
        if not torch.cuda.is_available():
            raise NativeUnavailable("oflibpytorch_amd: no HIP device visible -- this package has no CPU fallback "
                                    "(its compute path is libofl_hip.so on MI355X)")
        _hip_seen = True
    for t in operands:
        if isinstance(t, torch.Tensor) and t.device.type == 'cuda':
            return t.device
    return torch.device('cuda', torch.cuda.current_device())


def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


class _Here(object):
    """No-op launch context: the operands already live on torch's current device (the common case -- entering
    `torch.cuda.device` costs several microseconds per call, which a small batch pays with the GPU idle)."""
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_HERE = _Here()


def _on(dev):
    """Launch context: kernels go to the current stream OF THE OPERANDS' DEVICE, whatever torch's current device is."""
    return _HERE if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError("oflibpytorch_amd: %s failed with status %d" % (what, rc))


def _stream(dev):
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index))


def _planes(t: torch.Tensor, dev, dtype, n: int, what: str):
    """Stage tensor [Nb, (C,) H, W] for a kernel: on `dev`, `dtype`, planes contiguous; returns
    (tensor to keep alive, batch stride in elements -- 0 broadcasts one batch element)."""
    if t.device == dev and t.dtype == dtype and t.shape[0] == n and t.is_contiguous():
        return t, (0 if n == 1 else t.stride(0))           # (the common case, kept short: host time is exposed at small batch)
    if t.device != dev:
        t = t.to(dev)
    if t.dtype != dtype:
        t = t.to(dtype)
    nb = t.shape[0]
    if nb != n and nb != 1:
        raise ValueError("oflibpytorch_amd: %s batch size %d cannot broadcast to %d" % (what, nb, n))
    if nb > 1 and t.stride(0) == 0:
        t, nb = t[:1], 1          # an `expand`ed batch (utils.py:533-537) is a broadcast, not N copies
    if not t[0].is_contiguous():
        t = t.contiguous()
    return t, (0 if nb == 1 else t.stride(0))


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def flow_flags(vecs: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
    """Flag word per batch element (bits: FLAG_*), int32 tensor [N] on the HIP device (no host sync)."""
    lib, dev = load_library(), device(vecs, mask)
    n, _, h, w = vecs.shape
    if vecs.dtype == torch.float16 and vecs.device.type == 'cuda':       # fp16 storage: flags without an fp32 copy (5 B/px)
        with _on(dev):
            v, vbs = _planes(vecs.detach(), dev, torch.float16, n, "flow")
            if vbs != 0 or n == 1:
                m, mbs = (None, 0) if mask is None else _planes(mask, dev, torch.bool, n, "mask")
                flags = torch.zeros(n, dtype=torch.int32, device=dev)
                rc = lib.ofl_flow_from_f16(_ptr(v), vbs, _ptr(m), mbs, None, _ptr(flags), n, h, w, _stream(dev))
                if rc != -4:
                    _check(rc, "ofl_flow_from_f16")
                    return flags
        vecs = vecs.float()
    with _on(dev):
        v, vbs = _planes(vecs.detach(), dev, torch.float32, n, "flow")
        m, mbs = (None, 0) if mask is None else _planes(mask, dev, torch.bool, n, "mask")
        flags = torch.zeros(n, dtype=torch.int32, device=dev)
        _check(lib.ofl_flow_flags_f32(_ptr(v), vbs, _ptr(m), mbs, THRESHOLD, _ptr(flags), n, h, w, _stream(dev)),
               "ofl_flow_flags_f32")
    return flags


# -- validation read-back: the reduction's last block writes the words to host-visible memory, the host polls them ---------------
# A SLOT = the device work words (arrival counters + flag words) and the host-visible {serial, word} pairs of ONE call in flight.
# Slots are pooled per device: a call takes a free one (or makes a new one, up to _MAX_SLOTS) and holds it until its words have
# arrived, so two threads validating on one device run side by side instead of queueing behind one lock.  The
# wait is a host poll: it cannot be recorded into a stream capture / hipGraph (documented in include/oflib_hip.h).
_HOST_WORDS = 1 << 12        # flag words one call can hand over (larger batches take the copy + event route)
_WORK_EXTRA = 33             # OFL_FLAGS_HOST_WORK_EXTRA
_MAX_SLOTS = 8               # per device; a ninth concurrent caller waits for a slot
_host_slots = {}             # device index -> list of slots [lock, device work words, host address, int32 view of the pairs, last serial, retired]
_host_slots_lock = threading.Lock()
_SPIN_SLACK_SECONDS = 150e-6  # tight polling lasts ~2x the time the reduction's bytes take plus this; after that the wait yields the GIL between looks
_YIELD_SECONDS = 20e-3        # ... and after this long it sleeps 50 us between looks
HOST_POLL_SECONDS = 20.0     # a reduction that has not reported after this long is a failed launch, not a slow one


def _new_host_slot(lib, dev):
    addr = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _check(lib.ofl_host_words_alloc(2 * _HOST_WORDS, ctypes.byref(addr)), "ofl_host_words_alloc")
        work = torch.zeros(_HOST_WORDS + _WORK_EXTRA, dtype=torch.int32, device=dev)
    view = np.ctypeslib.as_array((ctypes.c_int32 * (2 * _HOST_WORDS)).from_address(addr.value))
    return [threading.Lock(), work, addr, view, 0, False]


def _acquire_host_slot(lib, dev):
    """A slot nobody is using, locked.  The first slot of a device is the fast path (one non-blocking acquire)."""
    while True:
        slots = _host_slots.get(dev.index)
        if slots is not None:
            for slot in tuple(slots):
                if slot[0].acquire(False):
                    if slot[5]:                      # retired after a failed wait while we were looking
                        slot[0].release()
                        continue
                    return slot
        with _host_slots_lock:
            slots = _host_slots.setdefault(dev.index, [])
            if len(slots) < _MAX_SLOTS:
                slot = _new_host_slot(lib, dev)
                slot[0].acquire()
                slots.append(slot)
                return slot
            busy = slots[0]
        busy[0].acquire()                            # the pool is full: queue behind its first slot
        if not busy[5]:
            return busy
        busy[0].release()                            # (it was retired while we waited: look again)


def _drop_host_slot(dev, slot):
    """After a failed wait the slot's work words may be dirty: forget it (its successor starts from zeroed words)."""
    with _host_slots_lock:
        slot[5] = True
        slots = _host_slots.get(dev.index)
        if slots is not None:
            _host_slots[dev.index] = [s for s in slots if s is not slot]


def flow_flags_host(vecs: torch.Tensor, mask: torch.Tensor = None):
    """Flag word per batch element as a list of host ints: `flow_flags` and its read-back in ONE launch
    (ofl_flow_flags_host: no memset, no copy, no event; the host polls the {serial, word} pairs the kernel's last block
    writes).  None when this call cannot take that route (CPU-resident or fp64 vectors, more than 4096 batch elements, an
    fp16 layout the vector kernel does not take): the caller then uses `flow_flags` + a copy.
    This is the wait every `Flow(...)` ends in, so the common case (contiguous operands already on the device) skips the
    general staging code."""
    if vecs.device.type != 'cuda' or vecs.dtype not in (torch.float32, torch.float16):
        return None
    n, _, h, w = vecs.shape
    if n > _HOST_WORDS:
        return None
    lib, dev = load_library(), vecs.device
    half = vecs.dtype == torch.float16
    switch = torch.cuda.current_device() != dev.index
    ctx = torch.cuda.device(dev) if switch else None
    if ctx is not None:
        ctx.__enter__()
    try:
        if vecs.is_contiguous() and (mask is None or (mask.dtype == torch.bool and mask.device == dev and mask.shape[0] == n
                                                      and mask.is_contiguous())):
            v, vbs, m, mbs = vecs, 2 * h * w, mask, h * w
        else:
            v, vbs = _planes(vecs.detach(), dev, vecs.dtype, n, "flow")
            if half and vbs == 0 and n != 1:
                return None
            m, mbs = (None, 0) if mask is None else _planes(mask, dev, torch.bool, n, "mask")
        slot = _acquire_host_slot(lib, dev)
        stream = torch._C._cuda_getCurrentRawStream(dev.index)
        try:
            serial = slot[4] = (slot[4] % 0x7ffffff0) + 1
            view = slot[3]
            rc = lib.ofl_flow_flags_host(v.data_ptr(), 1 if half else 0, vbs, 0 if m is None else m.data_ptr(), mbs, THRESHOLD,
                                         slot[1].data_ptr(), slot[2], serial, n, h, w, stream)
            if rc == -4:
                return None
            _check(rc, "ofl_flow_flags_host")
            try:
                tags = view[0:2 * n:2]
                # tight polling for about as long as the reduction itself can take (its bytes at ~4 TB/s, twice over, plus a
                # slack), then sleeps between looks: a stream with milliseconds of work queued in front must not keep the GIL
                # and a core busy, and other Python threads (data loaders) get to run
                t0 = time.perf_counter()
                spin_until = t0 + 2.0 * (n * h * w * (5 if half else 9)) / 4e12 + _SPIN_SLACK_SECONDS
                looks = 0
                while not (view[2 * n - 2] == serial and (n == 1 or bool((tags == serial).all()))):
                    looks += 1
                    if looks & 0x3f == 0:
                        now = time.perf_counter()
                        if now > spin_until:
                            # the stream has work queued in front of the reduction (the previous step's kernels): keep looking,
                            # but hand the GIL to any other Python thread between looks (sleep(0) returns at once when nobody
                            # wants it -- a timed sleep here cost 30-60 us of wake-up latency per validation, 14 % of a B = 8
                            # step); only a wait of many milliseconds backs off to timed sleeps
                            time.sleep(0 if now - t0 < _YIELD_SECONDS else 5e-5)
                            if now - t0 > HOST_POLL_SECONDS:
                                torch.cuda.synchronize(dev)          # surfaces a launch failure as the runtime's own error
                                if bool((tags == serial).all()):
                                    break
                                raise RuntimeError("oflibpytorch_amd: the flag reduction did not report back")
                return view[1:2 * n:2].tolist()
            except BaseException:
                # (KeyboardInterrupt included) the kernel may still be in flight on this slot's words: wait for the device, and
                # retire the slot rather than hand possibly dirty work words to the next call
                try:
                    torch.cuda.synchronize(dev)
                finally:
                    _drop_host_slot(dev, slot)
                raise
        finally:
            slot[0].release()
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)


def flow_from_half(vecs16: torch.Tensor, mask: torch.Tensor = None):
    """fp16-stored flow [N,2,H,W] -> (fp32 copy, flag words int32[N]) in one pass (ofl_flow_from_f16); shapes / alignments
    the kernel does not take are converted with torch and flagged by ofl_flow_flags_f32."""
    lib, dev = load_library(), device(vecs16, mask)
    n, _, h, w = vecs16.shape
    if _wants_grad(vecs16):                  # the reference's `.float()` is differentiable (utils.py:118): keep the graph
        dst = vecs16.to(dev).float()
        return dst, flow_flags(dst, mask)
    with _on(dev):
        v, vbs = _planes(vecs16, dev, torch.float16, n, "flow")
        if vbs != 0 or n == 1:
            m, mbs = (None, 0) if mask is None else _planes(mask, dev, torch.bool, n, "mask")
            dst = torch.empty((n, 2, h, w), dtype=torch.float32, device=dev)
            flags = torch.zeros(n, dtype=torch.int32, device=dev)
            rc = lib.ofl_flow_from_f16(_ptr(v), vbs, _ptr(m), mbs, _ptr(dst), _ptr(flags), n, h, w, _stream(dev))
            if rc != -4:
                _check(rc, "ofl_flow_from_f16")
                return dst, flags
    dst = vecs16.to(dev).float()
    return dst, flow_flags(dst, mask)


def _warp_bwd_lean(flow, src, flow_sign=1.0, src_mask=None, flow_mask=None, want_valid=False, addend=None, a_sign=1.0,
                   g_sign=1.0, round_mode=ROUND_NONE, want_flags=False, want_src_flags=False, want_dst_flags=False,
                   src_b=None, out_uint8=False):
    """The plain backward warp of `Flow.apply` 't' and `Flow.combine_with` (fp32 operands already on the current HIP device,
    contiguous, no src_b / flag by-products / rounding, nothing that wants a gradient) without the general staging code: a
    B = 1 call is HOST-bound (the kernel takes 16 us at 1080p, the general path 24 us of Python), and at small batch the
    host time in front of a launch is exposed.  Same C entry point, same arguments as `_warp_bwd_raw`; None when the call
    is not of this kind."""
    if round_mode or want_flags or want_dst_flags or src_b is not None:
        return None
    dev = flow.device
    if (src.dtype is not torch.float32 or src.device != dev or torch.cuda.current_device() != dev.index
            or not flow.is_contiguous() or not src.is_contiguous()):
        return None
    n = max(flow.shape[0], src.shape[0])
    for m in (src_mask, flow_mask):
        if m is not None:
            if m.dtype is not torch.bool or m.device != dev or not m.is_contiguous():
                return None
            n = max(n, m.shape[0])
    c, h, w = src.shape[1:]
    if addend is not None:
        if (addend.dtype is not torch.float32 or addend.device != dev or not addend.is_contiguous()
                or addend.requires_grad and torch.is_grad_enabled()):
            return None
        n = max(n, addend.shape[0])
    for t in (flow, src, src_mask, flow_mask, addend):
        if t is not None and t.shape[0] != n and t.shape[0] != 1:
            return None
    hw = h * w
    dst = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
    valid = torch.empty((n, h, w), dtype=torch.bool, device=dev) if want_valid else None
    one = n == 1
    rc = (_lib or load_library()).ofl_warp_bwd_f32(
        flow.data_ptr(), 0 if (one or flow.shape[0] == 1) else 2 * hw, float(flow_sign),
        src.data_ptr(), 0 if (one or src.shape[0] == 1) else c * hw, 0, 0,
        0 if src_mask is None else src_mask.data_ptr(), 0 if (src_mask is None or one or src_mask.shape[0] == 1) else hw,
        0 if flow_mask is None else flow_mask.data_ptr(), 0 if (flow_mask is None or one or flow_mask.shape[0] == 1) else hw,
        0 if addend is None else addend.data_ptr(), 0 if (addend is None or one or addend.shape[0] == 1) else c * hw,
        float(a_sign), float(g_sign), dst.data_ptr(), 0 if valid is None else valid.data_ptr(), 0, 0, 0, n, c, h, w, 0,
        torch._C._cuda_getCurrentRawStream(dev.index))
    _check(rc, "ofl_warp_bwd_f32")
    return dst, valid, None, None


def warp_bwd(flow, src, **kw):
    """G-family primitive; see `_warp_bwd_raw` for the arguments.  When autograd is recording and the flow, the source,
    `src_b` or the addend requires a gradient, the launch goes through `_autograd.WarpFn` (backward kernels:
    ofl_warp_bwd_grad_f32) -- the reference's outputs are differentiable wrt flow and target (utils.py:555)."""
    if (flow.dtype is torch.float32 and flow.device.type == 'cuda'
            and not (torch.is_grad_enabled() and (flow.requires_grad or src.requires_grad))):
        res = _warp_bwd_lean(flow, src, **kw)
        if res is not None:
            return res
    if flow.dtype == torch.float16:
        flow = flow.float()                       # (the warper itself is read as fp32: exact up-conversion, utils.py:118)
    if src.dtype == torch.float16:
        res = None if _wants_grad(flow, src, kw.get("addend"), kw.get("src_b")) else _warp_bwd_half_src(flow, src, **kw)
        if res is not None:
            return res
        src = src.float()
    for key in ("addend", "src_b"):
        if kw.get(key) is not None and kw[key].dtype == torch.float16:
            kw[key] = kw[key].float()
    if _wants_grad(flow, src, kw.get("addend"), kw.get("src_b")):
        from . import _autograd
        return _autograd.warp(flow, src, **kw)
    with _on(device(flow, src)):
        return _warp_bwd_raw(flow, src, **kw)


def splat_fwd(flow, data, **kw):
    """P-family primitive; see `_splat_fwd_raw`.  Differentiable wrt flow (or xs, ys), data and data_b through
    `_autograd.SplatFn` (ofl_splat_grad_f32) when autograd is recording (utils.py:1079-1080, 1167)."""
    out_half = bool(kw.pop("out_half", False))
    half_f, half_d = flow is not None and flow.dtype == torch.float16, data.dtype == torch.float16
    if half_f or half_d:
        res = None
        if half_f and half_d and not _wants_grad(flow, data, kw.get("data_b")):
            res = _splat_fwd_half(flow, data, out_half=out_half, **kw)
        if res is not None:
            return res
        flow = flow.float() if half_f else flow          # (exact up-conversion, utils.py:118)
        data = data.float() if half_d else data
    if kw.get("data_b") is not None and kw["data_b"].dtype == torch.float16:
        kw["data_b"] = kw["data_b"].float()
    if _wants_grad(flow, data, kw.get("xs"), kw.get("ys"), kw.get("data_b")):
        from . import _autograd
        return _autograd.splat(flow, data, **kw)
    with _on(device(flow, data, kw.get("xs"))):
        return _splat_fwd_raw(flow, data, **kw)


def _warp_bwd_half_src(flow, src, *, flow_sign=1.0, src_mask=None, flow_mask=None, want_valid=False, addend=None, a_sign=1.0,
                       g_sign=1.0, round_mode=ROUND_NONE, want_flags=False, want_src_flags=False, want_dst_flags=False,
                       src_b=None, out_uint8=False):
    """ofl_warp_bwd_h_f32: a flow stored in fp16 gathered straight from its halves (2 channels, valid mask wanted, optional
    fp32 src_b).  None when this launch is not of that kind: the caller up-converts and takes the fp32 kernel."""
    c, h, w = src.shape[1:]
    if not (c == 2 and want_valid and addend is None and round_mode == ROUND_NONE and not want_flags and not want_dst_flags
            and src.device.type == 'cuda'):
        return None
    lib, dev = load_library(), device(flow, src)
    n = max(flow.shape[0], src.shape[0], 1 if src_b is None else src_b.shape[0], 1 if src_mask is None else src_mask.shape[0],
            1 if flow_mask is None else flow_mask.shape[0])
    with _on(dev):
        f, fbs = _planes(flow, dev, torch.float32, n, "flow")
        s16, sbs = _planes(src, dev, torch.float16, n, "source")
        s2, s2bs = (None, 0) if src_b is None else _planes(src_b, dev, torch.float32, n, "source")
        sm, smbs = (None, 0) if src_mask is None else _planes(src_mask, dev, torch.bool, n, "source mask")
        fm, fmbs = (None, 0) if flow_mask is None else _planes(flow_mask, dev, torch.bool, n, "flow mask")
        dst = torch.empty((n, 2, h, w), dtype=torch.float32, device=dev)
        valid = torch.empty((n, h, w), dtype=torch.bool, device=dev)
        rc = lib.ofl_warp_bwd_h_f32(_ptr(f), fbs, float(flow_sign), _ptr(s16), sbs, _ptr(s2), s2bs, _ptr(sm), smbs, _ptr(fm), fmbs,
                                    _ptr(dst), _ptr(valid), n, h, w, _stream(dev))
        if rc == -4:
            return None
        _check(rc, "ofl_warp_bwd_h_f32")
    return dst, valid, None, None


def _splat_fwd_half(flow, data, *, out_half=False, xs=None, ys=None, flow_sign=1.0, data_sign=1.0, weight_mask=None,
                    chan_mask_a=None, chan_mask_b=None, want_valid=False, occlude=True, want_density=False, want_warped=False,
                    round_mode=ROUND_NONE, want_mask_chan=False, want_dst_flags=False, data_b=None):
    """ofl_splat_tiled_f16: warper and data both stored in fp16 (a flow splatted by a flow: switch_ref / invert /
    Flow.apply(Flow) on fp16 storage); fp32 or (out_half) fp16 result.  None when the launch is not of that kind."""
    c, h, w = data.shape[1:]
    if not (c == 2 and xs is None and data_b is None and round_mode == ROUND_NONE and not want_density and not want_warped
            and not want_mask_chan and w >= 4 and _splat_path != 1 and flow.device.type == 'cuda' and data.device.type == 'cuda'):
        return None
    lib, dev = load_library(), device(flow, data)
    n = max(data.shape[0], flow.shape[0], 1 if weight_mask is None else weight_mask.shape[0],
            1 if chan_mask_a is None else chan_mask_a.shape[0], 1 if chan_mask_b is None else chan_mask_b.shape[0])
    with _on(dev):
        f, fbs = _planes(flow, dev, torch.float16, n, "flow")
        d, dbs = _planes(data, dev, torch.float16, n, "data")
        wm, wmbs = (None, 0) if weight_mask is None else _planes(weight_mask, dev, torch.bool, n, "mask")
        ca, cabs = (None, 0) if chan_mask_a is None else _planes(chan_mask_a, dev, torch.bool, n, "mask")
        cb, cbbs = (None, 0) if chan_mask_b is None else _planes(chan_mask_b, dev, torch.bool, n, "mask")
        mch = 1 if want_valid else 0
        dst = torch.empty((n, 2, h, w), dtype=torch.float16 if out_half else torch.float32, device=dev)
        valid = torch.empty((n, h, w), dtype=torch.bool, device=dev) if want_valid else None
        dflags = torch.empty((n,), dtype=torch.int32, device=dev) if want_dst_flags else None
        ws = torch.empty(int(lib.ofl_splat_tiled_workspace_ints(n, h, w)), dtype=torch.int32, device=dev)
        accum = _fallback_accum(lib, n, c, mch, h, w, dev)
        rc = lib.ofl_splat_tiled_f16(_ptr(f), fbs, float(flow_sign), _ptr(d), dbs, float(data_sign), _ptr(wm), wmbs, _ptr(ca), cabs,
                                     _ptr(cb), cbbs, mch, 1 if occlude else 0, _ptr(dst), 1 if out_half else 0, _ptr(valid),
                                     _ptr(dflags), _ptr(ws), ws.numel(), _ptr(accum), n, h, w, _stream(dev))
        if rc == -4:
            return None
        _check(rc, "ofl_splat_tiled_f16")
        if collect_splat_stats:
            global _last_splat_stats
            _last_splat_stats = ws[:8].clone()
    if want_dst_flags:
        return dst, valid, None, None, dflags
    return dst, valid, None, None


def _fallback_accum(lib, n, c, mch, h, w, dev):
    """The two-pass fallback accumulator a gather-splat call must bring (touched only when the bin kernel flags an image): the
    pass, capped at 1 GiB by the library -- flagged images beyond that are served in rounds."""
    planes = 1 + min(c, 3) + mch
    return torch.empty((int(lib.ofl_splat_tiled_fallback_images(n, planes, h, w)), planes, h, w), dtype=torch.float32, device=dev)


def splat_tile_geometry() -> tuple:
    """(tile width, tile height, list capacity in 16 x 2 source subtiles) of the gather splat of the loaded library."""
    tw, th, cap = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _check(load_library().ofl_splat_tile_geometry(ctypes.byref(tw), ctypes.byref(th), ctypes.byref(cap)), "ofl_splat_tile_geometry")
    return tw.value, th.value, cap.value


def set_splat_gather_kernel(which: int):
    """Gather splat: 0 = the round-6 kernel (compact records, three blocks per CU; default), 1 = round 5's kernel.  The same sums
    in the same order -- bit-identical results; tests compare the two, tools time them against each other."""
    _check(load_library().ofl_set_option(6, int(which)), "ofl_set_option")


def last_kernel_name(demangle: bool = True) -> str:
    """Name of the kernel the library launched last (the instantiation its launchers picked), demangled when c++filt is about."""
    name = (load_library().ofl_last_kernel_name() or b"").decode()
    if demangle and name:
        import shutil
        import subprocess
        tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
        try:
            name = subprocess.run([tool, name], capture_output=True, text=True, timeout=10).stdout.strip() or name
        except Exception:  # noqa: BLE001
            pass
        name = name.replace("(anonymous namespace)::", "")
    return name


def set_splat_extra_lds(nbytes: int):
    """Measuring aid: bytes of dynamic LDS added to every launch of the round-6 gather kernel (0 = none): fewer blocks per CU, not
    one instruction changed (28 672: two blocks, 65 536: one)."""
    _check(load_library().ofl_set_option(7, int(nbytes)), "ofl_set_option")


def splat_gather_info(channels: int, with_mask_chan: bool = True, elem: int = 0, lean: bool = True, extra_lds: int = 0) -> dict:
    """Resources of the gather kernel as the HIP runtime reports them: blocks per CU, static LDS bytes, VGPRs, scratch bytes."""
    info = (ctypes.c_int32 * 4)()
    _check(load_library().ofl_splat_gather_info(int(channels), 1 if with_mask_chan else 0, int(elem), 1 if lean else 0, int(extra_lds), info),
           "ofl_splat_gather_info")
    return {"blocks_per_cu": info[0], "waves_per_cu": info[0] * 8, "lds_bytes": info[1], "vgprs": info[2], "scratch_bytes": info[3]}


def set_splat_fallback_slots(k: int):
    """Gather splat: the fallback accumulator holds k images (0 = automatic); tests use 1 to force several rounds."""
    _check(load_library().ofl_set_option(5, int(k)), "ofl_set_option")


def _warp_bwd_raw(flow, src, *, flow_sign=1.0, src_mask=None, flow_mask=None, want_valid=False, addend=None,
                  a_sign=1.0, g_sign=1.0, round_mode=ROUND_NONE, want_flags=False, want_src_flags=False,
                  want_dst_flags=False, src_b=None, out_uint8=False):
    """G-family kernel (include/oflib_hip.h: ofl_warp_bwd_f32).  `src_b`: gather src - src_b (subtracted in the kernel
    where the C ABI supports it, else materialised here).  A uint8 `src` is read as bytes by ofl_warp_bwd_u8 (no float
    copy); with `out_uint8` (round mode ROUND_U8 only: the caller is going to `.to(torch.uint8)` anyway) dst is uint8 too.

    flow [Nf,2,H,W], src [Ns,C,H,W], masks [*,H,W] bool or None, addend [*,C,H,W] or None.
    Returns (dst [N,C,H,W] fp32, valid [N,H,W] bool | None, flow_flags int32[N] | None, src_flags | None),
    all on the HIP device, N = max batch; with `want_dst_flags` (2 channels, valid wanted) a fifth result: the flag
    words int32[N] of dst read as a flow under `valid`.
    """
    lib, dev = load_library(), device(flow, src)
    c, h, w = src.shape[1:]
    n = max(flow.shape[0], src.shape[0], 1 if src_b is None else src_b.shape[0], 1 if src_mask is None else src_mask.shape[0],
            1 if flow_mask is None else flow_mask.shape[0], 1 if addend is None else addend.shape[0])
    if (src.dtype == torch.uint8 and addend is None and src_b is None and not want_flags and not want_dst_flags
            and w >= 4 and h >= 2):
        f, fbs = _planes(flow, dev, torch.float32, n, "flow")
        s8, sbs = _planes(src, dev, torch.uint8, n, "source")
        sm, smbs = (None, 0) if src_mask is None else _planes(src_mask, dev, torch.bool, n, "source mask")
        fm, fmbs = (None, 0) if flow_mask is None else _planes(flow_mask, dev, torch.bool, n, "flow mask")
        as_u8 = bool(out_uint8) and int(round_mode) == ROUND_U8
        dst = torch.empty((n, c, h, w), dtype=torch.uint8 if as_u8 else torch.float32, device=dev)
        valid = torch.empty((n, h, w), dtype=torch.bool, device=dev) if want_valid else None
        rc = lib.ofl_warp_bwd_u8(_ptr(f), fbs, float(flow_sign), _ptr(s8), sbs, _ptr(sm), smbs, _ptr(fm), fmbs,
                                 _ptr(dst), int(as_u8), _ptr(valid), n, c, h, w, int(round_mode), _stream(dev))
        if rc != -4:
            _check(rc, "ofl_warp_bwd_u8")
            return dst, valid, None, None
    f, fbs = _planes(flow, dev, torch.float32, n, "flow")
    s, sbs = _planes(src, dev, torch.float32, n, "source")
    sm, smbs = (None, 0) if src_mask is None else _planes(src_mask, dev, torch.bool, n, "source mask")
    fm, fmbs = (None, 0) if flow_mask is None else _planes(flow_mask, dev, torch.bool, n, "flow mask")
    ad, adbs = (None, 0) if addend is None else _planes(addend, dev, torch.float32, n, "addend")
    dst = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
    valid = torch.empty((n, h, w), dtype=torch.bool, device=dev) if want_valid else None
    ff = torch.zeros(n, dtype=torch.int32, device=dev) if want_flags else None
    sf = torch.zeros(n, dtype=torch.int32, device=dev) if (want_flags and want_src_flags) else None
    df = torch.empty(n, dtype=torch.int32, device=dev) if want_dst_flags else None
    s2, s2bs = (None, 0) if src_b is None else _planes(src_b, dev, torch.float32, n, "source")
    args = lambda: (_ptr(f), fbs, float(flow_sign), _ptr(s), sbs, _ptr(s2), s2bs, _ptr(sm), smbs, _ptr(fm), fmbs,
                    _ptr(ad), adbs, float(a_sign), float(g_sign), _ptr(dst), _ptr(valid), _ptr(ff),
                    _ptr(sf), _ptr(df), n, c, h, w, int(round_mode), _stream(dev))
    rc = lib.ofl_warp_bwd_f32(*args())
    if rc == -4 and s2 is not None:          # the difference is not formed in this kernel variant: materialise it
        s = (src.to(dev, torch.float32) - src_b.to(dev, torch.float32)).expand(n, -1, -1, -1).contiguous()
        sbs, s2, s2bs = c * h * w, None, 0
        rc = lib.ofl_warp_bwd_f32(*args())
    _check(rc, "ofl_warp_bwd_f32")
    if want_dst_flags:
        return dst, valid, ff, sf, df
    return dst, valid, ff, sf


def _splat_fwd_raw(flow, data, *, xs=None, ys=None, flow_sign=1.0, data_sign=1.0, weight_mask=None, chan_mask_a=None,
                   chan_mask_b=None, want_valid=False, occlude=True, want_density=False, want_warped=False,
                   round_mode=ROUND_NONE, want_mask_chan=False, want_dst_flags=False, data_b=None):
    """P-family kernels (ofl_splat_tiled_f32, or ofl_splat_fwd_f32 + ofl_splat_finalize_f32).

    Either flow [Nf,2,H,W] (endpoints computed in-kernel) or explicit positions xs, ys [N,H,W].
    Returns (dst [N,C,H,W], valid | None, density | None, warped | None) on the HIP device; with
    `want_mask_chan` the valid slot holds the warped mask channel itself (fp32) instead of its threshold.
    `data_b` [*,C,H,W], C <= 2: splat data - data_b (one fp32 subtraction in the kernel instead of a materialised difference).
    `want_dst_flags` (2-channel data): a fifth result, the device flag words int32[N] of dst read as a flow under `valid`.
    """
    lib, dev = load_library(), device(flow, data, xs)
    c, h, w = data.shape[1:]
    if want_dst_flags and c != 2:
        raise ValueError("oflibpytorch_amd: output flags are defined for 2-channel data only")
    n = max(data.shape[0], 1 if data_b is None else data_b.shape[0], 1 if flow is None else flow.shape[0], 1 if xs is None else xs.shape[0],
            1 if weight_mask is None else weight_mask.shape[0], 1 if chan_mask_a is None else chan_mask_a.shape[0],
            1 if chan_mask_b is None else chan_mask_b.shape[0])
    f, fbs = (None, 0) if flow is None else _planes(flow, dev, torch.float32, n, "flow")
    x, xbs = (None, 0) if xs is None else _planes(xs, dev, torch.float32, n, "x")
    y, ybs = (None, 0) if ys is None else _planes(ys, dev, torch.float32, n, "y")
    if x is not None and xbs != ybs:
        x, y = x.expand(n, h, w).contiguous(), y.expand(n, h, w).contiguous()
        xbs = ybs = h * w
    d, dbs = _planes(data, dev, torch.float32, n, "data")
    d2, d2bs = (None, 0) if data_b is None else _planes(data_b, dev, torch.float32, n, "data")
    wm, wmbs = (None, 0) if weight_mask is None else _planes(weight_mask, dev, torch.bool, n, "mask")
    ca, cabs = (None, 0) if chan_mask_a is None else _planes(chan_mask_a, dev, torch.bool, n, "mask")
    cb, cbbs = (None, 0) if chan_mask_b is None else _planes(chan_mask_b, dev, torch.bool, n, "mask")
    mch = 1 if (want_valid or want_mask_chan) else 0
    occ = 1 if (occlude and f is not None) else 0
    st = _stream(dev)
    dst = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
    valid = torch.empty((n, h, w), dtype=torch.bool, device=dev) if (want_valid and not want_mask_chan) else None
    mchan = torch.empty((n, h, w), dtype=torch.float32, device=dev) if want_mask_chan else None
    density = torch.empty((n, h, w), dtype=torch.float32, device=dev) if want_density else None
    warped = torch.empty((n, h, w), dtype=torch.bool, device=dev) if want_warped else None
    dflags = torch.empty((n,), dtype=torch.int32, device=dev) if want_dst_flags else None
    rc = -4
    if _splat_path != 1 and w >= 4:
        # fused tiled path: LDS accumulation per destination tile; `accum` is only touched if the flow is too rough
        ws = torch.empty(int(lib.ofl_splat_tiled_workspace_ints(n, h, w)), dtype=torch.int32, device=dev)
        accum = _fallback_accum(lib, n, c, mch, h, w, dev)
        rc = lib.ofl_splat_tiled_f32(_ptr(f), fbs, float(flow_sign), _ptr(x), _ptr(y), xbs, _ptr(d), dbs,
                                     float(data_sign), _ptr(d2), d2bs, _ptr(wm), wmbs, _ptr(ca), cabs, _ptr(cb), cbbs, mch, occ,
                                     _ptr(dst), _ptr(density), _ptr(warped), _ptr(valid), _ptr(mchan), _ptr(dflags),
                                     _ptr(ws), ws.numel(), _ptr(accum), n, c, h, w, int(round_mode), st)
        if rc not in (0, -4):
            _check(rc, "ofl_splat_tiled_f32")
        if collect_splat_stats:             # (a copy: a view would keep the whole workspace alive between calls)
            global _last_splat_stats
            _last_splat_stats = ws[:8].clone()   # [launch fell back to global atomics, tiles that left the exact path, -, -]
            if collect_splat_stats == 2:         # tools/splat_list_stats.py: the whole workspace (list lengths)
                global _last_splat_ws
                _last_splat_ws = ws
    if rc == -4:   # not eligible (alignment / channels): the general two-pass path
        if d2 is not None:
            d = (data.to(dev, torch.float32) - data_b.to(dev, torch.float32)).expand(n, -1, -1, -1).contiguous()
            dbs = c * h * w
        accum = torch.zeros((n, 1 + c + mch, h, w), dtype=torch.float32, device=dev)
        _check(lib.ofl_splat_fwd_f32(_ptr(f), fbs, float(flow_sign), _ptr(x), _ptr(y), xbs, _ptr(d), dbs,
                                     float(data_sign), _ptr(wm), wmbs, _ptr(ca), cabs, _ptr(cb), cbbs, mch, occ,
                                     _ptr(accum), n, c, h, w, st), "ofl_splat_fwd_f32")
        _check(lib.ofl_splat_finalize_f32(_ptr(accum), _ptr(f), fbs, _ptr(d), dbs, float(data_sign), _ptr(wm), wmbs,
                                          _ptr(ca), cabs, _ptr(cb), cbbs, mch, occ, _ptr(dst), _ptr(density),
                                          _ptr(warped), _ptr(valid), _ptr(mchan), n, c, h, w, int(round_mode), st),
               "ofl_splat_finalize_f32")
        if want_dst_flags:
            dflags = flow_flags(dst, valid)
    if want_dst_flags:
        return dst, (mchan if want_mask_chan else valid), density, warped, dflags
    return dst, (mchan if want_mask_chan else valid), density, warped


# ------------------------------------------------------------------------------------------------
# backward passes, point sampler, extents (csrc/ofl_aux_kernels.hip)
# ------------------------------------------------------------------------------------------------
def warp_bwd_grad(flow, src, grad_out, *, flow_sign=1.0, g_scale=1.0, want_src=True, want_flow=True):
    """Backward of G (ofl_warp_bwd_grad_f32).  flow [Nf,2,H,W], src [Ns,C,H,W] (the field that was gathered), grad_out
    [N,C,H,W] -> (grad_src [Ns,C,H,W] | None, grad_flow [N,2,H,W] | None); a broadcast source accumulates into its one image."""
    lib, dev = load_library(), device(flow, src, grad_out)
    n, c, h, w = grad_out.shape
    with _on(dev):
        f, fbs = _planes(flow.detach(), dev, torch.float32, n, "flow")
        s, sbs = _planes(src.detach(), dev, torch.float32, n, "source")
        g = grad_out.detach().to(dev, torch.float32).contiguous()
        ns = 1 if sbs == 0 else n
        gs = None
        if want_src:
            # grad wrt the source = the un-normalised forward splat of the upstream gradient along the negated flow: the gather
            # kernels instead of 4 * C global float atomics per pixel (ofl_splat_sum_f32; B=16 1080p C=3: 7.1 -> 1.0 ms)
            full = splat_sum(f if fbs != 0 or n == 1 else f.expand(n, -1, -1, -1), g, flow_sign=-float(flow_sign), data_sign=float(g_scale))
            if full is not None:
                gs = full.sum(0, keepdim=True) if (ns == 1 and n > 1) else full
        atomics = want_src and gs is None                   # shapes the gather splat does not take (W < 4 ...)
        if atomics:
            gs = torch.zeros((ns, c, h, w), dtype=torch.float32, device=dev)
        gf = torch.empty((n, 2, h, w), dtype=torch.float32, device=dev) if want_flow else None
        if atomics or want_flow:
            _check(lib.ofl_warp_bwd_grad_f32(_ptr(f), fbs, float(flow_sign), _ptr(s), sbs, _ptr(g), float(g_scale),
                                             _ptr(gs) if atomics else None, 0 if ns == 1 else c * h * w, _ptr(gf), n, c, h, w,
                                             _stream(dev)),
                   "ofl_warp_bwd_grad_f32")
    return gs, gf


def splat_sum(flow, data, *, flow_sign=1.0, data_sign=1.0):
    """ofl_splat_sum_f32: the weighted sums of the forward splat of data [N,C,H,W] along flow_sign * flow [N|1,2,H,W], NOT divided
    by the density (every pixel contributes) -- the transpose of the backward warp.  None for shapes the gather splat does not
    take (W < 4)."""
    n, c, h, w = data.shape
    if w < 4 or h * w >= (1 << 24) or h >= 32768 or w >= 32768:
        return None              # frames the gather splat does not take (its own limit, utils.py:1118): the caller's atomics kernel does
    lib, dev = load_library(), device(flow, data)
    with _on(dev):
        f, fbs = _planes(flow.detach(), dev, torch.float32, n, "flow")
        d = data.detach().to(dev, torch.float32).contiguous()
        ws = torch.empty(int(lib.ofl_splat_tiled_workspace_ints(n, h, w)), dtype=torch.int32, device=dev)
        accum = _fallback_accum(lib, n, c, 0, h, w, dev)
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        rc = lib.ofl_splat_sum_f32(_ptr(f), fbs, float(flow_sign), _ptr(d), c * h * w, float(data_sign), _ptr(out),
                                   _ptr(ws), ws.numel(), _ptr(accum), n, c, h, w, _stream(dev))
        if rc in (-4, -2):       # not eligible / shape beyond the gather splat's limits
            return None
        _check(rc, "ofl_splat_sum_f32")
        if collect_splat_stats:
            global _last_splat_stats
            _last_splat_stats = ws[:8].clone()
    return out


def splat_grad(flow, data, out, density, grad_out, *, xs=None, ys=None, flow_sign=1.0, weight_mask=None, occlude=True,
               grad_density=None, want_data=True, want_xy=True):
    """Backward of P (ofl_splat_grad_f32).  `data` is the data actually splatted.  -> (grad_data [N,C,H,W] | None,
    grad_xy [N,2,H,W] | None) with grad_xy = (d/dx, d/dy) of the end points."""
    lib, dev = load_library(), device(flow, data, xs, grad_out)
    n, c, h, w = grad_out.shape
    with _on(dev):
        f, fbs = (None, 0) if flow is None else _planes(flow.detach(), dev, torch.float32, n, "flow")
        x, xbs = (None, 0) if xs is None else _planes(xs.detach(), dev, torch.float32, n, "x")
        y, ybs = (None, 0) if ys is None else _planes(ys.detach(), dev, torch.float32, n, "y")
        if x is not None and xbs != ybs:
            x, y = x.expand(n, h, w).contiguous(), y.expand(n, h, w).contiguous()
            xbs = h * w
        d, dbs = _planes(data.detach(), dev, torch.float32, n, "data")
        wm, wmbs = (None, 0) if weight_mask is None else _planes(weight_mask, dev, torch.bool, n, "mask")
        o = out.detach().to(dev, torch.float32).contiguous()
        den = density.detach().to(dev, torch.float32).contiguous()
        g = grad_out.detach().to(dev, torch.float32).contiguous()
        gden = None if grad_density is None else grad_density.detach().to(dev, torch.float32).contiguous()
        gd = torch.empty((n, c, h, w), dtype=torch.float32, device=dev) if want_data else None
        gxy = None
        occ = 1 if (occlude and f is not None) else 0
        for c0 in range(0, c, 3):                       # (3 channels + the density term: one 16-byte slot per destination pixel)
            c1 = min(c0 + 3, c)
            part = torch.empty((n, 2, h, w), dtype=torch.float32, device=dev) if want_xy else None
            dd = d[:, c0:c1]
            if c1 - c0 != c:
                dd = dd.contiguous()
            ddbs = 0 if dbs == 0 else dd.stride(0)
            oo, gg = (o, g) if c1 - c0 == c else (o[:, c0:c1].contiguous(), g[:, c0:c1].contiguous())
            gdd = None if gd is None else (gd if c1 - c0 == c else torch.empty((n, c1 - c0, h, w), dtype=torch.float32, device=dev))
            scratch = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
            _check(lib.ofl_splat_grad_f32(_ptr(f), fbs, float(flow_sign), _ptr(x), _ptr(y), xbs, _ptr(dd), ddbs, _ptr(wm), wmbs,
                                          occ, _ptr(oo), _ptr(den), _ptr(gg), _ptr(gden if c0 == 0 else None), _ptr(scratch), _ptr(gdd),
                                          _ptr(part), n, c1 - c0, h, w, _stream(dev)), "ofl_splat_grad_f32")
            if gd is not None and gdd is not gd:
                gd[:, c0:c1] = gdd
            if part is not None:
                gxy = part if gxy is None else gxy + part
    return gd, gxy


def sample_pts(flow, pts):
    """track_pts' sampler (ofl_sample_pts_f32): flow [N|1,2,H,W], pts [N|1,M,2] float (y, x) -> pts + sampled flow [N,M,2]."""
    lib, dev = load_library(), device(flow, pts)
    n = max(flow.shape[0], pts.shape[0])
    m = pts.shape[1]
    h, w = flow.shape[2:]
    with _on(dev):
        f, fbs = _planes(flow.detach(), dev, torch.float32, n, "flow")
        q, qbs = _planes(pts.detach(), dev, torch.float32, n, "points")
        out = torch.empty((n, m, 2), dtype=torch.float32, device=dev)
        if m > 0:
            _check(lib.ofl_sample_pts_f32(_ptr(f), fbs, _ptr(q), qbs, _ptr(out), n, m, h, w, _stream(dev)), "ofl_sample_pts_f32")
    return out


def sample_pts_grad(flow, pts, grad_out, *, want_flow=True, want_pts=True):
    """Backward of `sample_pts` -> (grad_flow [N,2,H,W] | None, grad_pts [N,M,2] | None)."""
    lib, dev = load_library(), device(flow, pts, grad_out)
    n, m = grad_out.shape[:2]
    h, w = flow.shape[2:]
    with _on(dev):
        f, fbs = _planes(flow.detach(), dev, torch.float32, n, "flow")
        q, qbs = _planes(pts.detach(), dev, torch.float32, n, "points")
        g = grad_out.detach().to(dev, torch.float32).contiguous()
        gf = torch.zeros((n, 2, h, w), dtype=torch.float32, device=dev) if want_flow else None
        gp = torch.empty((n, m, 2), dtype=torch.float32, device=dev) if want_pts else None
        if m > 0:
            _check(lib.ofl_sample_pts_grad_f32(_ptr(f), fbs, _ptr(q), qbs, _ptr(g), _ptr(gf), _ptr(gp), n, m, h, w,
                                               _stream(dev)), "ofl_sample_pts_grad_f32")
    return gf, gp


def flow_from_matrix(matrix: torch.Tensor, n: int, h: int, w: int, sign: float = 1.0) -> torch.Tensor:
    """ofl_flow_from_matrix_f32: the flow field [n,2,h,w] of 3 x 3 matrices [n|1,3,3] (utils.py:339-376), negated when sign = -1,
    generated on the HIP device of the matrix (else torch's current device) -- write-only, 8 B/px."""
    lib, dev = load_library(), device(matrix)
    with _on(dev):
        m = matrix.detach().to(dev, torch.float32).reshape(-1, 9).contiguous()
        if m.shape[0] not in (1, n):
            raise ValueError("oflibpytorch_amd: %d matrices cannot broadcast to a batch of %d" % (m.shape[0], n))
        dst = torch.empty((n, 2, h, w), dtype=torch.float32, device=dev)
        _check(lib.ofl_flow_from_matrix_f32(_ptr(m), 0 if m.shape[0] == 1 else 9, float(sign), _ptr(dst), n, h, w, _stream(dev)),
               "ofl_flow_from_matrix_f32")
    return dst


def flag_words_or(words: torch.Tensor) -> torch.Tensor:
    """int32[N] flag words on a HIP device -> int32[N + 5] there: the words, then their OR over the batch as five 0 / 1
    integers, one per bit (ofl_flag_words_or_i32) -- what `distributed.with_global_or` all-reduces."""
    lib, dev = load_library(), words.device
    n = int(words.numel())
    with _on(dev):
        src = words.to(torch.int32).contiguous()
        out = torch.empty(n + 5, dtype=torch.int32, device=dev)
        _check(lib.ofl_flag_words_or_i32(_ptr(src), n, _ptr(out), _stream(dev)), "ofl_flag_words_or_i32")
    return out


def flow_extents(vecs, mask, sign: float) -> torch.Tensor:
    """Flow.get_padding's reduction (ofl_flow_extents_f32) -> fp32 [N,5] on the HIP device: min y, max y, min x, max x of the
    positions -(sign * thr(v) - grid) under the mask, and whether any pixel was valid."""
    lib, dev = load_library(), device(vecs, mask)
    n, _, h, w = vecs.shape
    with _on(dev):
        v, vbs = _planes(vecs.detach(), dev, torch.float32, n, "flow")
        m, mbs = (None, 0) if mask is None else _planes(mask, dev, torch.bool, n, "mask")
        ws = torch.empty(5 * n, dtype=torch.int32, device=dev)
        ext = torch.empty((n, 5), dtype=torch.float32, device=dev)
        _check(lib.ofl_flow_extents_f32(_ptr(v), vbs, _ptr(m), mbs, float(sign), _ptr(ws), _ptr(ext), n, h, w, _stream(dev)),
               "ofl_flow_extents_f32")
    return ext


def resize_bilinear(x: torch.Tensor, scale) -> torch.Tensor:
    """ofl_resize_bilinear_f32: F.interpolate(x [N,C,H,W], scale_factor=[sh, sw], mode='bilinear', align_corners=False) with the
    arithmetic of ATen's CPU kernels (bit-identical to the reference's PyTorch-CPU resize), on x's HIP device."""
    import math
    lib, dev = load_library(), device(x)
    n, c, h, w = x.shape
    sh, sw = float(scale[0]), float(scale[1])
    oh, ow = int(math.floor(float(h) * sh)), int(math.floor(float(w) * sw))          # torch/nn/functional.py: floor of the double product
    if oh < 1 or ow < 1:
        raise RuntimeError("Input and output sizes should be greater than 0, but got input (H: %d, W: %d) output (H: %d, W: %d)" % (h, w, oh, ow))
    with _on(dev):
        src = x.detach().to(dev, torch.float32).contiguous()
        dst = torch.empty((n, c, oh, ow), dtype=torch.float32, device=dev)
        planes = n * c
        for p0 in range(0, planes, 65535):          # (the grid's y dimension)
            k = min(65535, planes - p0)
            _check(lib.ofl_resize_bilinear_f32(src.data_ptr() + 4 * p0 * h * w, dst.data_ptr() + 4 * p0 * oh * ow, k, h, w, oh, ow,
                                               float(np.float32(1.0 / sh)), float(np.float32(1.0 / sw)), _stream(dev)),
                   "ofl_resize_bilinear_f32")
    return dst


def warp_valid(flow, mask, flow_sign: float = 1.0, thr: float = 0.9999) -> torch.Tensor:
    """ofl_warp_valid_f32: (an all-ones image warped backward along flow_sign * flow > thr) & mask -> bool [N,H,W] -- the 't'
    branch of Flow.valid_target / the 's' branch of Flow.valid_source (flow_class.py:1119-1122, 1151-1157) in one launch."""
    lib, dev = load_library(), device(flow, mask)
    n, _, h, w = flow.shape
    with _on(dev):
        f, fbs = _planes(flow.detach(), dev, torch.float32, n, "flow")
        m, mbs = (None, 0) if mask is None else _planes(mask, dev, torch.bool, n, "mask")
        out = torch.empty((n, h, w), dtype=torch.bool, device=dev)
        _check(lib.ofl_warp_valid_f32(_ptr(f), fbs, float(flow_sign), _ptr(m), mbs, float(thr), _ptr(out), n, h, w, _stream(dev)),
               "ofl_warp_valid_f32")
    return out


# ------------------------------------------------------------------------------------------------
# padded apply: the flow covers a window of the target's frame (no padded copies of the flow)
# ------------------------------------------------------------------------------------------------
def warp_bwd_win(flow, src, window, *, src_mask=None, flow_mask=None, want_valid=False, round_mode=ROUND_NONE):
    """ofl_warp_bwd_win_f32: 't' flow [Nf,2,fh,fw] placed at (top, left) = `window` inside the frame of src [Ns,C,H,W]; zero
    flow / False mask outside.  -> (dst [N,C,H,W] fp32, valid [N,H,W] bool | None)"""
    lib, dev = load_library(), device(flow, src)
    c, h, w = src.shape[1:]
    fh, fw = flow.shape[2:]
    n = max(flow.shape[0], src.shape[0], 1 if src_mask is None else src_mask.shape[0], 1 if flow_mask is None else flow_mask.shape[0])
    with _on(dev):
        f, fbs = _planes(flow.detach(), dev, torch.float32, n, "flow")
        s, sbs = _planes(src.detach(), dev, torch.float32, n, "source")
        sm, smbs = (None, 0) if src_mask is None else _planes(src_mask, dev, torch.bool, n, "source mask")
        fm, fmbs = (None, 0) if flow_mask is None else _planes(flow_mask, dev, torch.bool, n, "flow mask")
        dst = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        valid = torch.empty((n, h, w), dtype=torch.bool, device=dev) if want_valid else None
        _check(lib.ofl_warp_bwd_win_f32(_ptr(f), fbs, 1.0, fh, fw, int(window[0]), int(window[1]), _ptr(s), sbs, _ptr(sm), smbs,
                                        _ptr(fm), fmbs, _ptr(dst), _ptr(valid), n, c, h, w, int(round_mode), _stream(dev)),
               "ofl_warp_bwd_win_f32")
    return dst, valid


def splat_fwd_win(flow, data, window, *, weight_mask=None, chan_mask_a=None, chan_mask_b=None, want_valid=False, occlude=True,
                  round_mode=ROUND_NONE):
    """ofl_splat_tiled_win_f32: 's' flow [Nf,2,fh,fw] placed at (top, left) = `window` inside the frame of data [Nd,C,H,W]
    (replicated outside, masks False outside).  -> (dst [N,C,H,W] fp32, valid [N,H,W] bool | None); None when the frame is
    not eligible for the gather path (W < 4): the caller pads and takes the plain route."""
    lib, dev = load_library(), device(flow, data)
    c, h, w = data.shape[1:]
    fh, fw = flow.shape[2:]
    if w < 4 or _splat_path == 1:
        return None
    n = max(data.shape[0], flow.shape[0], 1 if weight_mask is None else weight_mask.shape[0],
            1 if chan_mask_a is None else chan_mask_a.shape[0], 1 if chan_mask_b is None else chan_mask_b.shape[0])
    with _on(dev):
        f, fbs = _planes(flow.detach(), dev, torch.float32, n, "flow")
        d, dbs = _planes(data.detach(), dev, torch.float32, n, "data")
        wm, wmbs = (None, 0) if weight_mask is None else _planes(weight_mask, dev, torch.bool, n, "mask")
        ca, cabs = (None, 0) if chan_mask_a is None else _planes(chan_mask_a, dev, torch.bool, n, "mask")
        cb, cbbs = (None, 0) if chan_mask_b is None else _planes(chan_mask_b, dev, torch.bool, n, "mask")
        mch = 1 if want_valid else 0
        dst = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        valid = torch.empty((n, h, w), dtype=torch.bool, device=dev) if want_valid else None
        ws = torch.empty(int(lib.ofl_splat_tiled_workspace_ints(n, h, w)), dtype=torch.int32, device=dev)
        accum = _fallback_accum(lib, n, c, mch, h, w, dev)
        rc = lib.ofl_splat_tiled_win_f32(_ptr(f), fbs, 1.0, fh, fw, int(window[0]), int(window[1]), _ptr(d), dbs, 1.0, _ptr(wm), wmbs,
                                         _ptr(ca), cabs, _ptr(cb), cbbs, mch, 1 if occlude else 0, _ptr(dst), None, None,
                                         _ptr(valid), None, _ptr(ws), ws.numel(), _ptr(accum), n, c, h, w, int(round_mode),
                                         _stream(dev))
        if rc == -4:
            return None
        _check(rc, "ofl_splat_tiled_win_f32")
        if collect_splat_stats:
            global _last_splat_stats
            _last_splat_stats = ws[:8].clone()
    return dst, valid
