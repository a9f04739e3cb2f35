"""CPU tier: the C-ABI library builds, loads, exports every symbol include/oflib_hip.h declares, and rejects bad
arguments before touching a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

from oflibpytorch_amd import _build, _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_loads():
    path = _build.build()
    assert os.path.exists(path)
    lib = _native.load_library()
    assert lib.ofl_version() >= 10


def test_header_and_library_agree_on_symbols():
    header = open(os.path.join(ROOT, 'include', 'oflib_hip.h')).read()
    declared = set(re.findall(r'^(?:int|int64_t|const char\*) (ofl_\w+)\(', header, flags=re.M))
    assert declared == set(_native.exported_symbols())
    lib = ctypes.CDLL(_build.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_argument_validation_needs_no_gpu():
    lib = _native.load_library()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)      # never dereferenced: rejected on shape / argument checks first
    # NULL required pointers
    assert lib.ofl_warp_bwd_f32(null, 0, 1.0, null, 0, null, 0, null, 0, null, 0, null, 0, 1.0, 1.0, null, null, null, null, null,
                                1, 1, 4, 4, 0, null) == -1
    # bad dims; h*w >= 2^24 is the forward splat's limit only (utils.py:1118: fp32 position index)
    assert lib.ofl_warp_bwd_f32(one, 0, 1.0, one, 0, null, 0, null, 0, null, 0, null, 0, 1.0, 1.0, one, null, null, null, null,
                                0, 1, 4, 4, 0, null) == -2
    assert lib.ofl_splat_fwd_f32(one, 0, 1.0, null, null, 0, one, 0, 1.0, null, 0, null, 0, null, 0, 0, 0, one,
                                 1, 1, 4096, 4096, null) == -2
    assert lib.ofl_warp_bwd_u8(one, 0, 1.0, one, 0, null, 0, null, 0, one, 0, null, 1, 1, 4096, 4096, 0, null) == -4
    # the new entry points: NULL / shape / argument checks
    assert lib.ofl_warp_bwd_grad_f32(null, 0, 1.0, null, 0, null, 1.0, null, 0, null, 1, 1, 4, 4, null) == -1
    assert lib.ofl_warp_bwd_grad_f32(one, 0, 1.0, one, 0, one, 1.0, null, 0, null, 1, 1, 4, 4, null) == -3
    assert lib.ofl_splat_grad_f32(one, 0, 1.0, null, null, 0, one, 0, null, 0, 1, one, one, one, null, one, one, null,
                                  1, 4, 4, 4, null) == -4
    assert lib.ofl_splat_sum_f32(null, 0, 1.0, one, 0, 1.0, one, one, 8, one, 1, 1, 4, 4, null) == -1
    assert lib.ofl_sample_pts_f32(one, 0, one, 0, one, 1, 0, 4, 4, null) == -2
    assert lib.ofl_flow_extents_f32(one, 0, null, 0, 0.5, one, one, 1, 4, 4, null) == -3
    assert lib.ofl_flag_words_or_i32(null, 1, one, null) == -1 and lib.ofl_flag_words_or_i32(one, 0, one, null) == -2
    # flow_sign must be +-1, round mode 0..2
    assert lib.ofl_warp_bwd_f32(one, 0, 0.5, one, 0, null, 0, null, 0, null, 0, null, 0, 1.0, 1.0, one, null, null, null, null,
                                1, 1, 4, 4, 0, null) == -3
    assert lib.ofl_warp_bwd_f32(one, 0, 1.0, one, 0, null, 0, null, 0, null, 0, null, 0, 1.0, 1.0, one, null, null, null, null,
                                1, 1, 4, 4, 7, null) == -3
    # dst_flags: two channels and a valid mask only
    assert lib.ofl_warp_bwd_f32(one, 0, 1.0, one, 0, null, 0, null, 0, null, 0, null, 0, 1.0, 1.0, one, one, null, null, one,
                                1, 3, 4, 4, 0, null) == -3
    assert lib.ofl_warp_bwd_f32(one, 0, 1.0, one, 0, null, 0, null, 0, null, 0, null, 0, 1.0, 1.0, one, null, null, null, one,
                                1, 2, 4, 4, 0, null) == -3
    assert lib.ofl_flow_flags_f32(null, 0, null, 0, 1e-3, null, 1, 4, 4, null) == -1
    assert lib.ofl_splat_fwd_f32(null, 0, 1.0, null, null, 0, null, 0, 1.0, null, 0, null, 0, null, 0, 0, 0, null,
                                 1, 1, 4, 4, null) == -1
    assert lib.ofl_set_option(1, 8) == -3 and lib.ofl_set_option(1, 7) == 0 and lib.ofl_set_option(1, 6) == 0 and lib.ofl_set_option(1, 5) == 0 and lib.ofl_set_option(1, 0) == 0      # (5: more than 3 channels as launches of 3; 6: the sheared rectangle instead of per-row extents; 7: four-tile row-table columns whatever the size)


def test_round3_entry_points_reject_bad_arguments_without_a_gpu():
    lib = _native.load_library()
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    # frames of 2^24 pixels and more are beyond the gather splat (utils.py:1118) -- a shape error from the C ABI, and
    # `_native.splat_sum` steps aside (None) so that the warp's backward takes its atomics kernel instead of raising
    assert lib.ofl_splat_sum_f32(one, 0, 1.0, one, 0, 1.0, one, one, 1 << 30, one, 1, 1, 4096, 4096, null) == -2
    import torch
    assert _native.splat_sum(torch.zeros(1, 2, 4096, 4096), torch.zeros(1, 1, 4096, 4096)) is None
    assert _native.splat_sum(torch.zeros(1, 2, 8, 3), torch.zeros(1, 1, 8, 3)) is None
    # validation read-back: required pointers, threshold, 8-byte aligned host words
    assert lib.ofl_flow_flags_host(null, 0, 0, null, 0, 1e-3, one, one, 1, 1, 4, 4, null) == -1
    assert lib.ofl_flow_flags_host(one, 0, 0, null, 0, 0.5, one, one, 1, 1, 4, 4, null) == -3
    assert lib.ofl_flow_flags_host(one, 0, 0, null, 0, 1e-3, one, ctypes.c_void_p(20), 1, 1, 4, 4, null) == -3
    assert lib.ofl_flow_flags_host(one, 1, 0, null, 0, 1e-3, one, one, 1, 1, 5, 7, null) == -4          # fp16 layout the vector kernel does not take
    assert lib.ofl_host_words_alloc(0, ctypes.byref(ctypes.c_void_p())) == -3 and lib.ofl_host_words_free(null) == 0
    # generators
    assert lib.ofl_flow_from_matrix_f32(null, 9, 1.0, one, 1, 4, 4, null) == -1
    assert lib.ofl_flow_from_matrix_f32(one, 9, 0.5, one, 1, 4, 4, null) == -3
    assert lib.ofl_flow_from_matrix_f32(one, 9, 1.0, one, 0, 4, 4, null) == -2
    # round 4: valid area of a backward warp, bit-exact resize
    assert lib.ofl_warp_valid_f32(null, 0, 1.0, null, 0, 0.9999, one, 1, 4, 4, null) == -1
    assert lib.ofl_warp_valid_f32(one, 0, 0.5, null, 0, 0.9999, one, 1, 4, 4, null) == -3
    assert lib.ofl_warp_valid_f32(one, 0, 1.0, null, 0, 0.9999, one, 0, 4, 4, null) == -2
    assert lib.ofl_resize_bilinear_f32(null, one, 1, 4, 4, 8, 8, 0.5, 0.5, null) == -1
    assert lib.ofl_resize_bilinear_f32(one, one, 1, 4, 4, 0, 8, 0.5, 0.5, null) == -2
    assert lib.ofl_resize_bilinear_f32(one, one, 70000, 4, 4, 8, 8, 0.5, 0.5, null) == -2
    assert lib.ofl_resize_bilinear_f32(one, one, 1, 4, 4, 8, 8, 0.0, 0.5, null) == -3
    # the fallback accumulator is bounded: the pass, capped at 1 GiB
    assert lib.ofl_splat_tiled_fallback_images(64, 5, 1080, 1920) == 25 and lib.ofl_splat_tiled_fallback_images(4, 5, 1080, 1920) == 4
    assert lib.ofl_splat_tiled_fallback_images(16, 4, 2160, 3840) == 8 and lib.ofl_splat_tiled_fallback_images(3, 5, 16384, 16384) == 1
    tw, th, cap = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.ofl_splat_tile_geometry(ctypes.byref(tw), ctypes.byref(th), ctypes.byref(cap)) == 0
    assert (tw.value, th.value) in ((32, 16), (64, 16)) and cap.value == 256 * (tw.value // 32)
    assert lib.ofl_splat_tile_geometry(None, None, None) == 0
    assert lib.ofl_set_option(5, -1) == -3 and lib.ofl_set_option(5, 0) == 0


def test_no_gpu_means_loud_failure(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    with pytest.raises(_native.NativeUnavailable):
        _native.device()
    import oflibpytorch_amd as ofl
    with pytest.raises(_native.NativeUnavailable):      # validation itself is a HIP reduction: no silent CPU path
        f = ofl.Flow(torch.zeros(1, 2, 8, 8) + 1.0)
        f.apply(torch.zeros(1, 1, 8, 8))
    with pytest.raises(_native.NativeUnavailable):
        ofl.apply_flow(torch.ones(1, 2, 8, 8), torch.zeros(1, 1, 8, 8), 't')
