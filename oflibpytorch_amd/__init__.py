"""oflibpytorch_amd -- MI355X-native (gfx950) dense optical-flow warping & composition.

Drop-in for the warp / compose hot path of oflibpytorch (`Flow.apply`, `Flow.combine_with` /
`combine_flows`, `Flow.switch_ref` / `invert`, `apply_flow`, `apply_s_flow`,
`grid_from_unstructured_data`): the reference's Python surface (reference `__init__.py:14-18`) over
hand-written HIP kernels.  No CPU fallback: see `_native.NativeUnavailable`.
"""
from .flow_class import Flow, set_revalidate_every_call, get_revalidate_every_call
from .flow_operations import (combine_flows, switch_flow_ref, invert_flow, valid_target, valid_source, batch_flows,
                              get_flow_padding)
from .utils import (from_matrix, from_transforms, resize_flow, apply_flow, is_zero_flow, get_pure_pytorch,
                    set_pure_pytorch, unset_pure_pytorch, to_numpy, to_tensor, move_axis, apply_s_flow,
                    grid_from_unstructured_data, get_flow_endpoints, threshold_vectors, normalise_coords, track_pts,
                    set_half_flow_outputs, get_half_flow_outputs)
from ._native import NativeUnavailable

__version__ = "0.1.0"
