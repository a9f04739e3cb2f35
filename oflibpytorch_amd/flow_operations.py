"""Tensor-level wrappers round the `Flow` methods of the hot path (reference
``src/oflibpytorch/flow_operations.py:84-277, 458-483``): same signatures, 3-D in -> 3-D out."""
from typing import Union

import numpy as np
import torch

from .flow_class import Flow
from .utils import get_valid_ref

FlowAlias = 'Flow'


def _unwrap(result: Flow, like) -> torch.Tensor:
    return result.vecs if len(like.shape) > 3 else result.vecs.squeeze(0)


def combine_flows(input_1, input_2, mode: int, ref: str = None, thresholded: bool = None):
    """flow_1 (+) flow_2 = flow_3; `mode` (1, 2 or 3) names the unknown (flow_operations.py:84-188).
    Arrays / tensors of shape (N-)2-H-W or (N-)H-W-2 in, torch tensor (N-)2-H-W out."""
    if isinstance(input_1, Flow) and isinstance(input_2, Flow):
        print("AVOID - future deprecation warning: using combine_flows(flow_obj1, flow_obj2) is deprecated and may "
              "not work anymore in future versions - use flow_obj1.combine_with(flow_obj2) instead. combine_flows() "
              "will be reserved for use with Torch tensors and NumPy arrays only.")
        return input_1.combine_with(input_2, mode=mode, thresholded=thresholded)
    # deferred validation: the composition kernel reports finiteness / zero flags of both operands as a by-product
    result = Flow._deferred(input_1, ref).combine_with(Flow._deferred(input_2, ref), mode=mode, thresholded=thresholded)
    return _unwrap(result, input_1)


def switch_flow_ref(flow, input_ref: str) -> torch.Tensor:
    """flow_operations.py:191-207"""
    return _unwrap(Flow(flow, input_ref).switch_ref(), flow)


def invert_flow(flow, input_ref: str, output_ref: str = None) -> torch.Tensor:
    """flow_operations.py:210-228"""
    output_ref = input_ref if output_ref is None else output_ref
    return _unwrap(Flow(flow, input_ref).invert(output_ref), flow)


def valid_target(flow, ref: str, consider_mask: bool = None) -> torch.Tensor:
    """flow_operations.py:231-254"""
    area = Flow(flow, ref).valid_target(consider_mask)
    return area if len(flow.shape) > 3 else area.squeeze(0)


def valid_source(flow, ref: str, consider_mask: bool = None) -> torch.Tensor:
    """flow_operations.py:257-277"""
    area = Flow(flow, ref).valid_source(consider_mask)
    return area if len(flow.shape) > 3 else area.squeeze(0)


def get_flow_padding(flow, ref: str) -> list:
    """[top, bottom, left, right] per batch member (one list for 3-D input): Flow(flow, ref).get_padding()
    (flow_operations.py:280-303)"""
    p = Flow(flow, ref).get_padding()
    return p if len(flow.shape) > 3 else p[0]


def batch_flows(flows: Union[list, tuple]) -> FlowAlias:
    """Concatenate flow objects of equal H, W, ref and device along the batch axis (flow_operations.py:458-483)"""
    if not isinstance(flows, (list, tuple)):
        raise TypeError("Error batching flows: Input needs to be a tuple or a list of flow objects")
    if not all(isinstance(f, Flow) for f in flows):
        raise TypeError("Error batching flows: Input needs to be a tuple or a list of flow objects")
    if len({f.shape[1:] for f in flows}) != 1:
        raise ValueError("Error batching flows: Flow objects need to have the same H-W shape")
    if len({f.ref for f in flows}) != 1:
        raise ValueError("Error batching flows: Flow objects need to have the same reference")
    if len({str(f.device) for f in flows}) != 1:
        raise ValueError("Error batching flows: Flow objects need to be on the same device")
    return Flow(torch.cat([f.vecs for f in flows], dim=0), flows[0].ref, torch.cat([f.mask for f in flows], dim=0))
