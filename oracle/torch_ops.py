"""The reference's torch-CPU OP SEQUENCE for the two operations of the headline metric -- TEST / BENCH INFRASTRUCTURE ONLY.

`bench.py`'s cpu_baseline leg times this next to the HIP path ("the same torch-CPU op sequence as
the reference -- F.grid_sample(align_corners=True), validation passes"), and `tests/test_oracle_golden.py` pins it against
the fixtures the imported reference produced.  Nothing in `oflibpytorch_amd/` imports it.

Unlike `ofl_oracle.c` (a scalar C restatement of the arithmetic), this file restates WHICH ATen ops the reference runs
and in which order, including its validation passes, so that its wall time on the host cores is what the reference would
spend there:

  Flow(...)            get_valid_vecs: isfinite().all()                      utils.py:98
                       get_valid_mask: ((m != 0) & (m != 1)).any()           utils.py:174
  Flow.apply 't'       cat mask channel -> apply_flow -> > 0.99999 -> & mask flow_class.py:896-898, 904, 921-934
  apply_flow 't'       is_zero_flow (threshold_vectors: clone + 2 compares)  utils.py:497, 623-643, 919-938
                       meshgrid / stack / permute / normalise_coords /
                       F.grid_sample(bilinear, zeros, align_corners=True)    utils.py:541-555, 445-466
  combine_with mode 3  is_zero x2 (masked), flow.apply(self) + flow          flow_class.py:1729-1744, 1808, 450-488
"""
import torch
import torch.nn.functional as F

THRESHOLD = 1e-3


def valid_vecs(v: torch.Tensor) -> torch.Tensor:
    v = v.float()
    if not torch.isfinite(v).all():                                    # utils.py:98
        raise ValueError("Input contains NaN, Inf or -Inf values")
    return v


def valid_mask(m: torch.Tensor) -> torch.Tensor:
    if ((m != 0) & (m != 1)).any():                                    # utils.py:174
        raise ValueError("Values must be 0 or 1")
    return m.to(torch.bool)


def threshold_vectors(vecs: torch.Tensor) -> torch.Tensor:            # utils.py:623-643
    f = vecs.clone()
    f[(vecs < THRESHOLD) & (vecs > -THRESHOLD)] = 0
    return f


def is_zero_flow(flow: torch.Tensor, thresholded: bool = True) -> torch.Tensor:   # utils.py:919-938
    f = threshold_vectors(flow) if thresholded else flow
    return torch.sum(f == 0, (1, 2, 3)) == f[0].numel()


def flow_is_zero(vecs, mask, thresholded=True) -> torch.Tensor:       # Flow.is_zero, flow_class.py:1226-1244 (masked)
    f = vecs.clone()
    f[~mask.unsqueeze(1).expand(-1, 2, -1, -1)] = 0
    return is_zero_flow(f, thresholded)


def normalise_coords(coords: torch.Tensor, shape) -> torch.Tensor:    # utils.py:445-466
    n = coords * 2
    n[..., 0] /= (shape[1] - 1)
    n[..., 1] /= (shape[0] - 1)
    n -= 1
    return n


def apply_flow_t(flow: torch.Tensor, target: torch.Tensor) -> torch.Tensor:      # utils.py:469-555, 't' branch
    if bool(torch.all(is_zero_flow(flow, thresholded=True))):                       # :497
        return target
    h, w = flow.shape[-2:]
    gy, gx = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing='ij')
    grid = torch.stack((gx, gy), dim=-1).to(torch.float).unsqueeze(0)
    field = normalise_coords(grid - flow.permute(0, 2, 3, 1), (h, w))
    if target.shape[0] != field.shape[0]:
        target = target.expand(field.shape[0], -1, -1, -1)
    return F.grid_sample(target, field, align_corners=True)


def flow_apply_t(f, m, target, target_mask):
    """Flow(f, 't', m).apply(target, target_mask, return_valid_area=True) -> (warped, valid)"""
    f, m = valid_vecs(f), valid_mask(m)
    t = torch.cat((target.float(), target_mask.unsqueeze(1).float()), dim=1)         # flow_class.py:896-898
    warped = apply_flow_t(f, t)                                                      # :904
    valid = warped[:, -1] > 0.99999                                                  # :922
    valid = valid & m                                                                # :934
    return warped[:, :-1], valid


def combine_mode3_t(f1, m1, f2, m2):
    """Flow(f1, 't', m1).combine_with(Flow(f2, 't', m2), 3) -> (vecs, mask)   (flow_class.py:1729-1744, 1808)"""
    f1, m1, f2, m2 = valid_vecs(f1), valid_mask(m1), valid_vecs(f2), valid_mask(m2)
    if bool(torch.all(flow_is_zero(f1, m1, False))):
        return f2, m2
    if bool(torch.all(flow_is_zero(f2, m2, False))):
        return f1, m1
    g, gm = flow_apply_t(f2, m2, f1, m1)                                             # flow.apply(self)
    return f2 + g, m2 & gm                                                           # flow + ...  (:450-488)


def kernel_only(f, src):
    """The bare ATen kernel of the path, for scale: F.grid_sample on a ready-made grid (no validation, no plumbing)."""
    return F.grid_sample(src, f, align_corners=True)
