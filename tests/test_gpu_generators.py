"""GPU tier, SURVEY.md 8(f) rank 4 -- the callers either side of the path: `from_matrix` / `from_transforms` (utils.py:646-807),
`resize_flow` (:878-916), `Flow.from_matrix` / `from_transforms` / `resize` / `pad` / `unpad` (flow_class.py:238-328, 694-753) on the
HIP device against the fixtures the imported reference produced (tests/golden/gen.npz, group `gen`).

Bars:
  * generated fields (`ofl_flow_from_matrix_f32`), pad, unpad: BIT-EXACT (the kernel restates the accumulation order of ATen's CPU
    batched matmul; pad / unpad move values) -- for a GIVEN matrix; where the matrix itself comes out of host linear algebra
    (`_host_algebra`) the fixture is met within 1e-5 of the field's scale and the kernel is pinned against the oracle on
    matrices formed on this host;
  * resize (round 4): BIT-EXACT -- `ofl_resize_bilinear_f32` restates the arithmetic of ATen's CPU kernels; until round 3 the package
    called `F.interpolate` on the flow's device as the reference does (utils.py:912) -- on a HIP device that is ATen's GPU kernel,
    whose interpolation weights round differently from ATen's CPU kernel in the last bits: |got - expected| <= 1e-5 * max|expected|
    (fp32 tolerance, stated here); the resized MASK (a rounded interpolation of 0 / 1 values) bit for bit.
"""
import numpy as np
import pytest
import torch

from conftest import golden_ids
import case_runner

pytestmark = pytest.mark.gpu

GEN = golden_ids(group='gen')


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu tier needs a HIP device"
    from oflibpytorch_amd import _native
    _native.load_library()
    return torch.device('cuda', 0)


def _host_algebra(case):
    """Does the case's 3 x 3 matrix go through HOST linear algebra before the field is generated (a product of transform
    matrices, utils.py:379-424, or `torch.pinverse`, :702)?  MKL / LAPACK choose their kernels by CPU model, so that matrix --
    nine numbers -- can differ in the last bit between the build container that wrote the fixtures and the GPU box's host; the
    reference itself would differ the same way.  Such cases are held to 1e-5 of the field's scale against the fixture here
    (bit for bit in the CPU tier, on the fixtures' own host), and `test_from_matrix_full_size_against_the_oracle` pins the
    device kernel bit for bit on matrices computed on THIS host."""
    a, op = case["args"], case["op"]
    if op in ('from_transforms', 'Flow.from_transforms'):
        return True
    return op in ('from_matrix', 'Flow.from_matrix') and a["ref"] == 't' and not a["matrix_is_inverse"]


@pytest.mark.parametrize("cid", GEN)
def test_generator_case_gpu(cid, golden, dev):
    case = golden.cases[cid]
    got = case_runner.run_case(case, golden, dev)
    if _host_algebra(case):                    # (round 4: resize is bit-exact -- ofl_resize_bilinear_f32 restates ATen's CPU kernels)
        _, exp = golden.arrays(case)
        scale = max([float(np.abs(v).max()) for v in exp.values() if v.dtype.kind == 'f' and v.size] + [1.0])
        case_runner.check_case(case, golden, got, exact_values=False, rtol=0.0, atol=1e-5 * scale, max_mask_flips=0)
    else:
        case_runner.check_case(case, golden, got, exact_values=True)
    for k, v in got.items():                                   # results live where the reference would put them
        if isinstance(v, torch.Tensor) and k in ("vecs", "mask"):
            assert v.device.type == 'cuda'


def test_from_matrix_full_size_against_the_oracle(dev):
    """1080p and 4K homographies, both references, a batch of matrices: bit for bit the oracle's restatement (pinned by the
    fixtures above) -- and generated on the device the Flow is wanted on."""
    import oflibpytorch_amd as ofl
    from oracle import oracle
    m = ofl.utils.matrix_from_transforms([['rotation', 960, 540, -30], ['scaling', 400, 300, 0.9]])
    p = m.clone()
    p[2, 0], p[2, 1] = 1.5e-5, -2e-5
    for (h, w) in ((1080, 1920), (2160, 3840), (37, 1001)):
        for ref in 'st':
            got = ofl.from_matrix(p.to(dev), (h, w), ref, matrix_is_inverse=True if ref == 't' else None)
            assert got.device.type == 'cuda'
            exp = oracle.flow_from_matrix(p.numpy()[None], 1, h, w, 1.0 if ref == 's' else -1.0)
            assert np.array_equal(got.cpu().numpy(), exp)
    # matrices that went through THIS host's linear algebra: the pseudo-inverse route of 't' and a chain of transforms
    pin = torch.pinverse(p.unsqueeze(0))
    got = ofl.from_matrix(p.to(dev), (270, 480), 't')
    assert np.array_equal(got.cpu().numpy(), oracle.flow_from_matrix(pin.numpy(), 1, 270, 480, -1.0))
    chain = [['translation', -3, 4.5], ['rotation', 200, 100, 33], ['scaling', 150, 120, 1.25]]
    got = ofl.from_transforms([list(t) for t in chain], (270, 480), 's')
    assert np.array_equal(got.numpy(), oracle.flow_from_matrix(ofl.utils.matrix_from_transforms(chain).numpy()[None], 1, 270, 480))
    mats = torch.stack([m, p, torch.eye(3)])
    got = ofl.from_matrix(mats.to(dev), (300, 400), 's')
    assert np.array_equal(got.cpu().numpy(), oracle.flow_from_matrix(mats.numpy(), 3, 300, 400))
    fl = ofl.Flow.from_transforms([['rotation', 200, 150, -30]], (300, 400), 't', device=dev)
    assert fl.vecs.device.type == 'cuda'
    fc = ofl.from_transforms([['rotation', 200, 150, -30]], (300, 400), 't')          # CPU matrix in -> CPU tensor out, same values
    assert fc.device.type == 'cpu' and torch.equal(fc, fl.vecs.cpu())


def test_from_matrix_keeps_the_graph_of_a_matrix_that_wants_a_gradient(dev):
    import oflibpytorch_amd as ofl
    m = torch.eye(3, device=dev).clone().requires_grad_()
    out = ofl.from_matrix(m, (20, 30), 's')
    assert out.grad_fn is not None
    out.sum().backward()
    assert m.grad is not None and bool(torch.isfinite(m.grad).all())


@pytest.mark.parametrize("ref", ['s', 't'])
def test_from_matrix_with_a_host_matrix_that_wants_a_gradient_and_a_hip_device(ref, dev):
    """`Flow.from_matrix(cpu_matrix_requires_grad, shape, device='cuda')` -- the differentiable branch built
    its grid on the wanted device and multiplied it with the matrix where the matrix lived.  The matrix is moved (the graph goes
    along), the field comes out on the wanted device, and the gradient arrives at the HOST leaf."""
    import oflibpytorch_amd as ofl
    base = torch.tensor([[1.02, 0.03, 2.0], [-0.01, 0.97, -1.0], [0.0, 0.0, 1.0]])
    m = base.clone().requires_grad_()
    fl = ofl.Flow.from_matrix(m, (24, 36), ref, device=dev)
    assert fl.vecs.device.type == 'cuda' and fl.vecs.grad_fn is not None
    fl.vecs.sum().backward()
    assert m.grad is not None and m.grad.device.type == 'cpu' and bool(torch.isfinite(m.grad).all())
    # same values as the non-differentiable route (the kernel) to fp32 rounding of the torch expression
    plain = ofl.Flow.from_matrix(base, (24, 36), ref, device=dev)
    assert torch.allclose(fl.vecs.detach(), plain.vecs, rtol=1e-5, atol=1e-4)


def test_flow_from_matrix_broadcasts_n_matrices_over_a_batch_of_one(dev):
    """utils.flow_from_matrix(matrix[N,3,3], [1,H,W]) broadcasts to N fields, as the reference's matmul does
    (utils.py:364-370); the reference's own from_matrix calls it exactly this way."""
    import oflibpytorch_amd as ofl
    from oracle import oracle
    mats = torch.stack([torch.eye(3), torch.tensor([[1.0, 0.1, 3.0], [0.0, 1.0, -2.0], [0.0, 0.0, 1.0]])])
    got = ofl.utils.flow_from_matrix(mats.to(dev), [1, 20, 28])
    assert tuple(got.shape) == (2, 2, 20, 28)
    assert np.array_equal(got.cpu().numpy(), oracle.flow_from_matrix(mats.numpy(), 2, 20, 28))


@pytest.mark.parametrize("shape", [(1, 2, 12, 16), (3, 2, 37, 53), (2, 1, 64, 64), (1, 2, 270, 480), (2, 2, 1080, 1920)])
def test_resize_on_the_device_equals_atens_cpu_kernels_bit_for_bit(shape, dev):
    """ofl_resize_bilinear_f32 == F.interpolate on the HOST (the reference's PyTorch-CPU path) == the oracle: both of ATen's
    kernels (oh + ow <= 128 and above), a copied dimension, up- and down-sampling; resize_flow's component scaling on top."""
    import torch.nn.functional as F
    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native
    from oracle import oracle
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) * 9
    for sf in [(0.5, 0.5), (2, 2), (1.5, 1.5), (0.7, 1.3), (1.3, 0.8), (0.3, 2.7), (1, 1), (1.01, 0.99), (0.125, 0.25)]:
        if shape[-1] >= 1000 and max(sf) > 1.5:
            continue
        ref = F.interpolate(x, scale_factor=[float(sf[0]), float(sf[1])], mode='bilinear', align_corners=False)
        got = _native.resize_bilinear(x.to(dev), sf)
        # bit for bit against ATen's CPU kernels on the torch version they were probed on (kernel selection and FMA
        # contraction are properties of a build); on any other version within 1e-6 of the scale -- and always exactly the oracle
        from oflibpytorch_amd import utils as _u
        if torch.__version__.startswith(_u.RESIZE_PROBED_TORCH):
            assert got.shape == ref.shape and torch.equal(got.cpu(), ref), (shape, sf)
        else:
            assert got.shape == ref.shape and float((got.cpu() - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), (shape, sf)
        assert np.array_equal(got.cpu().numpy(), oracle.resize_bilinear(x.numpy(), sf))
    if shape[1] == 2:
        out = ofl.resize_flow(x.to(dev), [1.5, 0.7])
        assert out.device.type == 'cuda' and np.array_equal(out.cpu().numpy(), oracle.resize_flow(x.numpy(), [1.5, 0.7]))
        if torch.__version__.startswith(_u.RESIZE_PROBED_TORCH):
            assert torch.equal(out.cpu(), ofl.resize_flow(x, [1.5, 0.7]))       # the host route (ATen's CPU kernel itself)
        _u.RESIZE_MATCHES_ATEN_CPU = False                                      # the reference's own route on a HIP device: ATen's GPU kernel
        try:
            aten = ofl.resize_flow(x.to(dev), [1.5, 0.7])
        finally:
            _u.RESIZE_MATCHES_ATEN_CPU = True
        assert aten.device.type == 'cuda' and float((aten.cpu() - out.cpu()).abs().max()) <= 1e-5 * float(out.abs().max())

