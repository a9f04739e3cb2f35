"""GPU tier, FULL SIZE (1080 x 1920, BASELINE.json configs 1-3): the public API on the HIP kernels against the CPU oracle
at the frame size the metric is quoted on (the C oracle does B = 2 frames in a fraction of a second), and -- at the full
batch of the bench, B = 64 -- through a size-independent property: batch elements are independent, so a batch that repeats
two frames must repeat their results bit for bit, whatever tile / XCD / pass the copies fall into.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W = 1080, 1920


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu tier needs a HIP device"
    from oflibpytorch_amd import _native
    _native.load_library()
    return torch.device('cuda', 0)


@pytest.fixture(scope="module")
def frames(dev):
    """Two bench frames (bench.py's generators: the SURVEY.md 8(d) synthetic inputs), on the GPU and as numpy."""
    import bench
    f1 = bench.smooth_flow(2, H, W, 8.0, 1000, dev)
    f2 = bench.smooth_flow(2, H, W, 8.0, 5000, dev)
    fs = bench.smooth_flow(2, H, W, 2.0, 7000, dev)          # a smoother one: no folds, the splat stays on its exact path
    _, _, img, m1, m2, tm = bench.make_inputs(2, H, W, dev, 0)
    t = dict(f1=f1, f2=f2, fs=fs, img=img, m1=m1, m2=m2, tm=tm)
    n = {k: v.cpu().numpy() for k, v in t.items()}
    return t, n


def test_apply_t_full_frame_bit_exact(frames, dev):
    import oflibpytorch_amd as ofl
    from oracle import oracle
    t, n = frames
    warped, valid = ofl.Flow(t["f2"], 't', t["m2"]).apply(t["img"], target_mask=t["tm"], return_valid_area=True)
    exp, expv = oracle.flow_apply(n["f2"], 't', n["m2"], n["img"], n["tm"])
    assert np.array_equal(warped.cpu().numpy(), exp)
    assert np.array_equal(valid.cpu().numpy(), expv)


def test_combine_mode3_full_frame_bit_exact(frames, dev):
    import oflibpytorch_amd as ofl
    from oracle import oracle
    t, n = frames
    for ref in 'ts':
        out = ofl.Flow(t["f1"], ref, t["m1"]).combine_with(ofl.Flow(t["f2"], ref, t["m2"]), 3)
        exp, expm, _ = oracle.combine_with(n["f1"], n["m1"], n["f2"], n["m2"], 3, ref)
        assert np.array_equal(out.vecs.cpu().numpy(), exp), ref
        assert np.array_equal(out.mask.cpu().numpy(), expm), ref


def test_apply_s_full_frame(frames, dev):
    """Smooth flow: the routed splat is bit-exact at full size.  Bench flow (folds): masks bit-exact, values within the
    stated tolerance on the tiles that fell back to float atomics -- and bit-exact everywhere else."""
    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native
    from oracle import oracle
    _native.collect_splat_stats = True
    t, n = frames
    warped, valid = ofl.Flow(t["fs"], 's', t["m1"]).apply(t["img"], target_mask=t["tm"], return_valid_area=True)
    assert _native._last_splat_stats.cpu().tolist()[:2] == [0, 0]
    exp, expv = oracle.flow_apply(n["fs"], 's', n["m1"], n["img"], n["tm"])
    assert np.array_equal(warped.cpu().numpy(), exp)
    assert np.array_equal(valid.cpu().numpy(), expv)

    warped, valid = ofl.Flow(t["f1"], 's', t["m1"]).apply(t["img"], target_mask=t["tm"], return_valid_area=True)
    st = _native._last_splat_stats.cpu().tolist()
    assert st[0] == 0                                   # no launch-level fallback on the bench workload
    exp, expv = oracle.flow_apply(n["f1"], 's', n["m1"], n["img"], n["tm"])
    got = warped.cpu().numpy()
    assert np.array_equal(valid.cpu().numpy(), expv)
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5 * 255)
    tiles = (2 * ((H + 15) // 16) * ((W + 31) // 32))
    differing = np.argwhere((got != exp).any(axis=1))                    # (n, y, x) of pixels that are not bit-identical
    touched = {(int(a), int(y) // 16, int(x) // 32) for a, y, x in differing}
    assert len(touched) <= st[1], "only tiles that left the exact path may differ (%d differ, %d fell back of %d)" % (len(touched), st[1], tiles)


def test_switch_ref_full_frame(frames, dev):
    import oflibpytorch_amd as ofl
    from oracle import oracle
    t, n = frames
    for ref in 'st':
        out = ofl.Flow(t["fs"], ref, t["m1"]).switch_ref()
        exp, expm, _ = oracle.switch_ref(n["fs"], ref, n["m1"])
        assert np.array_equal(out.vecs.cpu().numpy(), exp), ref
        assert np.array_equal(out.mask.cpu().numpy(), expm), ref


def test_batch_of_64_repeats_its_two_frames(frames, dev):
    """B = 64 (the bench batch): 32 copies of two frames -> 32 copies of their results, bit for bit, for the backward warp,
    the fused composition and the forward splat (several passes of the routed path)."""
    import oflibpytorch_amd as ofl
    t, _ = frames
    rep = lambda x: x.repeat((32,) + (1,) * (x.dim() - 1))
    f1, f2, fs, img, m1, m2, tm = (rep(t[k]) for k in ("f1", "f2", "fs", "img", "m1", "m2", "tm"))
    big = ofl.Flow(f2, 't', m2).apply(img, target_mask=tm, return_valid_area=True)
    small = ofl.Flow(t["f2"], 't', t["m2"]).apply(t["img"], target_mask=t["tm"], return_valid_area=True)
    for b, s in zip(big, small):
        assert torch.equal(b, rep(s))
    big = ofl.Flow(f1, 't', m1).combine_with(ofl.Flow(f2, 't', m2), 3)
    small = ofl.Flow(t["f1"], 't', t["m1"]).combine_with(ofl.Flow(t["f2"], 't', t["m2"]), 3)
    assert torch.equal(big.vecs, rep(small.vecs)) and torch.equal(big.mask, rep(small.mask))
    big = ofl.Flow(fs, 's', m1).apply(img, target_mask=tm, return_valid_area=True)
    small = ofl.Flow(t["fs"], 's', t["m1"]).apply(t["img"], target_mask=t["tm"], return_valid_area=True)
    for b, s in zip(big, small):
        assert torch.equal(b, rep(s))


def test_config5_4k_fp16_switch_ref_then_mode1(dev):
    """BASELINE.json configs[4]: 2160 x 3840 flows stored in fp16, ``switch_ref`` 's' -> 't', then ``combine_with`` mode 1
    in 't'.  The reference turns every flow into fp32 on entry (utils.py:95,118), so the oracle gets the same fp16 values
    as exact fp32 numbers; both steps are bit-exact while the splat stays on its routed exact path."""
    import bench
    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native
    from oracle import oracle
    _native.collect_splat_stats = True
    h, w = 2160, 3840
    f1 = bench.smooth_flow(1, h, w, 2.0, 7000, dev).half()
    f2 = bench.smooth_flow(1, h, w, 8.0, 5000, dev).half()
    m1 = torch.ones(1, h, w, dtype=torch.bool, device=dev)
    m1[:, 300:500, 1000:2500] = False
    m2 = torch.ones(1, h, w, dtype=torch.bool, device=dev)
    m2[:, 1500:1550] = False
    a = ofl.Flow(f1, 's', m1).switch_ref()
    assert a.ref == 't' and a.vecs.dtype == torch.float32
    exact = _native._last_splat_stats.cpu().tolist()[:2] == [0, 0]
    n1, n2 = f1.float().cpu().numpy(), f2.float().cpu().numpy()
    ev, em, _ = oracle.switch_ref(n1, 's', m1.cpu().numpy())
    assert np.array_equal(a.mask.cpu().numpy(), em)
    if exact:
        assert np.array_equal(a.vecs.cpu().numpy(), ev)
    else:
        np.testing.assert_allclose(a.vecs.cpu().numpy(), ev, rtol=2e-5, atol=2e-5 * 30)
    # mode 1 on the oracle's own intermediate, so that the second step is pinned by itself
    b = ofl.Flow(torch.from_numpy(ev).to(dev), 't', torch.from_numpy(em).to(dev)).combine_with(ofl.Flow(f2, 't', m2), 1)
    st = _native._last_splat_stats.cpu().tolist()[:2]
    cv, cm, _ = oracle.combine_with(ev, em, n2, m2.cpu().numpy(), 1, 't')
    assert np.array_equal(b.mask.cpu().numpy(), cm)
    if st == [0, 0]:
        assert np.array_equal(b.vecs.cpu().numpy(), cv)
    else:
        np.testing.assert_allclose(b.vecs.cpu().numpy(), cv, rtol=2e-5, atol=2e-5 * 60)
