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


def test_configs1_at_its_own_batch_bit_exact(frames, dev):
    """BASELINE.json configs[1] is B = ONE 1080p frame (only B = 2 was oracle-checked).  4 080 tiles take the
    one-tile-per-block instantiation of the row-table kernel, B = 2 the two-tiles-per-block one: Flow.apply 't' (C = 3, both masks,
    valid area) and mode 3 at B = 1, each frame on its own, against the oracle -- and the library must report a row-table kernel."""
    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native
    from oracle import oracle
    t, n = frames
    for i in (0, 1):
        sl = slice(i, i + 1)
        warped, valid = ofl.Flow(t["f2"][sl], 't', t["m2"][sl]).apply(t["img"][sl], target_mask=t["tm"][sl], return_valid_area=True)
        assert "warp_bwd_rows_kernel<1," in _native.last_kernel_name()
        exp, expv = oracle.flow_apply(n["f2"][sl], 't', n["m2"][sl], n["img"][sl], n["tm"][sl])
        assert np.array_equal(warped.cpu().numpy(), exp) and np.array_equal(valid.cpu().numpy(), expv)
        out = ofl.Flow(t["f1"][sl], 't', t["m1"][sl]).combine_with(ofl.Flow(t["f2"][sl], 't', t["m2"][sl]), 3)
        assert "warp_bwd_rows_kernel<1," in _native.last_kernel_name()
        exp, expm, _ = oracle.combine_with(n["f1"][sl], n["m1"][sl], n["f2"][sl], n["m2"][sl], 3, 't')
        assert np.array_equal(out.vecs.cpu().numpy(), exp) and np.array_equal(out.mask.cpu().numpy(), expm)


def test_apply_s_full_frame(frames, dev):
    """Smooth flow: the in-order (gather) splat is bit-exact at full size.  Bench flow (folds): masks bit-exact, values within the
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
    tw, th, _ = _native.splat_tile_geometry()
    tiles = (2 * ((H + th - 1) // th) * ((W + tw - 1) // tw))
    differing = np.argwhere((got != exp).any(axis=1))                    # (n, y, x) of pixels that are not bit-identical
    touched = {(int(a), int(y) // th, int(x) // tw) for a, y, x in differing}
    assert len(touched) <= st[1], "only tiles that left the exact path may differ (%d differ, %d fell back of %d)" % (len(touched), st[1], tiles)


def test_config3_on_exactly_its_bench_input(dev):
    """BASELINE configs[2] pinned on the input `bench.py` times: B = 16 1080 x 1920 'flow 's' apply, the sigma = 8 flow of
    seed 1000 + 3, image and masks of `bench.make_inputs(16, ..., seed=3)` -- every frame against the oracle: masks bit-exact, values
    within the stated tolerance, and ONLY tiles that left the exact path (the call's fold-tile count) may differ at all."""
    import bench
    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native
    from oracle import oracle
    f1, _, img, m1, _, tm = bench.make_inputs(16, H, W, dev, seed=3)
    _native.collect_splat_stats = True
    try:
        warped, valid = ofl.Flow(f1, 's', m1).apply(img, target_mask=tm, return_valid_area=True)
        st = _native._last_splat_stats.cpu().tolist()
    finally:
        _native.collect_splat_stats = False
    assert st[0] == 0 and st[2] == 0                    # no image on the two-pass path
    tw, th, _ = _native.splat_tile_geometry()
    touched = set()
    for lo in range(0, 16, 4):                          # (the oracle in chunks: it is the checker, its memory is bounded)
        sl = slice(lo, lo + 4)
        exp, expv = oracle.flow_apply(f1[sl].cpu().numpy(), 's', m1[sl].cpu().numpy(), img[sl].cpu().numpy(), tm[sl].cpu().numpy())
        got = warped[sl].cpu().numpy()
        assert np.array_equal(valid[sl].cpu().numpy(), expv)
        np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5 * 255)
        for a, y, x in np.argwhere((got != exp).any(axis=1)):
            touched.add((lo + int(a), int(y) // th, int(x) // tw))
    assert len(touched) <= st[1], "only tiles that left the exact path may differ (%d differ, %d fell back)" % (len(touched), st[1])


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
    the fused composition and the forward splat (several passes of the gather path)."""
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
    as exact fp32 numbers; both steps are bit-exact while the splat stays on its in-order path."""
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


def test_batch_of_64_switch_ref_and_mode1(frames, dev):
    """B = 64 at 1080p through the forward-splat family: `switch_ref` (both references) and `combine_with` mode 1 (both
    references) on 32 copies of two frames repeat the B = 2 results bit for bit -- several passes of the splat, every XCD."""
    import oflibpytorch_amd as ofl
    t, _ = frames
    rep = lambda x: x.repeat((32,) + (1,) * (x.dim() - 1))
    fs, f2, m1, m2 = (rep(t[k]) for k in ("fs", "f2", "m1", "m2"))
    for ref in 'st':
        big = ofl.Flow(fs, ref, m1).switch_ref()
        small = ofl.Flow(t["fs"], ref, t["m1"]).switch_ref()
        assert torch.equal(big.vecs, rep(small.vecs)) and torch.equal(big.mask, rep(small.mask)), ref
        big = ofl.Flow(fs, ref, m1).combine_with(ofl.Flow(f2, ref, m2), 1)
        small = ofl.Flow(t["fs"], ref, t["m1"]).combine_with(ofl.Flow(t["f2"], ref, t["m2"]), 1)
        assert torch.equal(big.vecs, rep(small.vecs)) and torch.equal(big.mask, rep(small.mask)), ref


def test_config5_at_its_per_gpu_batch(dev):
    """BASELINE.json configs[4] at the batch one GPU of eight holds: B = 16, 2160 x 3840, flows stored in fp16;
    `switch_ref` 's' -> 't', then `combine_with` mode 1 in 't' -- 8 copies of two frames must repeat the B = 2 results bit
    for bit (the B = 2 results themselves are pinned against the oracle by the B = 1 test above)."""
    import bench
    import oflibpytorch_amd as ofl
    h, w = 2160, 3840
    f1 = bench.smooth_flow(2, h, w, 2.0, 7000, dev).half()
    f2 = bench.smooth_flow(2, h, w, 8.0, 5000, dev).half()
    m1 = torch.ones(2, h, w, dtype=torch.bool, device=dev)
    m1[:, 300:500, 1000:2500] = False
    m2 = torch.ones(2, h, w, dtype=torch.bool, device=dev)
    m2[1, 1500:1550] = False
    rep = lambda x: x.repeat((8,) + (1,) * (x.dim() - 1))
    small_a = ofl.Flow(f1, 's', m1).switch_ref()
    small_b = small_a.combine_with(ofl.Flow(f2, 't', m2), 1)
    big_a = ofl.Flow(rep(f1), 's', rep(m1)).switch_ref()
    assert big_a.vecs.shape[0] == 16 and big_a.ref == 't'
    assert torch.equal(big_a.vecs, rep(small_a.vecs)) and torch.equal(big_a.mask, rep(small_a.mask))
    big_b = big_a.combine_with(ofl.Flow(rep(f2), 't', rep(m2)), 1)
    assert torch.equal(big_b.vecs, rep(small_b.vecs)) and torch.equal(big_b.mask, rep(small_b.mask))


def test_config5_unsharded_batch_of_128(dev):
    """BASELINE.json configs[4] NOT sharded: B = 128, 2160 x 3840, flows stored in fp16, on ONE GPU (2.1 * 10^9 flow values per
    operand: element offsets pass 2^31, the splat runs in several passes).  64 copies of two frames must repeat the B = 2
    results bit for bit through `switch_ref` 's' -> 't' and `combine_with` mode 1."""
    import bench
    import oflibpytorch_amd as ofl
    h, w = 2160, 3840
    f1 = bench.smooth_flow(2, h, w, 2.0, 7000, dev).half()
    f2 = bench.smooth_flow(2, h, w, 8.0, 5000, dev).half()
    m1 = torch.ones(2, h, w, dtype=torch.bool, device=dev)
    m1[:, 300:500, 1000:2500] = False
    m2 = torch.ones(2, h, w, dtype=torch.bool, device=dev)
    m2[1, 1500:1550] = False
    rep = lambda x: x.repeat((64,) + (1,) * (x.dim() - 1))
    small_a = ofl.Flow(f1, 's', m1).switch_ref()
    small_b = small_a.combine_with(ofl.Flow(f2, 't', m2), 1)
    big_a = ofl.Flow(rep(f1), 's', rep(m1)).switch_ref()
    assert big_a.vecs.shape[0] == 128 and big_a.ref == 't'

    def same(big, small):                     # (compared in slices: no second 8 GB tensor)
        return all(torch.equal(big[i:i + 2], small) for i in range(0, big.shape[0], 2))
    assert same(big_a.vecs, small_a.vecs) and same(big_a.mask, small_a.mask)
    big_b = big_a.combine_with(ofl.Flow(rep(f2), 't', rep(m2)), 1)
    assert same(big_b.vecs, small_b.vecs) and same(big_b.mask, small_b.mask)
    del big_a, big_b
    torch.cuda.empty_cache()


@pytest.mark.parametrize("size", [(300, 400), (1080, 1920)])
def test_uint8_warp_against_the_oracle(size, dev):
    """ofl_warp_bwd_u8 against the ORACLE (not against the float kernel): round-half-even + clamp of oracle.G on the
    converted image (flow_class.py:943-951, utils.py:613-618), and the valid mask of the same call, at 300 x 400
    (BASELINE config 1's frame) and 1080p."""
    import bench
    import oflibpytorch_amd as ofl
    from oracle import oracle
    h, w = size
    g = torch.Generator().manual_seed(17)
    img = torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8)
    f = bench.smooth_flow(2, h, w, 8.0, 1234, torch.device('cpu'))
    m = bench.hole_mask(2, h, w, torch.device('cpu'))
    tm = bench.hole_mask(2, h, w, torch.device('cpu')).flip(1)
    out, valid = ofl.Flow(f.to(dev), 't', m.to(dev)).apply(img.to(dev), target_mask=tm.to(dev), return_valid_area=True)
    # (PURE_PYTORCH, the mode this package implements, keeps the rounded values as floats: flow_class.py:943-949)
    assert out.dtype == torch.float32
    exp, expv = oracle.flow_apply(f.numpy(), 't', m.numpy(), img.float().numpy(), tm.numpy())
    exp8 = np.clip(np.rint(exp), 0, 255).astype(np.uint8)
    assert np.array_equal(out.cpu().numpy(), exp8.astype(np.float32))
    assert np.array_equal(valid.cpu().numpy(), expv)
    plain = ofl.apply_flow(f.to(dev), img.to(dev), 't')                   # apply_flow's own uint8 rule
    assert plain.dtype == torch.uint8 and np.array_equal(plain.cpu().numpy(), exp8)


def test_frames_beyond_2_pow_24_pixels_take_the_backward_warp(dev):
    """The 2^24-pixel limit is the forward splat's (fp32 position index, utils.py:1118); the reference's grid_sample path
    has none.  A 4096 x 4352 frame is warped ('t') by the generic kernel: equal to the oracle on a strip."""
    import oflibpytorch_amd as ofl
    from oracle import oracle
    h, w = 4096, 4352
    g = torch.Generator().manual_seed(3)
    lo = torch.randn(1, 2, 8, 9, generator=g) * 5
    f = torch.nn.functional.interpolate(lo, size=(h, w), mode='bicubic', align_corners=True).contiguous()
    img = torch.rand(1, 1, h, w, generator=g)
    fl = ofl.Flow(f.to(dev), 't')
    out, valid = fl.apply(img.to(dev), return_valid_area=True)
    exp, expv = oracle.flow_apply(f.numpy(), 't', np.ones((1, h, w), bool), img.numpy())
    assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(valid.cpu().numpy(), expv)
    assert bool(fl.is_zero()[0]) is False
    with pytest.raises((RuntimeError, ValueError)):
        ofl.Flow(f.to(dev), 's').apply(img.to(dev))                       # the splat keeps the reference's limit
