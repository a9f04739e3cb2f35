"""CPU tier: the oracle's CLOSED FORMS (`oracle.flow_apply`, `combine_with`, `switch_ref`, `invert`,
`flow_is_zero` -- oracle/oracle.py) against the fixtures the imported reference produced, bit for bit.

`tests/test_host_logic_golden.py` pins the oracle's primitives (G, P, theta, flags) through the host mirror;
this file pins the compositions that `tests/test_gpu_fullsize.py`, `__graft_entry__.smoke()`, the gloo test
and `bench.py`'s cpu_baseline leg compare against -- with nothing of the package in between.
Reference: flow_class.py:755-959 (apply), :1022-1086 (switch_ref / invert), :1226-1244 (is_zero),
:1648-1810 (combine_with); the reference's own KAT values of BASELINE.md config 1.
"""
import numpy as np
import pytest

from conftest import golden_ids
import codec
from oracle import oracle


def _exact(got, exp, what):
    got, exp = np.asarray(got), np.asarray(exp)
    assert got.shape == exp.shape, "%s: shape %s != %s" % (what, got.shape, exp.shape)
    if exp.dtype == np.bool_:
        assert np.array_equal(got.astype(bool), exp), "%s: %d mask bits differ" % (what, np.count_nonzero(got != exp))
    else:
        bad = ~((got == exp) | (np.isnan(got) & np.isnan(exp)))
        assert not bad.any(), "%s: %d of %d values differ" % (what, int(bad.sum()), bad.size)


def _mask(i, key, like):
    return np.ones((like.shape[0],) + like.shape[2:], bool) if key not in i else np.asarray(i[key], bool)


def _apply_cases():
    out = []
    for cid in golden_ids('Flow.apply'):
        out.append(cid)
    return out


@pytest.mark.parametrize("cid", _apply_cases())
def test_oracle_flow_apply(cid, golden):
    case = golden.cases[cid]
    a = case["args"]
    kw = a["kwargs"]
    if kw.get("padding") is not None:
        pytest.skip("padding is host plumbing around the closed form (pinned through test_host_logic_golden)")
    i, exp = golden.arrays(case)
    f = i["f"].astype(np.float32)
    f = f[None] if f.ndim == 3 else f
    m = _mask(i, "m", f)
    if "tf" in i:
        v, vm = oracle.flow_apply_to_flow(f, a["ref"], m, i["tf"], i["tm"])
        _exact(v, exp["vecs"], "vecs")
        _exact(vm, exp["mask"], "mask")
        return
    tgt = i["target"]
    dt = tgt.dtype
    t = tgt.astype(np.float32)
    t = t[None, None] if t.ndim == 2 else (t[None] if t.ndim == 3 else t)
    tm = i.get("target_mask")
    if tm is not None and tm.ndim == 2:
        tm = tm[None]
    w, valid = oracle.flow_apply(f, a["ref"], m, t, tm, kw.get("consider_mask", True) is not False)
    if dt.kind in 'ui':                                            # flow_class.py:943-951, utils.py:613-618
        w = np.rint(w)
        if dt == np.uint8:
            w = np.clip(w, 0, 255)
        w = w.astype(dt)
    e = exp["warped"]
    if w.shape[0] == 1 and e.ndim < 4:
        w = w[0, 0] if e.ndim == 2 else w[0]
    _exact(w, e, "warped")
    if "valid" in exp:
        _exact(valid, exp["valid"], "valid")


@pytest.mark.parametrize("cid", golden_ids('Flow.switch_ref'))
def test_oracle_switch_ref(cid, golden):
    case = golden.cases[cid]
    i, exp = golden.arrays(case)
    if case["args"].get("mode") == 'invalid':
        pytest.skip("relabelling only")
    v, vm, r = oracle.switch_ref(i["f"], case["args"]["ref"], _mask(i, "m", i["f"]))
    _exact(v, exp["vecs"], "vecs")
    _exact(vm, exp["mask"], "mask")
    assert r == case["args"]["out_ref"]


@pytest.mark.parametrize("cid", golden_ids('Flow.invert'))
def test_oracle_invert(cid, golden):
    case = golden.cases[cid]
    i, exp = golden.arrays(case)
    v, vm, r = oracle.invert(i["f"], case["args"]["ref"], _mask(i, "m", i["f"]), case["args"]["arg_ref"])
    _exact(v, exp["vecs"], "vecs")
    _exact(vm, exp["mask"], "mask")
    assert r == case["args"]["out_ref"]


@pytest.mark.parametrize("cid", golden_ids('Flow.is_zero'))
def test_oracle_flow_is_zero(cid, golden):
    case = golden.cases[cid]
    a = case["args"]
    i, exp = golden.arrays(case)
    th = True if a["thresholded"] is None else a["thresholded"]
    mk = True if a["masked"] is None else a["masked"]
    _exact(oracle.flow_is_zero(i["f"], _mask(i, "m", i["f"]), th, mk), exp["out"], "is_zero")


@pytest.mark.parametrize("cid", golden_ids('Flow.combine_with'))
def test_oracle_combine_with(cid, golden):
    case = golden.cases[cid]
    a = case["args"]
    i, exp = golden.arrays(case)
    v, vm, r = oracle.combine_with(i["f1"], _mask(i, "m1", i["f1"]), i["f2"], _mask(i, "m2", i["f2"]), a["mode"], a["ref"],
                                   bool(a.get("thresholded")))
    _exact(v, exp["vecs"], "vecs")
    _exact(vm, exp["mask"], "mask")
    assert r == a["out_ref"]


@pytest.mark.parametrize("cid", golden_ids('combine_flows'))
def test_oracle_combine_flows(cid, golden):
    case = golden.cases[cid]
    a = case["args"]
    i, exp = golden.arrays(case)

    def vec(x):
        x = np.asarray(x, np.float32)
        x = x[None] if x.ndim == 3 else x
        return np.moveaxis(x, -1, 1) if x.shape[1] != 2 else x
    f1, f2 = vec(i["f1"]), vec(i["f2"])
    ones = np.ones((f1.shape[0],) + f1.shape[2:], bool)
    v, _, _ = oracle.combine_with(f1, ones, f2, ones, a["mode"], 't' if a["ref"] is None else a["ref"])
    _exact(v[0] if exp["vecs"].ndim == 3 else v, exp["vecs"], "vecs")


@pytest.mark.parametrize("cid", golden_ids('switch_flow_ref') + golden_ids('invert_flow'))
def test_oracle_tensor_wrappers(cid, golden):
    case = golden.cases[cid]
    a = case["args"]
    i, exp = golden.arrays(case)
    f = np.asarray(i["f"], np.float32)
    f4 = f[None] if f.ndim == 3 else f
    if f4.shape[1] != 2:
        f4 = np.moveaxis(f4, -1, 1)
    ones = np.ones((f4.shape[0],) + f4.shape[2:], bool)
    if case["op"] == 'switch_flow_ref':
        v = oracle.switch_ref(f4, a["ref"], ones)[0]
    else:
        v = oracle.invert(f4, a["ref"], ones, a["out_ref"])[0]
    _exact(v[0] if exp["vecs"].ndim == 3 else v, exp["vecs"], "vecs")


# the reference's own known answers for BASELINE config 1 (300 x 400 crop of smudge.png, from_transforms flows):
# mask counts 99 591 / 100 274 / 58 800 ..., through the closed forms alone
@pytest.mark.parametrize("cid", golden_ids('kat_cfg1'))
def test_oracle_config1_known_answers(cid, golden):
    case = golden.cases[cid]
    a = case["args"]
    inp = golden.cases["kats.cfg1_inputs"]
    i, _ = golden.arrays(inp)
    _, exp = golden.arrays(case)

    def field(name):
        v = codec.decode_affine(i[name + "__params"], i[name + "__delta"], i[name + "__esc"])
        return v[None].astype(np.float32)
    img = i["img_u8"].astype(np.float32)
    img = img[None] if img.ndim == 3 else img
    if a["call"] == 'apply':
        f = field(a["flow"])
        vec, mask = oracle.flow_apply(f, a["flow"][-1], np.ones(f.shape[:1] + f.shape[2:], bool), img)
    elif a["call"] == 'switch_ref':
        f = field(a["flow"])
        vec, mask, _ = oracle.switch_ref(f, a["flow"][-1], np.ones(f.shape[:1] + f.shape[2:], bool))
    else:
        f1, f2 = field(a["self"]), field(a["flow"])
        ones = np.ones(f1.shape[:1] + f1.shape[2:], bool)
        vec, mask, _ = oracle.combine_with(f1, ones, f2, ones, a["mode"], a["self"][-1])
    m = np.unpackbits(exp["mask_packed"])[:mask.size].reshape(mask.shape).astype(bool)
    assert np.array_equal(mask, m)
    assert int(mask.sum()) == a["mask_count"]
    sub = codec.subsample(vec, 4)
    _exact(sub[0] if exp["sub4"].ndim == sub.ndim - 1 else sub, exp["sub4"], "vec")


# ------------------------------------------------------------------------------------------------
# oracle/torch_ops.py: the reference's torch-CPU op sequence that bench.py times as the CPU baseline
# ------------------------------------------------------------------------------------------------
def _torch_apply_cases():
    out = []
    for cid in golden_ids('Flow.apply'):
        from conftest import _GOLDEN
        c = _GOLDEN.cases[cid]
        if c["args"]["ref"] == 't' and c["args"]["kwargs"].get("padding") is None and "target_mask" in c["in"] \
                and c["args"]["kwargs"].get("return_valid_area") and "tf" not in c["in"]:
            out.append(cid)
    return out


@pytest.mark.parametrize("cid", _torch_apply_cases())
def test_torch_op_sequence_apply_t(cid, golden):
    import torch
    from oracle import torch_ops
    case = golden.cases[cid]
    i, exp = golden.arrays(case)
    t = torch.tensor(i["target"]).float()
    if t.dim() != 4 or i["f"].ndim != 4 or i["target_mask"].ndim != 3:
        pytest.skip("rank plumbing is not part of the timed sequence")
    f, m, tm = torch.tensor(i["f"]), torch.tensor(i["m"]), torch.tensor(i["target_mask"])
    if t.shape[0] != f.shape[0] or tm.shape[0] != f.shape[0] or i["target"].dtype.kind != 'f':
        pytest.skip("broadcast / integer targets are not part of the timed sequence")
    w, v = torch_ops.flow_apply_t(f, m, t, tm)
    _exact(w.numpy(), exp["warped"], "warped")
    _exact(v.numpy(), exp["valid"], "valid")


@pytest.mark.parametrize("cid", [c for c in golden_ids('Flow.combine_with')])
def test_torch_op_sequence_combine_mode3_t(cid, golden):
    import torch
    from oracle import torch_ops
    case = golden.cases[cid]
    a = case["args"]
    if a["mode"] != 3 or a["ref"] != 't' or a.get("thresholded"):
        pytest.skip("the timed sequence is mode 3, 't'")
    i, exp = golden.arrays(case)
    v, m = torch_ops.combine_mode3_t(torch.tensor(i["f1"]), torch.tensor(i["m1"]), torch.tensor(i["f2"]), torch.tensor(i["m2"]))
    _exact(v.numpy(), exp["vecs"], "vecs")
    _exact(m.numpy(), exp["mask"], "mask")


# ----------------------------------------------------------------------------------------------------------------------
# round 4: ATen's CPU bilinear resize restated (orc_resize_bilinear_f32) -- pinned on the reference's own resize fixtures and on torch
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cid", ["gen.resize_flow_half", "gen.resize_flow_x2", "gen.resize_flow_aniso", "gen.resize_flow_tuple",
                                 "gen.Flow_resize_half_s", "gen.Flow_resize_aniso_s", "gen.Flow_resize_half_t", "gen.Flow_resize_aniso_t"])
def test_oracle_resize_against_the_reference_fixtures(cid, golden):
    case = golden.cases[cid]
    i, o = golden.arrays(case)
    sc = case["args"]["scale"]
    sc = [sc, sc] if isinstance(sc, (int, float)) else list(sc)
    flow = i["flow"] if "flow" in i else i["f"]
    got = oracle.resize_flow(flow, sc)
    exp = o["out"] if "out" in o else o["vecs"]
    assert got.shape == exp.shape and np.array_equal(got, exp), cid
    if "mask" in o:                                           # flow_class.py:710-713: the mask interpolated as floats, then rounded
        m = oracle.resize_bilinear(i["m"].astype(np.float32)[:, None], sc)[:, 0]
        assert np.array_equal(np.rint(m).astype(bool), o["mask"].astype(bool)), cid


def test_oracle_resize_equals_atens_cpu_kernels():
    """Both of ATen's CPU kernels (it picks by output size: oh + ow <= 128), the copied dimension, up- and down-sampling."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(12)
    for shape in [(1, 2, 12, 16), (3, 2, 37, 53), (2, 1, 64, 64), (1, 2, 90, 140), (1, 2, 1, 9), (1, 2, 2, 2)]:
        x = torch.randn(*shape, generator=g) * 7
        for sf in [(0.5, 0.5), (2, 2), (1.5, 1.5), (0.7, 1.3), (1.3, 0.8), (0.3, 2.7), (1, 1), (1.01, 0.99), (0.125, 4.0)]:
            try:
                ref = F.interpolate(x, scale_factor=[float(sf[0]), float(sf[1])], mode='bilinear', align_corners=False).numpy()
            except RuntimeError:
                continue
            got = oracle.resize_bilinear(x.numpy(), sf)
            assert got.shape == ref.shape and np.array_equal(got, ref), (shape, sf)

