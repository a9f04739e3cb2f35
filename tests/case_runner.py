"""Replays a golden case (tests/golden/manifest.json) through the package's public API and compares the
result with what the reference produced.  Used by the CPU host-logic tier (oracle-backed primitives) and
by the GPU parity tier (HIP kernels)."""
import numpy as np
import torch

import codec
import oflibpytorch_amd as ofl
from oflibpytorch_amd import Flow


def T(a, device):
    return None if a is None else torch.tensor(np.asarray(a)).to(device)


def _flow(i, f, m, ref, device):
    return Flow(T(i[f], device), ref, T(i.get(m), device))


def _cfg1_fields(golden, device):
    case = golden.cases["kats.cfg1_inputs"]
    i, _ = golden.arrays(case)
    out = {"img": T(i["img_u8"], device).float()}
    for name in case["args"]["fields"]:
        v = codec.decode_affine(i[name + "__params"], i[name + "__delta"], i[name + "__esc"])
        out[name] = Flow(T(v[None], device), name[-1])
    return out


def run_case(case, golden, device):
    """-> dict name -> torch tensor / python value, keyed like case['out'] (plus '_ref', '_same')."""
    i, _ = golden.arrays(case)
    a, op = case["args"], case["op"]
    d = device
    if op == 'normalise_coords':
        return {"out": ofl.normalise_coords(T(i["coords"], d), a["shape"])}
    if op == 'get_flow_endpoints':
        x, y = ofl.get_flow_endpoints(T(i["flow"], d), a["ref"])
        return {"x": x, "y": y}
    if op == 'threshold_vectors':
        return {"out": ofl.threshold_vectors(T(i["flow"], d))}
    if op == 'is_zero_flow':
        return {"out": ofl.is_zero_flow(T(i["flow"], d), a["thresholded"])}
    if op == 'apply_flow':
        tgt = T(i["target"], d)
        out = ofl.apply_flow(T(i["flow"], d), tgt, a["ref"], T(i.get("mask"), d))
        return {"out": out, "_same": out is tgt}
    if op == 'grid_from_unstructured_data':
        data, den = ofl.grid_from_unstructured_data(T(i["x"], d), T(i["y"], d), T(i["data"], d), T(i.get("mask"), d))
        return {"data": data, "density": den}
    if op == 'apply_s_flow':
        data, m = ofl.apply_s_flow(T(i["flow"], d), T(i["data"], d), T(i.get("mask"), d), a["occlude_zero_flow"])
        return {"data": data, "mask": m}
    if op == 'Flow.apply':
        fl = _flow(i, "f", "m", a["ref"], d)
        kw = dict(a["kwargs"])
        if "tf" in i:
            out = fl.apply(Flow(T(i["tf"], d), a["target_ref"], T(i["tm"], d)), **kw)
            return {"vecs": out.vecs, "mask": out.mask, "_ref": out.ref}
        if "target_mask" in i:
            kw["target_mask"] = T(i["target_mask"], d)
        out = fl.apply(T(i["target"], d), **kw)
        return {"warped": out[0], "valid": out[1]} if isinstance(out, tuple) else {"warped": out}
    if op in ('Flow.switch_ref', 'Flow.invert', 'Flow.is_zero', 'Flow.valid_target', 'Flow.valid_source'):
        fl = _flow(i, "f", "m", a["ref"], d)
        if op == 'Flow.switch_ref':
            out = fl.switch_ref(a.get("mode"))
        elif op == 'Flow.invert':
            out = fl.invert(a["arg_ref"])
        elif op == 'Flow.is_zero':
            return {"out": fl.is_zero(a["thresholded"], a["masked"])}
        elif op == 'Flow.valid_target':
            return {"out": fl.valid_target(a["consider_mask"])}
        else:
            return {"out": fl.valid_source(a["consider_mask"])}
        return {"vecs": out.vecs, "mask": out.mask, "_ref": out.ref}
    if op == 'Flow.combine_with':
        f1, f2 = _flow(i, "f1", "m1", a["ref"], d), _flow(i, "f2", "m2", a["ref"], d)
        out = f1.combine_with(f2, a["mode"], a.get("thresholded"))
        ret = "flow" if out is f2 else ("self" if out is f1 else "new")
        return {"vecs": out.vecs, "mask": out.mask, "_ref": out.ref, "_returns": ret}
    if op == 'combine_flows':
        mk = (lambda x: np.asarray(x)) if a.get("numpy") else (lambda x: T(x, d))
        return {"vecs": ofl.combine_flows(mk(i["f1"]), mk(i["f2"]), a["mode"], a["ref"])}
    if op == 'switch_flow_ref':
        return {"vecs": ofl.switch_flow_ref(T(i["f"], d), a["ref"])}
    if op == 'invert_flow':
        return {"vecs": ofl.invert_flow(T(i["f"], d), a["ref"], a["out_ref"])}
    if op.startswith('Flow.') and op[5:] in ('add', 'sub', 'neg', 'mul'):
        f1 = _flow(i, "f1", "m1", a["ref"], d)
        if op == 'Flow.neg':
            out = -f1
        elif op == 'Flow.mul':
            out = f1 * a["scalar"]
        else:
            other = T(i["f2"], d) if a.get("tensor") else _flow(i, "f2", "m2", a["ref"], d)
            out = f1 + other if op == 'Flow.add' else f1 - other
        return {"vecs": out.vecs, "mask": out.mask}
    if op == 'Flow.combine':
        out = _flow(i, "f1", "m1", a["self_ref"], d).combine(_flow(i, "f2", "m2", a["other_ref"], d), a["mode"], a["ref"])
        return {"vecs": out.vecs, "mask": out.mask, "_ref": out.ref}
    if op in ('track_pts', 'Flow.track'):
        fv = T(i["flow_raw"], d) if "flow_raw" in i else \
            T(codec.decode_affine(i["flow__params"], i["flow__delta"], i["flow__esc"])[None], d)
        if op == 'track_pts':
            return {"out": ofl.track_pts(fv, a["ref"], T(i["pts"], d), a["int_out"])}
        fl = Flow(fv, a["ref"], T(i["m"], d))
        if a.get("batch"):
            fl = ofl.batch_flows([fl] * a["batch"])
        o, st = fl.track(T(i["pts"], d), get_valid_status=True)
        return {"out": o, "status": st}
    if op == 'Flow.get_padding':
        return {"_padding": _flow(i, "f", "m", a["ref"], d).get_padding(a["item"])}
    if op == 'get_flow_padding':
        v = T(i["flow_raw"], d)
        arg = {'nchw': v, 'chw': v[0], 'hwc_np': v[1].permute(1, 2, 0).cpu().numpy()}[a["layout"]]
        return {"_padding": ofl.get_flow_padding(arg, a["ref"])}
    if op in ('from_matrix', 'Flow.from_matrix'):
        mat = i["matrix"] if a.get("numpy") else T(i["matrix"], d)
        if op == 'from_matrix':
            return {"out": ofl.from_matrix(mat, a["shape"], a["ref"], a["matrix_is_inverse"])}
        fl = Flow.from_matrix(mat, tuple(a["shape"]), a["ref"], T(i["m"], d), matrix_is_inverse=a["matrix_is_inverse"])
        return {"vecs": fl.vecs, "mask": fl.mask}
    if op in ('from_transforms', 'Flow.from_transforms'):
        tl = [list(t) for t in a["transforms"]]
        if op == 'from_transforms':
            out = ofl.from_transforms(tl, a["shape"], a["ref"], a["padding"])
            return {"out": out.to(d) if d.type == 'cuda' else out, "_device": str(out.device)}
        fl = Flow.from_transforms(tl, tuple(a["shape"]), a["ref"], T(i["m"], d), device=d if d.type == 'cuda' else None, padding=a["padding"])
        return {"vecs": fl.vecs, "mask": fl.mask}
    if op == 'resize_flow':
        f = T(i["flow"], d)
        arg = f[0] if a.get("squeeze") else (f[1].permute(1, 2, 0).contiguous().cpu().numpy() if a.get("hwc_np") else f)
        sc = tuple(a["scale"]) if a.get("tuple") else a["scale"]
        return {"out": ofl.resize_flow(arg, sc)}
    if op in ('Flow.resize', 'Flow.pad', 'Flow.unpad', 'Flow.pad_unpad'):
        fl = _flow(i, "f", "m", a["ref"], d)
        if op == 'Flow.resize':
            if a.get("raises"):
                try:
                    fl.resize(a["scale"])
                except (ValueError, TypeError) as exc:
                    return {"_raised": type(exc).__name__}
                return {"_raised": None}
            out = fl.resize(a["scale"])
        elif op == 'Flow.pad':
            out = fl.pad(a["padding"], a["mode"])
        elif op == 'Flow.unpad':
            out = fl.unpad(a["padding"])
        else:
            out = fl.pad(a["padding"]).unpad(a["padding"])
        return {"vecs": out.vecs, "mask": out.mask, "_ref": out.ref}
    if op == 'raises':
        try:
            _bad_call(a["fn"], [list(x) if isinstance(x, list) else x for x in a["args"]], d)
        except Exception as exc:  # noqa: BLE001
            return {"_raised": type(exc).__name__, "_message": str(exc)}
        return {"_raised": None, "_message": None}
    if op == 'Flow.binop':
        fl = _flow(i, "f", "m", a["ref"], d)
        operand = a["operand"] if "operand" not in i else (np.asarray(i["operand"]) if a.get("numpy") else T(i["operand"], d))
        out = {'div': lambda x, y: x / y, 'pow': lambda x, y: x ** y, 'mul': lambda x, y: x * y}[a["op"]](fl, operand)
        return {"vecs": out.vecs, "mask": out.mask}
    if op in ('Flow.select', 'Flow.getitem', 'Flow.copy'):
        fl = _flow(i, "f", "m", a["ref"], d)
        if op == 'Flow.select':
            out = fl.select(a["item"])
        elif op == 'Flow.getitem':
            out = fl[tuple(slice(*it) if isinstance(it, list) else slice(None) for it in a["item"])]
        else:
            out = fl.copy()
            head = str(out)
            assert head[:head.index(';')].replace(str(out.device), 'cpu') == a["str"], (head, a["str"])
            assert (out.vecs.data_ptr() == fl.vecs.data_ptr()) == a["aliases"]
        return {"vecs": out.vecs, "mask": out.mask}
    if op == 'batch_flows':
        f, m = T(i["f"], d), T(i["m"], d)
        out = ofl.batch_flows([Flow(f[:1], a["ref"], m[:1]), Flow(f, a["ref"], m), Flow(f[1:], a["ref"])])
        return {"vecs": out.vecs, "mask": out.mask}
    if op.startswith('grad_'):
        return run_grad_case(case, golden, device)
    if op == 'kat_gfud':
        v = codec.decode_affine(i["flow__params"], i["flow__delta"], i["flow__esc"])
        vecs = T(np.stack([v, v]), d)                      # the reference test batches two copies
        x, y = ofl.get_flow_endpoints(vecs, 's')
        data, den = ofl.grid_from_unstructured_data(x, y, vecs)
        return {"_data": data, "_density": den}
    if op == 'kat_cfg1':
        fields = _cfg1_fields(golden, d)
        if a["call"] == 'apply':
            w, m = fields[a["flow"]].apply(fields["img"], return_valid_area=True)
            return {"_vec": w, "_mask": m}
        if a["call"] == 'switch_ref':
            out = fields[a["flow"]].switch_ref()
        else:
            out = fields[a["self"]].combine_with(fields[a["flow"]], a["mode"])
        return {"_vec": out.vecs, "_mask": out.mask}
    raise KeyError(op)


def _bad_call(fn, args, d):
    """One row of tests/golden/gen_golden.py's BAD_CALLS, against this package (same calls the reference was given)."""
    f = Flow(torch.ones(2, 2, 12, 16, device=d), 't')
    if fn == 'from_transforms':
        return ofl.from_transforms(*[([list(t) if isinstance(t, list) else t for t in x] if isinstance(x, list) and k == 0 else x)
                                     for k, x in enumerate(args)])
    if fn == 'from_matrix_eye':
        return ofl.from_matrix(torch.eye(3), *args)
    if fn == 'resize_flow_f':
        return ofl.resize_flow(f.vecs, *args)
    if fn == 'flow_pad':
        return f.pad(*args)
    if fn == 'flow_unpad':
        return f.unpad(*args)
    if fn == 'flow_mul':
        return f * args[0]
    if fn == 'flow_div':
        return f / args[0]
    if fn == 'flow_pow':
        return f ** args[0]
    if fn == 'flow_select':
        return f.select(*args)
    if fn == 'flow_apply_kw':
        return f.apply(torch.zeros(2, 1, 12, 16, device=d), **args[0])
    if fn == 'flow_combine_with':
        return f.combine_with(Flow(torch.ones(2, 2, 12, 16, device=d) * 2, 't'), *args)
    if fn == 'flow_switch_ref':
        return f.switch_ref(*args)
    if fn == 'flow_invert':
        return f.invert(*args)
    if fn == 'flow_is_zero':
        return f.is_zero(*args)
    if fn in ('track_pts', 'flow_track'):
        spec = args[0]
        pts = [[1, 2], [3, 4]] if spec["pts"] == 'list' else torch.ones(*spec["pts"], device=d)
        if fn == 'flow_track':
            return f.track(pts, **{k: v for k, v in spec.items() if k != 'pts'})
        vecs = torch.ones(*spec.get("flow_shape", [2, 2, 12, 16]), device=d)
        if spec.get("nan"):
            vecs[0, 0, 0, 0] = float('nan')
        return ofl.track_pts(vecs, spec["ref"], pts, *([spec["int_out"]] if "int_out" in spec else []))
    raise KeyError(fn)


def run_grad_case(case, golden, device):
    """Gradient cases (group 'grads'): the same call with requires_grad inputs, loss = sum(out * w_out), backward."""
    i, _ = golden.arrays(case)
    a, op = case["args"], case["op"]
    d = device
    leaf = lambda k: T(i[k], d).clone().requires_grad_()
    if op == 'grad_apply_flow':
        fv, tv = leaf("flow"), leaf("target")
        out = ofl.apply_flow(fv, tv, a["ref"], T(i.get("mask"), d))
        assert out.grad_fn is not None                       # test_utils.py:500
        (out * T(i["w_out"], d)).sum().backward()
        return {"g_flow": fv.grad, "g_target": tv.grad}
    if op == 'grad_gfud':
        xv, yv, dv = leaf("x"), leaf("y"), leaf("data")
        od, oden = ofl.grid_from_unstructured_data(xv, yv, dv, T(i.get("mask"), d))
        assert od.grad_fn is not None and oden.grad_fn is not None   # test_utils.py:1113-1114
        ((od * T(i["w_data"], d)).sum() + (oden * T(i["w_density"], d)).sum()).backward()
        return {"g_x": xv.grad, "g_y": yv.grad, "g_data": dv.grad}
    if op == 'grad_track_pts':
        fv, pv = leaf("flow"), leaf("pts")
        out = ofl.track_pts(fv, a["ref"], pv)
        assert out.grad_fn is not None
        (out * T(i["w_out"], d)).sum().backward()
        return {"g_flow": fv.grad, "g_pts": pv.grad, "out": out.detach()}
    fa, fb = leaf("f1"), leaf("f2")
    m1, m2 = T(i["m1"], d), T(i["m2"], d)
    if op == 'grad_Flow.apply':
        out = Flow(fa, a["ref"], m1).apply(Flow(fb, a["target_ref"], m2))
    elif op == 'grad_Flow.switch_ref':
        out = Flow(fa, a["ref"], m1).switch_ref()
    elif op == 'grad_Flow.invert':
        out = Flow(fa, a["ref"], m1).invert()
    elif op == 'grad_Flow.combine_with':
        out = Flow(fa, a["ref"], m1).combine_with(Flow(fb, a["ref"], m2), a["mode"])
    elif op == 'grad_Flow.combine':
        out = Flow(fa, a["self_ref"], m1).combine(Flow(fb, a["other_ref"], m2), a["mode"], a["ref"])
    else:
        raise KeyError(op)
    assert out.vecs.grad_fn is not None
    (out.vecs * T(i["w_out"], d)).sum().backward()
    z = lambda t, like: torch.zeros_like(like) if t.grad is None else t.grad
    return {"g_f1": z(fa, fa), "g_f2": z(fb, fb), "vecs": out.vecs.detach(), "mask": out.mask}


def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


GRAD_RTOL = 2e-4     # gradients: fp32 sums in a different order than ATen's CPU loops (and atomics on the device);
                     # bar: |got - expected| <= GRAD_RTOL * max|expected| per tensor (stated in DESIGN.md)


def check_case(case, golden, got, exact_values=True, rtol=0.0, atol=0.0, max_mask_flips=0):
    """Compare `got` (from run_case) with the reference's outputs.  Masks / bools must match bit for bit
    (up to `max_mask_flips`, only ever > 0 for the atomics-ordered GPU splat); floats exactly or within tol."""
    _, exp = golden.arrays(case)
    a, op = case["args"], case["op"]
    report = {}
    if op == 'kat_gfud':
        den, data = _np(got["_density"]), _np(got["_data"])
        cnt = int(np.count_nonzero(den > 0.95))
        assert abs(cnt - a["count_batch_of_two"]) <= 2 * max_mask_flips, (cnt, a["count_batch_of_two"])
        m = np.unpackbits(exp["density_gt_095_packed"])[:den[0].size].reshape(den[0].shape).astype(bool)
        assert np.count_nonzero((den[0] > 0.95) != m) <= max_mask_flips
        _cmp(codec.subsample(data[:1], 4), exp["data_sub4"], exact_values, rtol, atol, "data")
        _cmp(codec.subsample(den[:1], 4), exp["density_sub4"], exact_values, rtol, atol, "density")
        return report
    if op == 'kat_cfg1':
        mask, vec = _np(got["_mask"]), _np(got["_vec"])
        m = np.unpackbits(exp["mask_packed"])[:mask.size].reshape(mask.shape).astype(bool)
        flips = int(np.count_nonzero(mask != m))
        assert flips <= max_mask_flips, "mask flips %d" % flips
        if max_mask_flips == 0:
            assert int(mask.sum()) == a["mask_count"]
        _cmp(codec.subsample(vec, 4), exp["sub4"], exact_values, rtol, atol, "vec")
        s = float(vec.astype(np.float64).sum())
        assert abs(s - a["sum"]) <= 1e-6 * max(a["abs_sum"], 1.0), (s, a["sum"])
        return report
    if "raises" in a and a["raises"]:
        assert got["_raised"] == a["raises"], (got["_raised"], a["raises"])
        return report
    if op == 'raises':                                       # the exception class AND the reference's message
        assert got["_raised"] == a["exception"], (a["fn"], a["args"], got["_raised"], a["exception"])
        assert got["_message"] == a["message"], (got["_message"], a["message"])
        return report
    if op in ('Flow.get_padding', 'get_flow_padding'):
        assert got["_padding"] == a["padding"], (got["_padding"], a["padding"])
        return report
    for k, e in exp.items():
        g = _np(got[k])
        assert g.shape == e.shape, "%s: shape %s != %s" % (k, g.shape, e.shape)
        if k.startswith("g_"):                               # a gradient
            assert g.dtype == e.dtype
            scale = float(np.max(np.abs(e))) if e.size else 0.0
            err = float(np.max(np.abs(g.astype(np.float64) - e.astype(np.float64)))) if e.size else 0.0
            assert err <= GRAD_RTOL * max(scale, 1e-6), "%s: max |diff| %.3g vs gradient scale %.3g" % (k, err, scale)
            continue
        assert g.dtype == e.dtype, "%s: dtype %s != %s" % (k, g.dtype, e.dtype)
        if e.dtype == np.bool_:
            flips = int(np.count_nonzero(g != e))
            assert flips <= max_mask_flips, "%s: %d mask bits differ" % (k, flips)
        else:
            _cmp(g, e, exact_values, rtol, atol, k)
    if "out_ref" in a and "_ref" in got and op != 'invert_flow':
        assert got["_ref"] == a["out_ref"]
    if "returns" in a and "_returns" in got:
        assert got["_returns"] == a["returns"], (got["_returns"], a["returns"])
    if "same_object" in a and "_same" in got:
        assert got["_same"] == a["same_object"]
    return report


def _cmp(g, e, exact, rtol, atol, name):
    if exact:
        if g.dtype.kind == 'f':
            bad = ~((g == e) | (np.isnan(g) & np.isnan(e)))
        else:
            bad = g != e
        n = int(np.count_nonzero(bad))
        if n:
            idx = np.argwhere(bad)[0]
            raise AssertionError("%s: %d of %d values differ, first at %s: got %r expected %r"
                                 % (name, n, g.size, tuple(idx), g[tuple(idx)], e[tuple(idx)]))
    else:
        np.testing.assert_allclose(g, e, rtol=rtol, atol=atol, err_msg=name)
