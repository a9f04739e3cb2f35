"""GPU tier: the validation read-back (`ofl_flow_flags_host`: the reduction's last block writes the flag words to host-visible
memory), the flag cache against in-place edits (fp16-stored flows whose `.vecs` was handed out; inference tensors),
and the batch-sharding code on a real RCCL communicator (world size 1, in a child process of its own).

Checker: the CPU oracle (`oracle/`) on the same seeded inputs; reference semantics utils.py:98, 497-498, 919-938,
flow_class.py:1046, 1226-1244, 1729-1744."""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu tier needs a HIP device"
    from oflibpytorch_amd import _native
    _native.load_library()
    return torch.device('cuda', 0)


def _oracle_flags(v, m):
    from oracle import oracle
    return [int(x) for x in oracle.flow_flags(v.float().cpu().numpy(), None if m is None else m.cpu().numpy())]


@pytest.mark.parametrize("shape", [(1, 8, 8), (3, 33, 47), (5, 120, 160), (2, 301, 403), (64, 64, 96), (1, 1080, 1920)])
@pytest.mark.parametrize("masked", [False, True])
def test_host_flag_words_equal_the_device_words_and_the_oracle(shape, masked, dev):
    from oflibpytorch_amd import _native
    n, h, w = shape
    g = torch.Generator().manual_seed(n * 1000 + h)
    v = torch.randn(n, 2, h, w, generator=g)
    v[0] = 0.0                                       # an all-zero element
    if n > 1:
        v[1] = v[1] * 1e-4                            # below the threshold everywhere
    if n > 2:
        v[2, 1, h // 2, w // 3] = float('nan')
    m = None
    if masked:
        m = torch.rand(n, h, w, generator=g) > 0.3
        if n > 2:
            m[2, h // 2, w // 3] = False              # the NaN sits under the mask: still non-finite (utils.py:98 ignores masks)
    vd, md = v.to(dev), None if m is None else m.to(dev)
    for _ in range(3):                                # the work words are left clean by every call
        host = _native.flow_flags_host(vd, md)
        assert host is not None
        assert host == _native.flow_flags(vd, md).cpu().tolist() == _oracle_flags(v, m)


def test_host_flag_words_fp16_and_broadcast_mask(dev):
    from oflibpytorch_amd import _native
    g = torch.Generator().manual_seed(5)
    v = (torch.randn(4, 2, 64, 96, generator=g) * 3).half()
    v[3] = 0
    m = torch.rand(1, 64, 96, generator=g) > 0.5
    host = _native.flow_flags_host(v.to(dev), m.to(dev))
    assert host == _oracle_flags(v, m.expand(4, -1, -1))
    # a layout the vector kernel does not take (H * W not a multiple of 4): the route declines, the constructor still validates
    v2 = (torch.randn(2, 2, 5, 7, generator=g)).half().to(dev)
    assert _native.flow_flags_host(v2, None) is None
    import oflibpytorch_amd as ofl
    assert not bool(ofl.Flow(v2, 't').is_zero().any())


def test_constructor_raises_on_non_finite_through_the_host_route(dev):
    import oflibpytorch_amd as ofl
    v = torch.zeros(2, 2, 40, 52, device=dev)
    ofl.Flow(v)                                       # fine
    v[1, 0, 3, 4] = float('inf')
    with pytest.raises(ValueError, match="NaN, Inf or -Inf"):
        ofl.Flow(v)
    with pytest.raises(ValueError, match="NaN, Inf or -Inf"):
        ofl.apply_flow(v, torch.zeros(2, 1, 40, 52, device=dev), 't')


def test_host_route_from_two_threads(dev):
    """One slot per device, shared: two threads validating different tensors must each get their own words."""
    from oflibpytorch_amd import _native
    a = torch.zeros(6, 2, 200, 300, device=dev)
    b = torch.ones(3, 2, 100, 140, device=dev)
    want_a, want_b = _native.flow_flags(a).cpu().tolist(), _native.flow_flags(b).cpu().tolist()
    bad = []

    def work(t, want, stream):
        with torch.cuda.stream(stream):
            for _ in range(200):
                if _native.flow_flags_host(t, None) != want:
                    bad.append(1)
    ths = [threading.Thread(target=work, args=(a, want_a, torch.cuda.Stream(dev))),
           threading.Thread(target=work, args=(b, want_b, torch.cuda.Stream(dev)))]
    torch.cuda.synchronize()
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad


def test_host_route_with_more_threads_than_slots_and_an_interrupted_wait(dev, monkeypatch):
    """The slots are a pool of _MAX_SLOTS per device: a dozen threads queue for them and each still gets its own words; a wait
    that is interrupted (KeyboardInterrupt in the poll) retires its slot -- nobody is handed its possibly dirty work words,
    not even a thread that was queueing for it -- and the next call works."""
    from oflibpytorch_amd import _native
    tensors = [torch.full((1 + k % 3, 2, 64 + 8 * k, 96), float(k % 2), device=dev) for k in range(12)]
    wants = [_native.flow_flags(t).cpu().tolist() for t in tensors]
    bad = []

    def work(t, want):
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            for _ in range(100):
                if _native.flow_flags_host(t, None) != want:
                    bad.append(1)
    ths = [threading.Thread(target=work, args=(t, w)) for t, w in zip(tensors, wants)]
    torch.cuda.synchronize()
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad
    pool = _native._host_slots[dev.index]
    assert 1 <= len(pool) <= _native._MAX_SLOTS

    class Interrupting(object):                # stands in for the `time` module inside _native: the first look at the clock raises
        def __init__(self):
            self.armed = True

        def perf_counter(self):
            if self.armed:
                self.armed = False
                raise KeyboardInterrupt()
            return time.perf_counter()
        sleep = staticmethod(time.sleep)
    before = list(pool)
    monkeypatch.setattr(_native, "time", Interrupting())
    with pytest.raises(KeyboardInterrupt):
        _native.flow_flags_host(tensors[0], None)
    monkeypatch.undo()
    after = _native._host_slots[dev.index]
    retired = [s for s in before if s[5]]
    assert len(retired) == 1 and all(s is not retired[0] for s in after)
    assert not retired[0][0].locked()
    for t, w in zip(tensors, wants):
        assert _native.flow_flags_host(t, None) == w


def test_fp16_flow_vecs_edit_reaches_the_kernels(dev):
    """`.vecs` of an fp16-constructed flow IS its storage once handed out -- an in-place edit must change what
    apply / switch_ref read, and must be re-validated."""
    import oflibpytorch_amd as ofl
    from oracle import oracle
    g = torch.Generator().manual_seed(3)
    v16 = (torch.randn(2, 2, 48, 64, generator=g) * 2).half()
    img = torch.rand(2, 3, 48, 64, generator=g)
    f = ofl.Flow(v16.to(dev), 't')
    before = f.apply(img.to(dev)).cpu().numpy()
    assert np.array_equal(before, oracle.flow_apply(v16.float().numpy(), 't', np.ones((2, 48, 64), bool), img.numpy(), None)[0])
    f.vecs[:, 0] += 1.5                               # the reference's idiom: edit the flow through .vecs
    edited = v16.float()
    edited[:, 0] += 1.5
    after = f.apply(img.to(dev)).cpu().numpy()
    assert np.array_equal(after, oracle.flow_apply(edited.numpy(), 't', np.ones((2, 48, 64), bool), img.numpy(), None)[0])
    assert not np.array_equal(after, before)
    assert torch.equal(f.copy().vecs.cpu(), edited)
    f.vecs[0, 0, 0, 0] = float('nan')                 # ... and a NaN written in place is caught by the next operation
    with pytest.raises(ValueError):
        f.apply(img.to(dev))
    # a flow whose .vecs nobody asked for stays in fp16
    h = ofl.Flow(v16.to(dev), 's')
    h.switch_ref()
    assert h._half is not None


def test_inference_mode_inplace_edit_is_seen(dev):
    """inference tensors carry no version counter, so their flag words are never cached across calls."""
    import oflibpytorch_amd as ofl
    from oracle import oracle
    g = torch.Generator().manual_seed(9)
    img = torch.rand(1, 2, 40, 56, generator=g)
    with torch.inference_mode():
        f = ofl.Flow(torch.zeros(1, 2, 40, 56, device=dev))
        tgt = img.to(dev)
        assert f.apply(tgt) is not None and bool(f.is_zero().all())
        f.vecs[:] = 5.0
        assert not bool(f.is_zero().any())
        out = f.apply(tgt).cpu().numpy()
        exp = oracle.flow_apply(np.full((1, 2, 40, 56), 5.0, np.float32), 't', np.ones((1, 40, 56), bool), img.numpy(), None)[0]
        assert np.array_equal(out, exp)
        f.vecs[0, 1, 2, 3] = float('nan')
        with pytest.raises(ValueError):
            f.apply(tgt)


def test_flag_cache_still_caches_ordinary_tensors(dev, monkeypatch):
    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native
    calls = []
    real = _native.flow_flags_host
    monkeypatch.setattr(_native, "flow_flags_host", lambda v, m=None: (calls.append(1), real(v, m))[1])
    f = ofl.Flow(torch.randn(2, 2, 32, 48, device=dev), 't')
    img = torch.rand(2, 1, 32, 48, device=dev)
    f.apply(img); f.apply(img); f.is_zero()
    assert len(calls) == 1
    f.vecs[0, 0, 0, 0] = 2.0                          # version bump: one more reduction, then cached again
    f.apply(img); f.apply(img)
    assert len(calls) == 2


_RCCL_CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)        # RCCL, before any other GPU work of this process
import oflibpytorch_amd as ofl
from oflibpytorch_amd import _native, distributed as ofd
from oracle import oracle
ofd.enable_batch_sharding()
assert not ofd.is_enabled()                      # one rank: nothing to reduce over ...
g = torch.Generator().manual_seed(21)
f1 = torch.randn(3, 2, 60, 84, generator=g) * 3; f2 = torch.randn(3, 2, 60, 84, generator=g) * 2
img = torch.rand(3, 3, 60, 84, generator=g); m1 = torch.rand(3, 60, 84, generator=g) > 0.2
# ... so force the sharded code path itself (collectives over the real communicator of size 1)
ofd._force_collectives = True
assert ofd.is_enabled()
assert ofd.host_exchange_active()                # one node: the 5-bit words cross the ranks through shared memory ...
A, B = ofl.Flow(f1.to(dev), 't', m1.to(dev)), ofl.Flow(f2.to(dev), 't')
assert A._flag_cache[2] == (True, A._batch_flags())
ofd.USE_HOST_EXCHANGE = False                    # ... and everything below takes the communicator route (RCCL all-reduce)
ofd.enable_batch_sharding()
assert ofd.is_enabled() and not ofd.host_exchange_active()
A, B = ofl.Flow(f1.to(dev), 't', m1.to(dev)), ofl.Flow(f2.to(dev), 't')
w, v = B.apply(img.to(dev), return_valid_area=True)
ow, ov = oracle.flow_apply(f2.numpy(), 't', np.ones((3, 60, 84), bool), img.numpy(), None)
assert np.array_equal(w.cpu().numpy(), ow) and np.array_equal(v.cpu().numpy(), ov)
C = A.combine_with(B, 3)
o3, om3, _ = oracle.combine_with(f1.numpy(), m1.numpy(), f2.numpy(), np.ones((3, 60, 84), bool), 3, 't')
assert np.array_equal(C.vecs.cpu().numpy(), o3) and np.array_equal(C.mask.cpu().numpy(), om3)
S = ofl.Flow(f1.to(dev), 's', m1.to(dev)).switch_ref()
assert S.ref == 't' and bool(torch.isfinite(S.vecs).all())
words = _native.flow_flags(f1.to(dev), m1.to(dev))
out = ofd.with_global_or(words)                  # device words -> words + 5 OR bits, all-reduced (MAX) on a slice view
host = out.cpu().tolist()
ws, glob = ofd.split_global_or(host)
want = oracle.flow_flags(f1.numpy(), m1.numpy())
assert ws == [int(x) for x in want] and glob == int(np.bitwise_or.reduce(np.asarray(want)))
z = ofl.Flow(torch.zeros(3, 2, 60, 84, device=dev), 't')
assert z.combine_with(B, 3) is B                 # batch-global early exit through the collective
assert ofd.reduce_flags(5, dev) == 5
b = ofd.broadcast_operand(img[:1].to(dev)); assert torch.equal(b.cpu(), img[:1])
gth = ofd.all_gather_batch(w); assert torch.equal(gth, w)
bad = f1.clone(); bad[2, 0, 5, 5] = float("nan")
try:
    ofl.Flow(bad.to(dev), 't'); raise SystemExit("no ValueError for a NaN under sharding")
except ValueError:
    pass
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_CHILD_OK")
'''


def _free_port() -> str:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


def test_sharded_path_on_a_real_rccl_communicator(dev, tmp_path):
    """`init_process_group("nccl")`, `enable_batch_sharding()`, Flow / apply / combine_with / with_global_or /
    broadcast_operand / all_gather_batch on the HIP path against the oracle -- in a fresh child, RCCL first."""
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script), ROOT, _free_port()], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "RCCL_CHILD_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_bench_runs_under_torch_distributed_run_with_one_rank(dev):
    """bench.py's launcher path (RANK / WORLD_SIZE from the environment, init_process_group("nccl"), barriers) at N = 1."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-probe", "--no-secondary", "--blocks", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["launcher"] == "torch.distributed (nccl)"


def test_bench_starts_its_own_ranks_when_not_under_a_launcher(dev):
    """plain `python bench.py --gpus N` (no RANK in the environment) must start its ranks itself -- fresh child processes
    through torch.distributed.run, before the parent touches the GPU -- and relay rank 0's line.  One GPU here: `--self-launch`
    takes the same path at N = 1 (N > 1 takes it by itself); the line must say so, carry the communicator size seen inside the timed
    region and the kernel the launcher really picked."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--self-launch", "--steps", "3", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-probe", "--no-secondary", "--blocks", "1", "--n1-ms", "1.0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly ONE JSON line reaches the caller
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["launcher"] == "torch.distributed (nccl)"
    assert out["rccl_ranks"] == 1 and out["speedup_vs_n1"] > 0
    assert "warp_bwd" in out["roofline"]["kernel"] and "warp_bwd" in out["kernels"]["combine3_kernel"]
    # a child that fails must fail the parent, with the child's stderr
    bad = subprocess.run(cmd + ["--height", "-5"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert bad.returncode != 0 and "failed with exit code" in bad.stderr


def test_bench_shared_image_and_ragged_strong_scaling_under_torch_distributed_run(dev):
    """`bench.py --shared-image` drives the reference's 1 <-> N broadcast case (utils.py:527-537; flow_class.py:896-897)
    through the only data-path collective the path has -- a B = 1 image + target mask broadcast from rank 0 (RCCL, forced on at one
    rank) inside the timed region, warped through the stride-0 batch broadcast -- and `--scaling strong --batch 5` takes a batch that
    does not divide.  World size 2 runs in the gloo tier (tests/test_distributed_gloo.py, section 3b)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "5",
           "--scaling", "strong", "--shared-image", "--force-collectives", "--no-cpu-baseline", "--no-probe", "--no-secondary", "--blocks", "3"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "strong" and out["config"]["global_batch"] == 5
    assert "broadcast" in out["config"]["parallelism"] and out["broadcast_ms"] is not None and out["broadcast_ms"] > 0
    assert abs(out["config"]["bytes_per_px"] - (22 + 13 / 5 + 27)) < 1e-2


def test_kernel_outputs_keep_their_flag_words_under_inference_mode(dev, monkeypatch):
    """under torch.inference_mode() tensors carry no version counter, so the flag cache of a flow never matched and
    every use re-ran the reduction and waited for it -- also for kernel OUTPUTS nobody else holds.  Those are now `private`:
    their by-product flag word is used; handing the storage out (`.vecs` / `.mask`) ends it, and an in-place edit through the
    handed-out tensor is seen by the next use."""
    import oflibpytorch_amd as ofl
    from oflibpytorch_amd import _native, flow_class
    from oracle import oracle
    calls = {"n": 0}
    real = flow_class._host_flags

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(flow_class, "_host_flags", counting)
    g = torch.Generator().manual_seed(5)
    f1 = (torch.randn(2, 2, 40, 56, generator=g) * 2).to(dev)
    f2 = (torch.randn(2, 2, 40, 56, generator=g) * 2).to(dev)
    with torch.inference_mode():
        a, b = ofl.Flow(f1, 't'), ofl.Flow(f2, 't')
        base = calls["n"]
        c = a.combine_with(b, 3)                    # a kernel output: private
        assert c._private
        n0 = calls["n"]
        d = c.combine_with(b, 3)                    # its flag word is needed here
        d2 = c.combine_with(b, 3)
        # (a and b are the caller's inference tensors: re-validated per use, as before; c must not add a reduction of its own
        # beyond the one that reads its by-product / first word)
        per_use = (calls["n"] - n0) / 2.0
        assert torch.equal(d.vecs, d2.vecs)
        handed = c.vecs                             # the storage escapes
        assert not c._private
        handed.zero_()                              # ... and is edited in place: c is now the zero flow
        e = c.combine_with(b, 3)                    # zero first operand: the second comes back (flow_class.py:1729-1737)
        assert e is b
        # another flow object over the same storage ends the privacy too: an edit through the copy must reach the original
        c2 = a.combine_with(b, 3)
        assert c2._private
        twin = c2.copy()
        assert not c2._private and not twin._private
        twin.vecs.zero_()
        assert c2.combine_with(b, 3) is b
        c3 = a.combine_with(b, 3)
        relabelled = c3.switch_ref(mode='invalid')  # same tensors under the other reference
        assert not c3._private
        relabelled.vecs.zero_()
        assert c3.combine_with(b, 3) is b
        c4 = a.combine_with(b, 3)
        _ = c4.invert()                             # an internal temporary over c4's tensors does not end it
        assert c4._private
        # the public setters store the caller's tensor as it is -- privacy must end, so that an in-place edit of
        # that tensor is seen by the next use
        c5 = a.combine_with(b, 3)
        assert c5._private
        v = torch.zeros(2, 2, 40, 56, device=dev)
        c5.vecs = v
        assert not c5._private
        assert c5.combine_with(b, 3) is b           # the zero flow: the second operand comes back
        v += 3.0
        shifted = c5.combine_with(b, 3)
        assert shifted is not b
        assert torch.equal(shifted.vecs, ofl.Flow(v.clone(), 't').combine_with(b, 3).vecs)
        c6 = a.combine_with(b, 3)
        m = torch.zeros(2, 40, 56, dtype=torch.bool, device=dev)
        c6.mask = m
        assert not c6._private
        k0 = c6._flags()
        m.fill_(True)
        assert not c6._flags_known()                # the edit of the caller's mask invalidates the word
        assert c6._flags() != k0                    # (masked bits: none under an all-False mask, set under an all-True one)
    assert per_use <= 2.0
