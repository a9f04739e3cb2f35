"""CPU tier, world_size 2 over gloo: the batch-sharded path (oflibpytorch_amd.distributed).

Shards never exchange pixels; what must cross ranks is the reference's BATCH-GLOBAL early-exit decision
(`all(is_zero_flow(...))`, utils.py:497; flow_class.py:1729-1744): a rank whose shard is all-zero must still warp when
another rank's shard is not.  The native primitives are served by the CPU oracle in each rank (tests/oracle_backend.py).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        for pth in (ROOT, os.path.join(ROOT, "tests")):
            if pth not in sys.path:
                sys.path.insert(0, pth)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle_backend
        import oflibpytorch_amd as ofl
        from oflibpytorch_amd import _native, distributed as ofd
        from oracle import oracle
        for name in ("flow_flags", "warp_bwd", "splat_fwd", "device"):
            setattr(_native, name, getattr(oracle_backend, name))

        # full batch of 4, identical on both ranks; rank r owns elements [2r, 2r+2)
        g = torch.Generator().manual_seed(3)
        h, w = 20, 28
        f1 = torch.zeros(4, 2, h, w)
        f1[2:] = torch.randn(2, 2, h, w, generator=g) * 2      # rank 0's shard of `self` is ALL ZERO, rank 1's is not
        f2 = torch.randn(4, 2, h, w, generator=g) * 2
        m1 = torch.rand(4, h, w, generator=g) > 0.1
        m2 = torch.rand(4, h, w, generator=g) > 0.1
        lo, hi = ofd.shard_bounds(4, rank, world)
        assert (lo, hi) == (2 * rank, 2 * rank + 2)

        # 1. without sharding rank 0 takes the reference's early exit (returns `flow` itself) ...
        a, b = ofl.Flow(f1[lo:hi], 't', m1[lo:hi]), ofl.Flow(f2[lo:hi], 't', m2[lo:hi])
        out_local = a.combine_with(b, 3)
        assert (out_local is b) == (rank == 0)
        # 2. ... with batch sharding the decision is batch-global: nobody exits, results equal the un-sharded oracle
        ofd.enable_batch_sharding()
        assert ofd.is_enabled()
        assert ofd.host_exchange_active()        # both ranks on this host: the flag words cross through shared memory
        a, b = ofl.Flow(f1[lo:hi], 't', m1[lo:hi]), ofl.Flow(f2[lo:hi], 't', m2[lo:hi])
        out = a.combine_with(b, 3)
        assert out is not b and out is not a
        ev, em, _ = oracle.combine_with(f1.numpy(), m1.numpy(), f2.numpy(), m2.numpy(), 3, 't')
        assert np.array_equal(out.vecs.numpy(), ev[lo:hi]) and np.array_equal(out.mask.numpy(), em[lo:hi])
        # apply(): thresholded zero test of the warping flow is batch-global too
        img = torch.rand(4, 3, h, w, generator=g)
        wv, vv = ofl.Flow(f1[lo:hi], 't', m1[lo:hi]).apply(img[lo:hi], return_valid_area=True)
        ow, ov = oracle.flow_apply(f1.numpy(), 't', m1.numpy(), img.numpy())
        assert np.array_equal(wv.numpy(), ow[lo:hi]) and np.array_equal(vv.numpy(), ov[lo:hi])
        # non-finite input on ONE rank raises on EVERY rank (isfinite().all() is batch-global, utils.py:98)
        bad = f2.clone()
        bad[3, 0, 0, 0] = float('nan')
        with pytest.raises(ValueError):
            ofl.Flow(bad[lo:hi], 't')
        # 3. helpers
        assert ofd.reduce_flags(1 << rank, torch.device('cpu')) == 0b11
        shared = torch.full((1, 3, 4, 4), 7.0) if rank == 0 else torch.zeros(1, 3, 4, 4)
        assert float(ofd.broadcast_operand(shared, src=0).sum()) == 7.0 * 48
        gathered = ofd.all_gather_batch(out.vecs)
        assert np.array_equal(gathered.numpy(), ev)
        # ragged shards (3 elements over 2 ranks: 2 + 1; 1 element: 1 + 0) gather in order
        for total in (3, 1):
            a0, a1 = ofd.shard_bounds(total, rank, world)
            part = torch.arange(float(total))[a0:a1].reshape(-1, 1, 1) * torch.ones(1, 2, 3)
            assert ofd.all_gather_batch(part)[:, 0, 0].tolist() == [float(v) for v in range(total)]
        # device-side words: the local words come back unchanged, followed by the bits of the OR over both ranks
        words = ofd.with_global_or(torch.tensor([1 << rank, 4], dtype=torch.int32))
        assert words.tolist() == [1 << rank, 4, 1, 1, 1, 0, 0]
        assert ofd.split_global_or(words.tolist()) == ([1 << rank, 4], 0b111)
        # many exchanges back to back (the ring of slots is re-used; the ranks run at different speeds)
        for i in range(200):
            if rank == 1 and i % 37 == 0:
                import time
                time.sleep(0.002)
            assert ofd.reduce_flags(((i >> rank) & 1) << rank, torch.device('cpu')) == ((i & 1) | (((i >> 1) & 1) << 1))
        ofd.disable_batch_sharding()
        assert not ofd.is_enabled()
        # 4. the communicator route (ranks on different nodes, or USE_HOST_EXCHANGE off): same decisions, same results
        ofd.USE_HOST_EXCHANGE = False
        ofd.enable_batch_sharding()
        assert ofd.is_enabled() and not ofd.host_exchange_active()
        a, b = ofl.Flow(f1[lo:hi], 't', m1[lo:hi]), ofl.Flow(f2[lo:hi], 't', m2[lo:hi])
        out2 = a.combine_with(b, 3)
        assert out2 is not b and np.array_equal(out2.vecs.numpy(), ev[lo:hi]) and np.array_equal(out2.mask.numpy(), em[lo:hi])
        assert ofd.reduce_flags(1 << rank, torch.device('cpu')) == 0b11
        with pytest.raises(ValueError):
            ofl.Flow(bad[lo:hi], 't')
        ofd.disable_batch_sharding()
        ofd.USE_HOST_EXCHANGE = True
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (exc, traceback.format_exc())))


def test_batch_sharding_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results


def test_shard_bounds_ragged():
    from oflibpytorch_amd import distributed as ofd
    assert [ofd.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert ofd.shard_bounds(2, 3, 4) == (2, 2)
