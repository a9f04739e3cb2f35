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
        # 3b. the reference's 1 <-> N broadcast (utils.py:527-537; flow_class.py:896-897) END TO END under sharding: ONE image and ONE
        # target mask live on rank 0, travel with `broadcast_operand` (the only data-path collective the path has), and every rank
        # warps it with its own shard of the flows through the stride-0 batch broadcast -- equal to the unsharded oracle
        one_img = torch.rand(1, 3, h, w, generator=g)
        one_tm = torch.rand(1, h, w, generator=g) > 0.2
        mine_img = ofd.broadcast_operand(one_img.clone() if rank == 0 else torch.zeros(1, 3, h, w), src=0)
        mine_tm = ofd.broadcast_operand((one_tm.clone() if rank == 0 else torch.zeros(1, h, w, dtype=torch.bool)).to(torch.uint8), src=0).bool()
        assert torch.equal(mine_img, one_img) and torch.equal(mine_tm, one_tm)
        sw, sv = ofl.Flow(f2[lo:hi], 't', m2[lo:hi]).apply(mine_img, target_mask=mine_tm, return_valid_area=True)
        ew, evalid = oracle.flow_apply(f2.numpy(), 't', m2.numpy(), one_img.expand(4, -1, -1, -1).numpy(), one_tm.expand(4, -1, -1).numpy())
        assert sw.shape == (2, 3, h, w) and np.array_equal(sw.numpy(), ew[lo:hi]) and np.array_equal(sv.numpy(), evalid[lo:hi])
        # ... and the other way round: ONE flow (rank 0's) warps every rank's shard of a batch of images
        one_flow = ofd.broadcast_operand(f2[0:1].clone() if rank == 0 else torch.zeros(1, 2, h, w), src=0)
        bw = ofl.Flow(one_flow, 't').apply(img[lo:hi])
        eb, _ = oracle.flow_apply(f2[0:1].expand(4, -1, -1, -1).numpy(), 't', np.ones((4, h, w), bool), img.numpy())
        assert np.array_equal(bw.numpy(), eb[lo:hi])
        # ragged strong scaling (bench.py --scaling strong --batch 3 on 2 ranks: 2 + 1): unequal shards, same batch-global decisions
        r0, r1 = ofd.shard_bounds(3, rank, world)
        assert (r1 - r0) == (2 if rank == 0 else 1)
        rg = ofl.Flow(f1[1:4][r0:r1], 't', m1[1:4][r0:r1]).combine_with(ofl.Flow(f2[1:4][r0:r1], 't', m2[1:4][r0:r1]), 3)
        assert rg.vecs.shape[0] == r1 - r0 and np.array_equal(ofd.all_gather_batch(rg.vecs).numpy(), ev[1:4])
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
        # 5. ADVICE r4: (a) or_reduce from two threads of one rank -- the lock makes (sequence number, slot store, reads) one step
        ofd.enable_batch_sharding()
        assert ofd.host_exchange_active()
        import threading
        got = []

        def many():
            for _ in range(100):
                got.append(ofd.reduce_flags(1 << rank, torch.device('cpu')))
        ths = [threading.Thread(target=many) for _ in range(2)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert got == [0b11] * 200
        assert ofd._exchange_timeout(None) >= 600.0       # the process group's own timeout (gloo: 30 min), not a private 120 s
        ofd.EXCHANGE_TIMEOUT_SECONDS = 7.5                # read at every exchange, so setting it AFTER enable_batch_sharding() counts
        assert ofd._exchange_timeout(ofd._exchange.group) == 7.5
        ofd.EXCHANGE_TIMEOUT_SECONDS = None
        ofd.disable_batch_sharding()
        # 6. VERDICT r5: a persistent Flow round a receive buffer must not keep the flag word of the buffer's PREVIOUS contents.
        # c10d's in-place broadcast leaves `_version` alone (asserted here), so broadcast_operand bumps it itself: the Flow built
        # on the all-zero buffer took the zero-flow early exit (returns its target), after the broadcast of rank 0's flow it warps.
        buf = torch.zeros(1, 2, h, w)
        shared = ofl.Flow(buf, 't')
        img = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(11))
        assert shared.apply(img) is img                   # utils.py:497-498: all-zero flow -> the target itself
        probe = torch.zeros(3)
        v0 = probe._version
        dist.broadcast(probe, src=0)
        assert probe._version == v0                       # (the premise: c10d does not bump versions)
        new_flow = torch.randn(1, 2, h, w, generator=torch.Generator().manual_seed(12)) * 3
        if rank == 0:
            buf.copy_(new_flow)
            buf = buf.detach()
        v0 = shared.vecs._version
        got = ofd.broadcast_operand(shared.vecs, src=0)
        assert got is shared.vecs and shared.vecs._version > v0
        assert torch.equal(shared.vecs, new_flow)
        warped = shared.apply(img)
        assert warped is not img
        exp, _ = oracle.flow_apply(new_flow.numpy(), 't', None, img.numpy(), None)
        assert np.array_equal(warped.numpy(), exp)
        # (b) rank 0 cannot create the segment: EVERY rank falls back to the communicator route together (nobody is left in a collective)
        from multiprocessing import shared_memory
        real_shm = shared_memory.SharedMemory

        def failing(*a, **k):
            if k.get("create"):
                raise OSError("no shared memory here")
            return real_shm(*a, **k)
        shared_memory.SharedMemory = failing if rank == 0 else real_shm
        try:
            ofd.enable_batch_sharding()
            assert ofd.is_enabled() and not ofd.host_exchange_active()
            assert ofd.reduce_flags(1 << rank, torch.device('cpu')) == 0b11
            ofd.disable_batch_sharding()
        finally:
            shared_memory.SharedMemory = real_shm
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
    assert [ofd.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert ofd.shard_bounds(2, 3, 4) == (2, 2) and ofd.shard_bounds(2, 1, 4) == (1, 2)
    # BASELINE configs[3] with a batch that does not divide (bench.py --scaling strong --batch 60 on 8 GPUs): 8, 8, 8, 8, 7, 7, 7, 7
    b = [ofd.shard_bounds(60, r, 8) for r in range(8)]
    assert [hi - lo for lo, hi in b] == [8, 8, 8, 8, 7, 7, 7, 7] and b[0][0] == 0 and b[-1][1] == 60
    assert all(b[i][1] == b[i + 1][0] for i in range(7))
