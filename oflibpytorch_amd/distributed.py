"""Batch-axis sharding of the hot path over the GPUs of one node (one process per GPU, torch.distributed,
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The path is embarrassingly parallel along N (every kernel is per (n, pixel)), so shards never exchange
pixels.  The only cross-rank state the reference's semantics need is its BATCH-GLOBAL early exits
(`all(is_zero_flow(flow))`, utils.py:497; `all(self.is_zero())`, flow_class.py:1046, 1729, 1738; the
`isfinite().all()` validation, utils.py:98): a shard that is locally all-zero must still warp if any other
shard is not.  Those decisions are a 5-bit flag word per operand; `reduce_flags` ORs it over the ranks with
one tiny all-reduce, cached per tensor version like the local flags.  A shared B=1 operand (the reference's
1<->N broadcast) is distributed with `broadcast_operand`; `all_gather_batch` reassembles results when asked.
"""
import atexit
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

_group = None
_enabled = False
_force_collectives = False   # tests only: run the collectives on a communicator of ONE rank too (tests/test_gpu_validation.py)
_exchange = None             # _HostExchange of the group when every rank lives on this host (the 8 GPUs of one node), else None
USE_HOST_EXCHANGE = True     # False: the flag words always travel through the communicator (RCCL / gloo), as in rounds 1-3
EXCHANGE_TIMEOUT_SECONDS = None   # how long a rank waits for a peer's flag word; None: the process group's own timeout (RCCL /
#                                   gloo tolerate that much skew between ranks -- a rank that checkpoints or compiles -- and so must this)


class _HostExchange(object):
    """OR of a small host integer over the ranks of ONE node through shared memory -- the reference's batch-global early
    exits need 5 bits per operand, and those bits are ALREADY on the host of every rank when the validation wait ends
    (`_native.flow_flags_host`).  Sending them back to the device, through an RCCL all-reduce and home again cost two copies,
    a collective launch and an event per validation (the sharded route had never been timed: +45 us on a 46 us
    wait); here every rank stores ONE 8-byte word {sequence number, bits} into its slot of a shared segment and reads the
    other ranks' slots until they carry the same sequence number: no launch, no copy, microseconds.

    Like any collective it relies on every rank making the same sequence of calls.  Slots are rings of RING entries: a rank
    can be at most one call ahead of the slowest reader (it cannot finish call k + 1 before everyone has posted k + 1, which a
    rank does only after it has read everybody's k), so entry k % RING is never overwritten while somebody still needs it."""
    RING = 4

    def __init__(self, group):
        from multiprocessing import shared_memory
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.seq = 0
        self.lock = threading.Lock()          # or_reduce from two threads of one rank: `seq` and the slot store are one step
        self.group = group                    # (the timeout is resolved at every exchange: EXCHANGE_TIMEOUT_SECONDS may be set after enable_batch_sharding)
        size = self.world * self.RING * 8
        name = [None]
        self.shm = None
        if self.rank == 0:
            # a failure here (no /dev/shm, no room) must not leave the other ranks in the broadcast below: rank 0 sends
            # name = None and EVERY rank falls back to the communicator route together
            try:
                self.shm = shared_memory.SharedMemory(create=True, size=size)
                np.ndarray((self.world * self.RING,), dtype=np.int64, buffer=self.shm.buf)[:] = 0
                name[0] = self.shm.name
            except Exception:  # noqa: BLE001
                self.shm = None
        dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ok = 1 if name[0] is not None else 0
        if self.rank != 0 and ok:
            try:
                self.shm = shared_memory.SharedMemory(name=name[0])
                # (the segment belongs to rank 0: this process must not unlink it at exit)
                try:
                    from multiprocessing import resource_tracker
                    resource_tracker.unregister(self.shm._name, "shared_memory")
                except Exception:  # noqa: BLE001
                    pass
            except Exception:  # noqa: BLE001        (another node: no such segment here)
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=_collective_device(group))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self.usable = bool(int(flag.cpu()[0]))
        self.slots = None if self.shm is None else np.ndarray((self.world, self.RING), dtype=np.int64, buffer=self.shm.buf)

    def or_reduce(self, bits: int) -> int:
        # Memory-ordering assumption: the word is ONE aligned 8-byte store of a NumPy int64 and the readers'
        # loads are aligned 8-byte loads -- single-copy atomic on x86-64 and AArch64, and the word carries its own sequence
        # number, so no ordering BETWEEN words is relied upon.  Thread safety: the lock makes (seq, slot store, reads) one step
        # per rank -- two threads of one rank take two consecutive sequence numbers; as with any collective, all ranks must
        # then make their calls in the same order.
        with self.lock:
            return self._or_reduce_locked(bits)

    def _or_reduce_locked(self, bits: int) -> int:
        self.seq += 1
        k, e = self.seq, self.seq % self.RING
        self.slots[self.rank, e] = (k << 8) | (bits & 0xff)          # one aligned 8-byte store: sequence number and bits arrive together
        total, t0, looks, timeout = bits & 0xff, None, 0, None
        for r in range(self.world):
            if r == self.rank:
                continue
            while True:
                v = int(self.slots[r, e])
                if (v >> 8) == k:
                    total |= v & 0xff
                    break
                looks += 1
                if looks & 0x3ff == 0 or t0 is not None:
                    # (the first 1 024 looks spin: peers post within microseconds; after that every look checks the clock and, a
                    # millisecond in, sleeps between looks -- a rank waiting for a peer that is compiling or checkpointing must not
                    # burn a core while it holds `lock`)
                    now = time.perf_counter()
                    t0 = now if t0 is None else t0
                    if timeout is None:
                        timeout = _exchange_timeout(self.group)
                    if now - t0 > timeout:
                        raise RuntimeError("oflibpytorch_amd.distributed: rank %d did not post flag exchange %d within %.0f s (ranks "
                                           "must make the same sequence of calls; EXCHANGE_TIMEOUT_SECONDS overrides the process "
                                           "group's timeout)" % (r, k, timeout))
                    if now - t0 > 1e-3:
                        time.sleep(5e-5 if now - t0 < 1.0 else 1e-3)
        return total

    def close(self):
        shm, self.shm, self.slots = self.shm, None, None
        if shm is not None:
            try:
                shm.close()
                if self.rank == 0:
                    shm.unlink()
            except Exception:  # noqa: BLE001
                pass


def _exchange_timeout(group) -> float:
    """Seconds a rank waits for a peer's flag word: EXCHANGE_TIMEOUT_SECONDS when set, else the timeout of the process group's
    own backend (torch's defaults: 10 min for RCCL, 30 min for gloo), else 30 min."""
    if EXCHANGE_TIMEOUT_SECONDS is not None:
        return float(EXCHANGE_TIMEOUT_SECONDS)
    try:
        pg = group if group is not None else dist.group.WORLD
        return float(pg._get_backend(_collective_device(group)).options._timeout.total_seconds())
    except Exception:  # noqa: BLE001
        pass
    try:
        return float(dist.distributed_c10d._get_default_timeout(dist.get_backend(group)).total_seconds())
    except Exception:  # noqa: BLE001
        return 1800.0


def _collective_device(group):
    """Where a tensor must live to go through `group`'s backend (RCCL: the current HIP device; gloo: the host)."""
    try:
        backend = dist.get_backend(group)
    except Exception:  # noqa: BLE001
        backend = "gloo"
    return torch.device('cuda', torch.cuda.current_device()) if backend == "nccl" else torch.device('cpu')


def enable_batch_sharding(group=None):
    """Declare that the flows this process holds are one shard of a batch split over `group`'s ranks.  A COLLECTIVE call
    (every rank of the group makes it): it sets up the host-side flag exchange when all ranks share this node."""
    global _group, _enabled, _exchange
    if not dist.is_initialized():
        raise RuntimeError("oflibpytorch_amd.distributed: torch.distributed is not initialised")
    _group, _enabled = group, True
    _close_exchange()
    if USE_HOST_EXCHANGE:
        try:
            ex = _HostExchange(group)
            _exchange = ex if ex.usable else None
            if not ex.usable:
                ex.close()
        except Exception:  # noqa: BLE001      (no shared memory on this platform: the communicator carries the words)
            _exchange = None


def _close_exchange():
    global _exchange
    if _exchange is not None:
        _exchange.close()
        _exchange = None


atexit.register(_close_exchange)


def disable_batch_sharding():
    global _group, _enabled
    _group, _enabled = None, False
    _close_exchange()


def is_enabled() -> bool:
    return _enabled and dist.is_initialized() and (dist.get_world_size(_group) > 1 or _force_collectives)


def reduce_flags(local_or: int, device) -> int:
    """OR of a 5-bit flag word over all ranks (identity when sharding is off).  Host word in, host word out: one
    collective and one sync; the kernels' device-side words go through `with_global_or` instead."""
    if not is_enabled():
        return local_or
    if _exchange is not None:                # every rank on this node: host words through shared memory, no launch
        return _exchange.or_reduce(int(local_or))
    host = with_global_or(torch.tensor([local_or], dtype=torch.int32, device=device)).cpu().tolist()
    return split_global_or(host)[1]


def host_exchange_active() -> bool:
    return _exchange is not None


def with_global_or(words: torch.Tensor) -> torch.Tensor:
    """int32[N] per-element flag words ON THE DEVICE (a kernel's by-product) -> int32[N + 5]: the same words followed by
    the five bits (0 / 1 each) of their OR over the whole batch and over every rank.  One launch (ofl_flag_words_or_i32)
    and one tiny all-reduce (MAX: NCCL / RCCL has no bitwise OR) -- no host round trip here; the caller reads the N + 5
    integers with a single sync and puts the word together with `split_global_or`."""
    if words.device.type == 'cuda':
        from . import _native
        out = _native.flag_words_or(words)
    else:                                   # host-resident words (the gloo tier of the tests)
        w = words.to(torch.int32).reshape(-1)
        shifts = torch.arange(5, dtype=torch.int32)
        bits = ((w.reshape(-1, 1) >> shifts) & 1).amax(0) if w.numel() else torch.zeros(5, dtype=torch.int32)
        out = torch.cat([w, bits.to(torch.int32)])
    if is_enabled():
        dist.all_reduce(out[-5:], op=dist.ReduceOp.MAX, group=_group)
    return out


def split_global_or(host: list) -> tuple:
    """The host copy of `with_global_or`'s result -> (per-element words, their OR over batch and ranks)"""
    n = len(host) - 5
    return [int(v) for v in host[:n]], sum(int(host[n + k]) << k for k in range(5))


def shard_bounds(n: int, rank: int = None, world: int = None) -> tuple:
    """Contiguous batch chunk [lo, hi) of rank `rank`.  A batch that does not divide is BALANCED: the first n % world ranks hold
    one element more (60 over 8: 8, 8, 8, 8, 7, 7, 7, 7; 2 over 4: 1, 1, 0, 0 -- empty shards are allowed)."""
    rank = dist.get_rank(_group) if rank is None else rank
    world = dist.get_world_size(_group) if world is None else world
    per, extra = divmod(n, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def broadcast_operand(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Distribute a shared (batch-1) image / flow operand from rank `src` to every rank."""
    if dist.is_initialized() and (dist.get_world_size(_group) > 1 or _force_collectives):
        t = t.contiguous()
        dist.broadcast(t, src=src, group=_group)
        _bump_version(t)
    return t


def _bump_version(t: torch.Tensor):
    """c10d's in-place collectives write a tensor WITHOUT touching its version counter (checked: `dist.broadcast` leaves
    `_version` where it was), and a `Flow` caches the flag word of its vectors -- finite? all zero? -- under that counter
    (flow_class.py `_key`): a persistent Flow round a re-used receive buffer would keep the word of the previous contents and take
    (or miss) the reference's early exits (utils.py:497-498, flow_class.py:1729-1744) on stale data.  Every in-place collective
    of this module therefore bumps the counter itself.  (Inference tensors have no counter; their flag words are never cached
    across calls unless the tensor is private to a Flow, which a caller-supplied receive buffer is not.)"""
    try:
        torch.autograd.graph.increment_version(t)
    except Exception:  # noqa: BLE001   (an inference tensor: nothing to bump, nothing cached)
        pass


def all_gather_batch(t: torch.Tensor) -> torch.Tensor:
    """Concatenate the ranks' shards along the batch axis on every rank (not part of the compute metric).  Shards may
    be ragged (`shard_bounds` hands out a shorter, possibly empty, tail): the batch sizes are exchanged first and every
    shard travels padded to the longest."""
    if not (dist.is_initialized() and (dist.get_world_size(_group) > 1 or _force_collectives)):
        return t
    world = dist.get_world_size(_group)
    sizes = torch.zeros(world, dtype=torch.int64, device=t.device)
    sizes[dist.get_rank(_group)] = t.shape[0]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=_group)
    sizes = [int(v) for v in sizes.cpu().tolist()]
    longest = max(sizes)
    if longest == 0:
        return t
    padded = t.contiguous()
    if t.shape[0] != longest:
        padded = torch.zeros((longest,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        padded[:t.shape[0]] = t
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=_group)
    return torch.cat([p[:k] for p, k in zip(parts, sizes)], dim=0)
