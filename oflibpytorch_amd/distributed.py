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
import torch
import torch.distributed as dist

_group = None
_enabled = False
_force_collectives = False   # tests only: run the collectives on a communicator of ONE rank too (tests/test_gpu_validation.py)


def enable_batch_sharding(group=None):
    """Declare that the flows this process holds are one shard of a batch split over `group`'s ranks."""
    global _group, _enabled
    if not dist.is_initialized():
        raise RuntimeError("oflibpytorch_amd.distributed: torch.distributed is not initialised")
    _group, _enabled = group, True


def disable_batch_sharding():
    global _group, _enabled
    _group, _enabled = None, False


def is_enabled() -> bool:
    return _enabled and dist.is_initialized() and (dist.get_world_size(_group) > 1 or _force_collectives)


def reduce_flags(local_or: int, device) -> int:
    """OR of a 5-bit flag word over all ranks (identity when sharding is off).  Host word in, host word out: one
    collective and one sync; the kernels' device-side words go through `with_global_or` instead."""
    if not is_enabled():
        return local_or
    host = with_global_or(torch.tensor([local_or], dtype=torch.int32, device=device)).cpu().tolist()
    return split_global_or(host)[1]


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
    """Contiguous batch chunk [lo, hi) of rank `rank` (ragged tail allowed)."""
    rank = dist.get_rank(_group) if rank is None else rank
    world = dist.get_world_size(_group) if world is None else world
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def broadcast_operand(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Distribute a shared (batch-1) image / flow operand from rank `src` to every rank."""
    if dist.is_initialized() and (dist.get_world_size(_group) > 1 or _force_collectives):
        t = t.contiguous()
        dist.broadcast(t, src=src, group=_group)
    return t


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
