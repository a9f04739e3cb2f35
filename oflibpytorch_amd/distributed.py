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
    return _enabled and dist.is_initialized() and dist.get_world_size(_group) > 1


def reduce_flags(local_or: int, device) -> int:
    """OR of a 5-bit flag word over all ranks (identity when sharding is off)."""
    if not is_enabled():
        return local_or
    bits = torch.tensor([(local_or >> b) & 1 for b in range(5)], dtype=torch.int32, device=device)
    dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=_group)
    return sum(int(v) << b for b, v in enumerate(bits.cpu().tolist()))


def shard_bounds(n: int, rank: int = None, world: int = None) -> tuple:
    """Contiguous batch chunk [lo, hi) of rank `rank` (ragged tail allowed)."""
    rank = dist.get_rank(_group) if rank is None else rank
    world = dist.get_world_size(_group) if world is None else world
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def broadcast_operand(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Distribute a shared (batch-1) image / flow operand from rank `src` to every rank."""
    if dist.is_initialized() and dist.get_world_size(_group) > 1:
        t = t.contiguous()
        dist.broadcast(t, src=src, group=_group)
    return t


def all_gather_batch(t: torch.Tensor) -> torch.Tensor:
    """Concatenate equally-sized shards along the batch axis on every rank (not part of the compute metric)."""
    if not (dist.is_initialized() and dist.get_world_size(_group) > 1):
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(_group))]
    dist.all_gather(parts, t.contiguous(), group=_group)
    return torch.cat(parts, dim=0)
