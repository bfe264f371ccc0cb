"""Batch sharding over the GPUs of one node: the only parallel axis of the tagging path.

Sequences are independent, so each rank (one process per GPU, torch.distributed over RCCL/xGMI;
"nccl" IS RCCL on ROCm) tags a contiguous slice of the batch with its own replica of the
weights, and ONE collective returns the tag ids: an all-gather of int32 [B/N, L] -- 512 KiB per
GPU at the largest BASELINE config, latency- not bandwidth-bound on the 7 point-to-point xGMI
links.  The reference has no multi-GPU facility at all; this is the MI355X-native
addition of section 8e.  Works unchanged on CPU tensors with the gloo backend (tests).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n, rank, world_size):
    """Contiguous, balanced slice [lo, hi) of n sequences for `rank` (the first n % world_size
    ranks get one more)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x, lengths, rank=None, world_size=None):
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    lo, hi = shard_bounds(x.shape[0], rank, world_size)
    return x[lo:hi], lengths[lo:hi], (lo, hi)


def gather_tags(local_tags, n_total, group=None):
    """All-gather the per-rank tag blocks [n_r, L] into [n_total, L] (same on every rank).
    Ragged shards (n_total % world != 0) are padded to the largest shard for the collective."""
    _, w = world()
    if w == 1:
        return local_tags
    L = local_tags.shape[1]
    sizes = [shard_bounds(n_total, r, w) for r in range(w)]
    biggest = max(hi - lo for lo, hi in sizes)
    buf = local_tags
    if buf.shape[0] < biggest:
        pad = torch.full((biggest - buf.shape[0], L), -1, dtype=buf.dtype, device=buf.device)
        buf = torch.cat([buf, pad], dim=0)
    out = torch.empty((w * biggest, L), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    if all(hi - lo == biggest for lo, hi in sizes):
        return out
    return torch.cat([out[r * biggest: r * biggest + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def tag_sharded(tag_fn, x, lengths, group=None):
    """Tag a replicated batch: every rank runs `tag_fn(x_shard, lengths_shard) -> int32 [n_r, L]`
    on its slice; returns the gathered [B, L] tags on every rank."""
    xs, ls, _ = shard_batch(x, lengths)
    return gather_tags(tag_fn(xs, ls), x.shape[0], group=group)
