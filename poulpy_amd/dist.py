"""Multi-GPU plumbing for the batched path: shard independent ciphertexts across ranks and
broadcast the prepared evaluation key once.  One process per GPU,
torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  No reduction,
no all-to-all: the data path has no cross-GPU dependency.
"""
from __future__ import annotations


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block partition of `total` independent units; the first `total % world`
    ranks take one extra unit."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_key(pmat, src: int = 0, bucket_bytes: int = 64 << 20):
    """Broadcast a prepared key (a flat torch tensor, device or host) from `src` to all ranks
    in buckets of `bucket_bytes` (xGMI links are point-to-point: a few large transfers, not many
    small ones)."""
    import torch.distributed as dist

    flat = pmat.view(-1)
    per = max(1, bucket_bytes // flat.element_size())
    for off in range(0, flat.numel(), per):
        dist.broadcast(flat[off:off + per], src=src)
    return pmat
