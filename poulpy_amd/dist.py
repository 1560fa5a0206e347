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


def broadcast_key(pmat, src: int = 0, bucket_bytes: int = 0):
    """Broadcast a prepared key (a flat torch tensor, device or host) from `src` to all ranks
    in buckets of `bucket_bytes` (default 64 MiB; POULPY_BCAST_BUCKET_BYTES overrides it - the CPU tests use small buckets; xGMI links are
    point-to-point: a few large transfers, not many small ones)."""
    import os

    import torch.distributed as dist

    bucket_bytes = bucket_bytes or int(os.environ.get("POULPY_BCAST_BUCKET_BYTES", str(64 << 20)))

    flat = pmat.view(-1)
    per = max(1, bucket_bytes // flat.element_size())
    for off in range(0, flat.numel(), per):
        dist.broadcast(flat[off:off + per], src=src)
    return pmat


def broadcast_key_cabi(mod, pmat, src: int = 0):
    """The same broadcast through the C ABI (`pz_bcast_key`: ncclBroadcast on the module's own stream), i.e. the route a Rust /
    C++ caller takes.  torch.distributed is only the out-of-band channel for the 128-byte RCCL id (any process group works,
    gloo included).  The communicator is created on first use and owned by the module."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if mod.lib.pz_comm_world_size(mod.handle) == 0:
        obj = [mod.comm_unique_id() if rank == src else None]
        dist.broadcast_object_list(obj, src=src)
        mod.comm_init_rank(world, rank, obj[0])
    flat = pmat.view(-1)
    if flat.is_cuda:
        torch.cuda.current_stream(flat.device).synchronize()   # the module stream is not ordered with torch's streams
    mod.bcast_key(C.c_void_p(flat.data_ptr()), flat.numel() * flat.element_size(), src)
    mod.sync()
    return pmat


def all_agree(ok: bool) -> bool:
    """True iff `ok` on EVERY rank: one all_reduce(MIN) of a flag on the default process group (device tensor under nccl / RCCL, host
    tensor under gloo).  The step that keeps ranks from taking different routes into a collective."""
    import torch
    import torch.distributed as dist

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def broadcast_key_agreed(mod, pmat, src: int = 0, route: str = "auto", log=None) -> str:
    """Broadcast the prepared key by ONE route on every rank and return its name ("cabi" | "torch").

    route "cabi": the C-ABI communicator or an exception on every rank; "torch": torch.distributed only; "auto": the C ABI when EVERY rank
    can use it, torch.distributed on every rank otherwise.  No rank falls back alone: each local step that may fail (loading RCCL / drawing
    an id, ncclCommInitRank) is followed by `all_agree`, so a rank whose librccl is missing cannot leave the others inside ncclBroadcast."""
    import torch.distributed as dist

    if route not in ("auto", "cabi", "torch"):
        raise ValueError(f"unknown broadcast route {route!r}")
    if route == "torch":
        broadcast_key(pmat, src=src)
        return "torch"
    world, rank = dist.get_world_size(), dist.get_rank()
    err = None
    have_comm = mod.lib.pz_comm_world_size(mod.handle) != 0
    # step 1 (local): RCCL loads here; only src draws an id (ncclGetUniqueId opens a bootstrap listener: the other ranks probe with
    # pz_comm_available, which loads the library and nothing else)
    my_id = None
    if not have_comm:
        try:
            if rank == src:
                my_id = mod.comm_unique_id()
            else:
                mod.comm_available()
        except Exception as e:   # noqa: BLE001 - any failure means "not by this route"
            err = e
    ok = all_agree(err is None)
    # step 2 (collective, entered by all ranks or by none): the communicator
    if ok and not all_agree(have_comm):
        if have_comm:   # a communicator on some ranks only: start over on all of them
            mod.comm_destroy()
        if rank == src and my_id is None:   # src had a communicator (so it drew no id in step 1) and another rank did not
            try:
                my_id = mod.comm_unique_id()
            except Exception as e:   # noqa: BLE001
                err = e
        ok = all_agree(err is None)   # nobody enters ncclCommInitRank with an id that was never drawn
        if ok:
            obj = [my_id if rank == src else None]
            dist.broadcast_object_list(obj, src=src)
            try:
                mod.comm_init_rank(world, rank, obj[0])
            except Exception as e:   # noqa: BLE001
                err = e
            ok = all_agree(err is None)
            if not ok and err is None:
                mod.comm_destroy()
    if not ok:
        if route == "cabi":
            raise RuntimeError(f"rank {rank}: the C-ABI communicator is not available on every rank" + (f" (here: {err})" if err else ""))
        if log and err is not None:
            log(f"rank {rank}: pz_comm_* failed ({err}); every rank uses torch.distributed.broadcast")
        broadcast_key(pmat, src=src)
        return "torch"
    broadcast_key_cabi(mod, pmat, src=src)
    return "cabi"


def gather_per_rank(entry: dict) -> list:
    """Every rank contributes one small dict (its own rate, step time, dominant-kernel time, achieved GB/s); every rank gets the list
    ordered by rank.  One all_gather_object, outside the timed region - the N > 1 bench line reports per-GPU figures beside the
    max-over-ranks headline (north_star: "achieved HBM GB/s at 1, 2, 4 and 8 GPUs")."""
    import torch.distributed as dist

    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, dict(entry, rank=dist.get_rank()))
    return sorted(out, key=lambda e: e["rank"])


def scaling_efficiency(value: float, world: int, ref_value: float):
    """Weak scaling: whole-job value / (world x the 1-GPU value).  None without a reference."""
    if not ref_value or ref_value <= 0 or world < 1:
        return None
    return value / (world * ref_value)
