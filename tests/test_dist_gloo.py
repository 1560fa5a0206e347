"""world_size-2 `gloo` test of the multi-GPU path (no GPU needed): the prepared key is broadcast
from rank 0, independent ciphertexts are block-sharded, every rank runs its shard (here through the
CPU oracle, standing in for the per-rank device op), and the gathered result equals the
single-process one.  There is no reduction on the data path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.ref import RefModule
    from poulpy_amd import dist as pdist
    from poulpy_amd.layouts import MatZnx, VecZnx, VmpPMat

    n, base2k, cols, size = 64, 12, 2, 3
    ref = RefModule(n)
    key = torch.zeros(n * size * cols * cols * size, dtype=torch.float64)
    if rank == 0:
        mat = MatZnx(n, size, cols, cols, size).fill_uniform(base2k, np.random.default_rng(1))
        pm = ref.vmp_pmat_alloc(size, cols, cols, size)
        ref.vmp_prepare(pm, mat)
        key.copy_(torch.from_numpy(pm.data))
    pdist.broadcast_key(key, src=0, bucket_bytes=4096)  # several buckets
    pm = VmpPMat(n, size, cols, cols, size, key.numpy().copy())
    lo, hi = pdist.shard_range(total, world, rank)
    out = {}
    for idx in range(lo, hi):
        a = VecZnx(n, cols, size).fill_uniform(base2k, np.random.default_rng(1000 + idx))
        res = VecZnx(n, cols, size)
        ref.glwe_external_product(res, base2k, a, base2k, pm, 1, base2k)
        out[idx] = res.data.copy()
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    # the per-rank block of the N > 1 bench line: one small dict from every rank, ordered by rank on every rank
    per_rank = pdist.gather_per_rank({"value": 100.0 + rank, "ms_per_step": 10.0 - rank, "device": f"gpu{rank}", "parity_ok": True})
    assert [e["rank"] for e in per_rank] == list(range(world))
    assert [e["value"] for e in per_rank] == [100.0 + r for r in range(world)]
    if rank == 0:
        merged = {}
        for g in gathered:
            merged.update(g)
        outq.put((merged, per_rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_external_products_match_single_process():
    from oracle.ref import RefModule
    from poulpy_amd.layouts import MatZnx, VecZnx
    total, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged, per_rank = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import json
    from poulpy_amd import dist as pdist
    assert json.loads(json.dumps(per_rank)) == per_rank and len(per_rank) == world      # goes into the JSON line as is
    assert set(per_rank[1]) == {"value", "ms_per_step", "device", "parity_ok", "rank"}
    assert pdist.scaling_efficiency(190.0, 2, 100.0) == pytest.approx(0.95)
    assert pdist.scaling_efficiency(190.0, 2, None) is None and pdist.scaling_efficiency(190.0, 2, 0.0) is None
    n, base2k, cols, size = 64, 12, 2, 3
    ref = RefModule(n)
    mat = MatZnx(n, size, cols, cols, size).fill_uniform(base2k, np.random.default_rng(1))
    pm = ref.vmp_pmat_alloc(size, cols, cols, size)
    ref.vmp_prepare(pm, mat)
    assert sorted(merged) == list(range(total))
    for idx in range(total):
        a = VecZnx(n, cols, size).fill_uniform(base2k, np.random.default_rng(1000 + idx))
        res = VecZnx(n, cols, size)
        ref.glwe_external_product(res, base2k, a, base2k, pm, 1, base2k)
        assert np.array_equal(res.data, merged[idx])


class _FakeLib:
    def __init__(self, owner):
        self.owner = owner

    def pz_comm_world_size(self, handle):
        return self.owner.world if self.owner.comm else 0


class _FakeModule:
    """Stands in for hal.Module's communicator surface (comm_unique_id / comm_init_rank / comm_destroy / bcast_key): the agreement step
    is host logic, the RCCL calls behind it need GPUs."""

    def __init__(self, fail_id=False, fail_init=False):
        self.fail_id, self.fail_init = fail_id, fail_init
        self.comm, self.world, self.handle = False, 0, None
        self.lib = _FakeLib(self)
        self.calls = []

    def comm_available(self):
        self.calls.append("probe")
        if self.fail_id:
            raise RuntimeError("RCCL not found (dlopen librccl.so.1)")

    def comm_unique_id(self):
        self.calls.append("id")
        if self.fail_id:
            raise RuntimeError("RCCL not found (dlopen librccl.so.1)")
        return b"\x01" * 128

    def comm_init_rank(self, world, rank, uid):
        self.calls.append("init")
        assert uid == b"\x01" * 128
        if self.fail_init:
            raise RuntimeError("ncclCommInitRank failed")
        self.comm, self.world = True, world

    def comm_destroy(self):
        self.calls.append("destroy")
        self.comm = False

    def bcast_key(self, ptr, nbytes, root):
        self.calls.append("bcast")   # the payload travels by torch in this stand-in (below)

    def sync(self):
        pass


def _agree_worker(rank, world, port, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from poulpy_amd import dist as pdist
    results = {}
    assert pdist.all_agree(True) is True
    assert pdist.all_agree(rank != 1) is False
    # (a) every rank can use the C ABI -> "cabi" everywhere
    m = _FakeModule()
    key = torch.full((1000,), float(rank == 0))
    results["ok"] = (pdist.broadcast_key_agreed(m, key, src=0, route="auto"), list(m.calls))
    # (b) rank 1 cannot load RCCL -> every rank takes torch, nobody initialises a communicator, the key still arrives
    m = _FakeModule(fail_id=(rank == 1))
    key = torch.full((1000,), 7.0 if rank == 0 else 0.0)
    route = pdist.broadcast_key_agreed(m, key, src=0, route="auto")
    results["no_rccl_on_1"] = (route, list(m.calls), float(key.sum()))
    # (c) ncclCommInitRank fails on rank 1 only -> rank 0 destroys its communicator, both take torch
    m = _FakeModule(fail_init=(rank == 1))
    key = torch.full((10,), 3.0 if rank == 0 else 0.0)
    route = pdist.broadcast_key_agreed(m, key, src=0, route="auto")
    results["init_fails_on_1"] = (route, list(m.calls), float(key.sum()), m.comm)
    # (d) --bcast cabi: an error on EVERY rank, not a hang
    m = _FakeModule(fail_id=(rank == 1))
    try:
        pdist.broadcast_key_agreed(m, torch.zeros(4), src=0, route="cabi")
        results["cabi_strict"] = "no error"
    except RuntimeError as e:
        results["cabi_strict"] = "raised" + (" here" if "here:" in str(e) else "")
    # (e) ADVICE r05: src already has a communicator, rank 1 has none -> src destroys its own, draws a FRESH id, both initialise, route "cabi"
    m = _FakeModule()
    if rank == 0:
        m.comm, m.world = True, world
    key = torch.full((10,), 5.0 if rank == 0 else 0.0)
    route = pdist.broadcast_key_agreed(m, key, src=0, route="cabi")
    results["src_has_comm"] = (route, list(m.calls), m.comm)
    outq.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_route_is_agreed_before_any_collective():
    """bench.py --bcast auto: a rank whose C-ABI communicator fails must not fall back alone while the others sit
    in ncclBroadcast; the route is settled by all_reduce(MIN) after each step that may fail locally."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r]["ok"] == ("cabi", ["id" if r == 0 else "probe", "init", "bcast"])   # only src draws an id (ncclGetUniqueId opens a listener)
        route, calls, total = got[r]["no_rccl_on_1"]
        assert route == "torch" and "init" not in calls and "bcast" not in calls and total == 7000.0
        route, calls, total, comm = got[r]["init_fails_on_1"]
        assert route == "torch" and "bcast" not in calls and total == 30.0 and comm is False
    assert got[0]["init_fails_on_1"][1] == ["id", "init", "destroy"]
    assert got[0]["src_has_comm"] == ("cabi", ["destroy", "id", "init", "bcast"], True)
    assert got[1]["src_has_comm"] == ("cabi", ["probe", "init", "bcast"], True)
    assert got[0]["cabi_strict"] == "raised" and got[1]["cabi_strict"] == "raised here"
