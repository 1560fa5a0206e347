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
