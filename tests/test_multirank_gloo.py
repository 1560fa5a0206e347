"""world-size-2 `gloo` tests of the multi-rank mode of tools/bench_blind_rotation.py, bench_circuit_bootstrapping.py and bench_tensor.py
(BASELINE configs[3] / [4] are worded "sharded over 8 MI355X"): each tool's key set - the blind-rotation key as ONE tensor
of n_lwe prepared GGSWs, the automorphism and tensor keys - prepared on rank 0 and broadcast through tools/multirank.py in several buckets, the
shard bounds, the per-rank block, the max-over-ranks clock and the AND of the per-rank parity flags.  No GPU: host tensors, a stand-in for the
module's communicator surface (tests/test_dist_gloo.py::_FakeModule)."""
import importlib.util
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def key_sets():
    """(tool, [(key name, elements)]) at the tools' own shapes, n_lwe cut to 5 so that the test stays small: the LAYOUT is what is checked."""
    br = _tool("bench_blind_rotation").SHAPES["cbt"]
    cols = br["rank"] + 1
    pm = br["n"] * br["dnum"] * cols * cols * br["brk_size"]
    cb = _tool("bench_circuit_bootstrapping").SHAPE
    ccols = cb["rank"] + 1
    log_n = cb["n"].bit_length() - 1
    cpm = cb["n"] * cb["brk_dnum"] * ccols * ccols * cb["glwe_size"]
    return {
        "br": [("brk", 5 * pm), ("ksk", br["n"] * br["res_size"] * 1 * cols * br["res_size"])],
        "cbt": [("brk", 5 * cpm)] + [(f"atk{i}", cb["n"] * cb["atk_dnum"] * cb["rank"] * ccols * cb["atk_size"]) for i in range(log_n)] +
               [(f"tsk{i}", cb["n"] * cb["tsk_dnum"] * cb["rank"] * ccols * cb["tsk_size"]) for i in range(cb["rank"])],
        "tensor": [("tensor_key", 4096 * 4 * 1 * 2 * 4)],
    }


def _worker(rank, world, port, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["POULPY_BCAST_BUCKET_BYTES"] = str(64 << 10)   # every key of the sets below spans several buckets
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import multirank
    from tests.test_dist_gloo import _FakeModule
    R = multirank.Ranks("tool.py", world).attach(torch, dist, torch.device("cpu"))
    results = {}
    for tool, keys in key_sets().items():
        mod = _FakeModule(fail_id=True)   # no RCCL in this stand-in: route "auto" must settle on torch.distributed on every rank, for every key
        tensors = [torch.arange(numel, dtype=torch.float64) * (i + 1) if rank == 0 else torch.zeros(numel, dtype=torch.float64) for i, (_, numel) in enumerate(keys)]
        route = R.broadcast_keys(mod, tensors, "auto")
        ok = all(bool(torch.equal(t, torch.arange(numel, dtype=torch.float64) * (i + 1))) for i, (t, (_, numel)) in enumerate(zip(tensors, keys)))
        buckets = [-(-numel * 8 // (64 << 10)) for _, numel in keys]
        lo, hi = R.shard(1024 * world)
        per = R.gather({"value": 1000.0 + rank, "global_first_index": lo, "parity_ok": rank == 0 or tool != "cbt", "rounding_margin": 1e-3 * (rank + 1), "device": rank})
        results[tool] = {"route": route, "keys_ok": ok, "min_buckets": min(buckets), "shard": (lo, hi), "per_rank": per,
                         "max_dt": R.max_seconds(0.5 + rank), "and": R.all_true(rank == 0 or tool != "cbt"),
                         "fields": R.line_fields(2000.0, 1100.0, route, mod), "calls": list(mod.calls)}
    outq.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_rank_mode_of_the_secondary_benches():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for tool in ("br", "cbt", "tensor"):
            g = got[r][tool]
            assert g["route"] == "torch" and g["keys_ok"], (r, tool)
            assert g["min_buckets"] >= 2, "every key must span several buckets in this test"
            assert g["shard"] == (1024 * r, 1024 * (r + 1))                                   # weak scaling: 1024 per GPU, global indices
            assert [e["rank"] for e in g["per_rank"]] == [0, 1] and [e["global_first_index"] for e in g["per_rank"]] == [0, 1024]
            assert g["max_dt"] == 1.5                                                         # the slowest rank's clock
            assert g["and"] is (tool != "cbt")                                                # one rank's failed parity fails the line
            f = g["fields"]
            assert f["n_gpus"] == 2 and f["scaling"] == "weak" and "torch" in f["parallelism"] and f["rccl_ranks"] == 2
            assert f["scaling_efficiency"] == pytest.approx(2000.0 / (2 * 1100.0))
            assert "init" not in g["calls"] and "bcast" not in g["calls"]                     # nobody entered a C-ABI collective alone


def test_tools_refuse_a_world_size_other_than_gpus():
    """As bench.py: under torch.distributed.run a tool never reports a line for another GPU count than --gpus (no GPU is touched before the check)."""
    import subprocess
    for tool in ("bench_blind_rotation.py", "bench_circuit_bootstrapping.py", "bench_tensor.py"):
        env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--gpus", "1"], capture_output=True, text=True, env=env, cwd=ROOT)
        assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and "--gpus 1" in out.stderr, (tool, out.stderr[-300:])
        assert not any(l.startswith("{") for l in out.stdout.splitlines())


def test_launcher_command_of_a_tool():
    import multirank
    script = os.path.join(ROOT, "tools", "bench_tensor.py")
    argv = multirank.launcher_argv(script, 8, ["--gpus", "8", "--relin"], 29555)
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in argv and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index(script) + 1:] == ["--gpus", "8", "--relin"]
