"""bench.py's multi-GPU contract: `--gpus N` started plainly launches the ranks itself (child process,
before torch / HIP are touched); under torch.distributed.run it refuses to report a line for another GPU count than asked."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_argv_one_rank_per_gpu_on_localhost():
    b = _load_bench()
    argv = b.launcher_argv(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29511)
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29511"
    i = argv.index(BENCH)
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]          # the ranks see the same arguments
    assert isinstance(b.free_port(), int) and b.free_port() > 1024


def test_parent_does_not_import_torch_before_launching():
    """The launcher path must not initialise anything: the module-level imports of bench.py are torch-free."""
    code = "import sys, importlib.util as u; s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); s.loader.exec_module(m); " \
           "print('torch' in sys.modules)" % BENCH
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.strip() == "False", out.stderr


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "WORLD_SIZE=2" in out.stderr and "--gpus 1" in out.stderr
    assert not any(l.startswith("{") for l in out.stdout.splitlines())                # never a JSON line for the wrong count
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_physical_core_selection():
    b = _load_bench()
    cores = b.physical_core_cpus()
    assert cores and len(set(cores)) == len(cores)
    assert set(cores) <= set(os.sched_getaffinity(0))
    assert isinstance(b.cpu_model(), str)


@pytest.mark.gpu
def test_single_rank_rccl_path_through_the_c_abi():
    """`bench.py --gpus 1` with the distributed path forced: process group, pz_comm_init_rank, pz_bcast_key (RCCL inside the C ABI),
    global-index shard, per-rank parity sample — everything the N-rank run does, with one rank."""
    env = dict(os.environ, POULPY_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64", "--setup-calls", "1", "--bcast", "cabi",
                          "--no-cpu-baseline", "--parity-samples", "2", "--timing-steps", "1"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1
    assert "cabi" in line["config"]["parallelism"]
    assert line["parity_sample"]["ok"] is True


@pytest.mark.gpu
def test_fewer_devices_than_ranks_is_a_one_line_refusal():
    """A rank that finds fewer HIP devices than WORLD_SIZE exits 4 with one line on rank 0 and no JSON:
    run here as rank 0 of a pretended 64-rank job, more than any node has."""
    env = dict(os.environ, WORLD_SIZE="64", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert out.returncode == 4, (out.returncode, out.stderr[-500:])
    mine = [l for l in out.stderr.splitlines() if l.startswith("bench.py:")]
    assert len(mine) == 1 and "needs 64 HIP devices" in mine[0] and "no line reported" in mine[0]
    assert not any(l.startswith("{") for l in out.stdout.splitlines())


def test_roofline_traffic_is_keyed_on_the_kernel_signature(tmp_path, monkeypatch):
    """bench.py quotes the committed PMC traffic only while the kernel in the library that runs NOW carries the code-object signature the
    profile was taken on (tools/traffic_json.py::kernel_signatures); a profile of another build yields traffic = null instead of a stale number."""
    import glob as _glob
    import shutil
    b = _load_bench()
    if not os.path.exists(os.path.join(ROOT, "poulpy_amd", "libpoulpy_hip.so")) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("needs the built library and llvm-readelf")
    files = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    doc = json.load(open(files[-1]))
    assert doc.get("kernel_signatures") and "error" not in doc["kernel_signatures"], "the newest profiles/r*_traffic.json must carry kernel signatures"
    live = b.pmc_traffic("fused_mid", int(doc["batch_per_launch"]))
    assert live is not None and 1.6e10 < live < 1.9e10, live          # ~17.3 GB per 1024 products: one HBM round trip of the middle kernel
    # the same profile with another build's signature: not this run's traffic
    stale = dict(doc, kernel_signatures={k: "1/2/3/4/5" for k in doc["kernel_signatures"]})
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99_traffic.json").write_text(json.dumps(stale))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    shutil.copytree(os.path.join(ROOT, "tools"), str(tmp_path / "tools"))
    assert b.pmc_traffic("fused_mid", int(doc["batch_per_launch"])) is None
