"""One rank per GPU for every bench of this repo (bench.py and tools/bench_*.py): the launcher, the refusals, the key broadcast, the
max-over-ranks clock and the per-rank block - the logic bench.py carried alone until round 6.

    R = multirank.enter(__file__, args.gpus, sys.argv[1:])   # started plainly with --gpus N > 1: becomes the launcher (child process per GPU) and exits
    torch, dist = R.init()                                   # device checks, torch.cuda.set_device, process group "nccl" (= RCCL) when N > 1
    ... rank 0 prepares the keys, every rank allocates them ...
    route = R.broadcast_keys(mod, [brk, *atk, *tsk], args.bcast)   # one route on every rank (poulpy_amd.dist.broadcast_key_agreed)
    lo, hi = R.shard(args.batch * R.world)                   # weak scaling: `batch` units per GPU, global indices
    R.sync_all(); t0 = ...; timed calls; mod.sync(); R.sync_all(); dt = R.max_seconds(dt)
    per_rank = R.gather(mine)                                # every rank's own figures, ordered by rank
    if R.rank == 0: print(json.dumps({..., **R.line_fields(value, ref_value, route)}))
    R.finish()

Nothing at module level imports torch or touches HIP: the launching parent must stay GPU-free (a process that has initialised the GPU must not
start another program in its place on this pool; children are started, never exec'd)."""
from __future__ import annotations

import os
import sys


def launcher_argv(script: str, gpus: int, argv: list, port: int) -> list:
    """The command `<bench> --gpus N` (N > 1, no WORLD_SIZE) runs as a child: one rank per GPU on this node, rendezvous on 127.0.0.1
    (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(script), *argv]


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(script: str, gpus: int, argv: list) -> int:
    """Parent side of `--gpus N`: nothing here imports torch or initialises HIP (the child processes own the GPUs)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver: RCCL needs it
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = launcher_argv(script, gpus, argv, int(os.environ.get("POULPY_BENCH_PORT", "0")) or free_port())
    return subprocess.run(cmd, env=env).returncode


class Ranks:
    """This process's place in the job.  world == 1 without POULPY_BENCH_FORCE_DIST: no process group at all (the single-GPU line is what it
    always was)."""

    def __init__(self, name: str, gpus: int):
        self.name = name
        self.gpus = gpus
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.distributed = self.world > 1 or os.environ.get("POULPY_BENCH_FORCE_DIST") == "1"   # (the knob runs the RCCL path with one rank)
        self.torch = self.dist = self.dev = None

    def init(self):
        """Device checks in the rank process (the launching parent never touches torch or HIP): fewer visible devices than local ranks is a
        clean one-line refusal - no JSON line, exit code 4 on every rank."""
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise SystemExit(f"{self.name} needs a HIP device (there is no CPU fallback)")
        ndev = torch.cuda.device_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))
        if ndev < local_world or self.local_rank >= ndev:
            if self.rank == 0:
                print(f"{self.name}: --gpus {self.gpus} needs {local_world} HIP devices on this node but only {ndev} are visible: no line reported",
                      file=sys.stderr, flush=True)
            raise SystemExit(4)
        torch.cuda.set_device(self.local_rank)
        self.torch, self.dist, self.dev = torch, dist, torch.device("cuda", self.local_rank)
        if self.distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=self.dev)
        return torch, dist

    def attach(self, torch, dist, dev):
        """Adopt an already initialised process group (the world-size-2 gloo tests: host tensors, no GPU)."""
        self.torch, self.dist, self.dev = torch, dist, dev
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.distributed = True
        return self

    def shard(self, total: int):
        from poulpy_amd import dist as pdist
        return pdist.shard_range(total, self.world, self.rank)

    def broadcast_keys(self, mod, tensors, route: str = "auto", log=None):
        """Every prepared key of the job (flat device tensors, allocated on every rank, filled on rank 0) by ONE route on every rank; returns
        the route's name, None on a single rank.  The first tensor settles the route (C-ABI communicator or torch.distributed), the others follow."""
        if not self.distributed:
            return None
        from poulpy_amd import dist as pdist
        used = None
        for t in tensors:
            used = pdist.broadcast_key_agreed(mod, t, src=0, route=(route if used is None else used), log=log)
        return used

    def sync_all(self):
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_seconds(self, dt: float) -> float:
        if not self.distributed:
            return dt
        t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_true(self, ok) -> bool:
        """AND of a per-rank check (None counts as true: a rank without a checker does not veto)."""
        if not self.distributed:
            return ok is not False
        v = self.torch.tensor([0 if ok is False else 1], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MIN)
        return bool(v.item())

    def gather(self, mine: dict) -> list:
        if not self.distributed:
            return [dict(mine, rank=self.rank)]
        from poulpy_amd import dist as pdist
        return pdist.gather_per_rank(mine)

    def line_fields(self, value: float, ref_value: float, route, mod=None) -> dict:
        """The fields every N-rank line carries beside its own: n_gpus, scaling, parallelism, rccl_ranks, scaling_efficiency."""
        from poulpy_amd import dist as pdist
        rccl = 0
        if self.distributed:
            rccl = int(mod.lib.pz_comm_world_size(mod.handle)) if (route == "cabi" and mod is not None) else self.dist.get_world_size()
        return {"n_gpus": self.world, "scaling": "weak",
                "parallelism": (f"batch-sharded x{self.world}, keys broadcast once over RCCL ({route})" if self.distributed else "single GPU (no collective)"),
                "rccl_ranks": rccl, "scaling_efficiency": pdist.scaling_efficiency(value, self.world, ref_value)}

    def finish(self):
        if self.distributed:
            self.dist.destroy_process_group()


def enter(script: str, gpus: int, argv: list) -> Ranks:
    """Started plainly with --gpus N > 1: become the launcher (a child process per GPU; never exec from here, never touch the GPU here) and exit with
    its code.  Under torch.distributed.run: refuse a WORLD_SIZE other than --gpus (a line never reports a GPU count other than the one asked for)."""
    name = os.path.basename(script)
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and gpus > 1:
        raise SystemExit(self_launch(script, gpus, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != gpus:
        raise SystemExit(f"{name}: --gpus {gpus} but WORLD_SIZE={world}: refusing to report a line for a different GPU count "
                         f"(launch with --nproc-per-node {gpus}, or plainly and let --gpus start the ranks)")
    return Ranks(name, gpus)


def add_arguments(ap):
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU of this node); started plainly the tool launches them itself")
    ap.add_argument("--bcast", choices=("auto", "cabi", "torch"), default=os.environ.get("POULPY_BENCH_BCAST", "auto"),
                    help="how the prepared keys reach the other ranks: cabi = pz_bcast_key (RCCL inside the C ABI), torch = torch.distributed.broadcast; "
                         "auto = cabi when every rank can, torch on every rank otherwise")
    ap.add_argument("--ref-value", type=float, default=0.0, help="the 1-GPU value: the line then carries scaling_efficiency = value / (N x ref)")
