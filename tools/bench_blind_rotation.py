#!/usr/bin/env python3
"""Secondary measurement (BASELINE configs[3]): CGGI blind rotations / s on one MI355X for a batch of LWE ciphertexts.

Shapes: `ref` = poulpy-bench/src/bench_suite/schemes/blind_rotation.rs:39-75 (n_glwe 512, n_lwe 687, rank 3, block 3,
base2k 18, dnum 1, key size 2, res size 1); `cbt` = the blind-rotation layout of the circuit-bootstrapping bench
(circuit_bootstrapping.rs:48-55: n_glwe 1024, n_lwe 574, base2k 13, dnum 3, rank 1), block size 7.
Synthetic key material (uniform digits); the oracle is timed single-threaded on a few ciphertexts beside it.

    python tools/bench_blind_rotation.py [--shape ref|cbt] [--batch 1024] [--reps 3] [--gpus N]

--gpus N (round 6; BASELINE configs[3]: "batch = 8192 sharded over 8 MI355X" = 1024 per GPU): one rank per GPU (tools/multirank.py), `--batch`
LWE ciphertexts per GPU (weak scaling), the blind-rotation key - n_lwe prepared GGSWs - and the key-switching key prepared on rank 0 and
broadcast once (poulpy_amd.dist.broadcast_key_agreed: pz_bcast_key = RCCL inside the C ABI, or torch.distributed); no other collective.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SHAPES = {
    "ref": dict(n=512, n_lwe=687, rank=3, block_size=3, base2k=18, dnum=1, brk_size=2, res_size=1),
    "cbt": dict(n=1024, n_lwe=574, rank=1, block_size=7, base2k=13, dnum=3, brk_size=3, res_size=3),
    # the large ring degree of BASELINE configs[3] ("N = 2^10 / 2^14"): same key layout at N = 2^14 (composed path: the
    # accumulators do not fit in LDS)
    "big": dict(n=16384, n_lwe=574, rank=1, block_size=7, base2k=13, dnum=3, brk_size=3, res_size=3),
    "n2048": dict(n=2048, n_lwe=574, rank=1, block_size=7, base2k=13, dnum=3, brk_size=3, res_size=3),
    "n4096": dict(n=4096, n_lwe=574, rank=1, block_size=7, base2k=13, dnum=3, brk_size=3, res_size=3),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", choices=sorted(SHAPES), default="ref")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--block-size", type=int, default=0, help="override the block size (1 = execute_standard)")
    ap.add_argument("--base2k", type=int, default=0, help="override base2k (the exactness margin shrinks with N * 2^(2 base2k))")
    ap.add_argument("--n-lwe", type=int, default=0)
    ap.add_argument("--cpu-cts", type=int, default=2)
    ap.add_argument("--with-keyswitch", action="store_true",
                    help="also time blind rotation + GLWE key switch of the result (the two heavy steps of a gate bootstrap, "
                         "BASELINE configs[3]); rank-1 shapes only")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import multirank
    multirank.add_arguments(ap)
    args = ap.parse_args()
    R = multirank.enter(__file__, args.gpus, sys.argv[1:])   # --gpus N > 1 started plainly: launches the ranks and exits with their code
    torch, dist = R.init()
    from poulpy_amd.hal import BlindRotationParams, GlweOpParams, Module
    s = dict(SHAPES[args.shape])
    if args.block_size:
        s["block_size"] = args.block_size
    if args.base2k:
        s["base2k"] = args.base2k
    if args.n_lwe:
        s["n_lwe"] = args.n_lwe
    n, cols = s["n"], s["rank"] + 1
    dev = R.dev
    mod = Module(n, device=R.local_rank)
    half = 1 << (s["base2k"] - 1)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    pm_elems = n * s["dnum"] * cols * cols * s["brk_size"]
    brk = torch.empty((s["n_lwe"], pm_elems), dtype=torch.float64, device=dev)
    mat = torch.randint(-half, half, (pm_elems,), dtype=torch.int64, device=dev, generator=g)   # (same seed on every rank: the lookup table and the key digits of the CPU check)
    for i in range(s["n_lwe"] if R.rank == 0 else 0):   # rank 0 prepares the key; same synthetic GGSW for every coefficient would let caches lie: permute it per key
        mi = torch.roll(mat, i * 977)
        torch.cuda.synchronize()   # torch's stream and the module's stream are not ordered: the roll must have finished
        mod._ck(mod.lib.pz_vmp_prepare(mod.handle, C.c_void_p(brk[i].data_ptr()), C.c_void_p(mi.data_ptr()), C.c_size_t(s["dnum"]),
                                       C.c_size_t(cols), C.c_size_t(cols), C.c_size_t(s["brk_size"])))
        mod.sync()   # device-pointer calls are stream-ordered on the module's stream: `mi` must outlive the kernels that read it
    mod.sync()
    # the only collective: the whole blind-rotation key (n_lwe prepared GGSWs, several 64 MiB buckets) from rank 0 to every rank
    route = R.broadcast_keys(mod, [brk], args.bcast, log=lambda m: print(f"[bench_blind_rotation] {m}", file=sys.stderr, flush=True))
    lut = torch.randint(-half, half, (s["res_size"], 1, n), dtype=torch.int64, device=dev, generator=g)
    lo, hi = R.shard(args.batch * R.world)   # this rank's block of the global batch (weak scaling: args.batch per GPU)
    g.manual_seed(0xB007 + lo)               # the LWE ciphertexts of a rank are drawn from its first global index
    lwe = torch.randint(-n, n, (args.batch, s["n_lwe"] + 1), dtype=torch.int64, device=dev, generator=g)
    res = torch.empty((args.batch, s["res_size"], cols, n), dtype=torch.int64, device=dev)
    p = BlindRotationParams(rank=s["rank"], n_lwe=s["n_lwe"], block_size=s["block_size"], dnum=s["dnum"], brk_size=s["brk_size"],
                            base2k=s["base2k"], res_size=s["res_size"], lut_size=s["res_size"])
    ptr = lambda t: C.c_void_p(t.data_ptr())
    torch.cuda.synchronize()
    mod.dispatch_notes(reset=True)
    mod.blind_rotation_execute_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), p, args.batch)   # warm-up
    mod.sync()
    # at least ~0.3 s of timed work (a 7 ms call timed over 3 repetitions ran on clocks still ramping: the gate-bootstrap line, timed later in the same
    # process, came out FASTER than the rotation inside it)
    t0 = time.perf_counter()
    mod.blind_rotation_execute_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), p, args.batch)
    mod.sync()
    one = max(time.perf_counter() - t0, 1e-4)
    reps = max(args.reps, min(100, int(0.3 / one) + 1))
    R.sync_all()
    t0 = time.perf_counter()
    for _ in range(reps):
        mod.blind_rotation_execute_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), p, args.batch)
    mod.sync()
    R.sync_all()
    dt_mine = (time.perf_counter() - t0) / reps
    dt = R.max_seconds(dt_mine)   # the slowest rank's clock around the same barrier-bracketed region
    mod.set_kernel_timing(True)   # one more pass with per-class HIP-event timing (adds event overhead: not the timed run)
    mod.blind_rotation_execute_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), p, args.batch)
    mod.sync()
    kstats = {k: (v[0], round(v[1], 3)) for k, v in mod.kernel_stats().items() if v[0]}
    mod.set_kernel_timing(False)
    margin = mod.rounding_margin_of(lambda: mod.blind_rotation_execute_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), p, args.batch))
    ks_stats = None
    if args.with_keyswitch and s["rank"] == 1:
        # key-switching key: GGLWE rows = dnum, cols_in = 1, cols_out = 2, same size / base2k as the accumulator
        ksz = s["res_size"]
        g.manual_seed(0x4B53)   # the key digits: the same on every rank (rank 0 prepares and broadcasts, every rank's CPU check re-prepares them)
        kmat = torch.randint(-half, half, (n * ksz * 1 * cols * ksz,), dtype=torch.int64, device=dev, generator=g)
        g.manual_seed(0xB107 + lo)
        kpm = torch.empty(kmat.numel(), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        if R.rank == 0:
            mod._ck(mod.lib.pz_vmp_prepare(mod.handle, C.c_void_p(kpm.data_ptr()), C.c_void_p(kmat.data_ptr()), C.c_size_t(ksz), C.c_size_t(1),
                                           C.c_size_t(cols), C.c_size_t(ksz)))
            mod.sync()
        R.broadcast_keys(mod, [kpm], route or args.bcast)
        kp = GlweOpParams(rank=1, dnum=ksz, dsize=1, key_size=ksz, key_base2k=s["base2k"], a_size=ksz, a_base2k=s["base2k"],
                          res_size=ksz, res_base2k=s["base2k"], rank_out=1)
        # the whole gate bootstrap on the device (BASELINE configs[3]): 2-limb LWE -> mod_switch_2n -> blind rotation ->
        # lwe_from_glwe (key switch + sample extract) back to an LWE of the input dimension
        lwe_in = torch.randint(-half, half, (args.batch, 2, s["n_lwe"] + 1), dtype=torch.int64, device=dev, generator=g)
        lwe_out = torch.empty((args.batch, ksz, s["n_lwe"] + 1), dtype=torch.int64, device=dev)
        mod.sync()

        def bootstrap():
            mod.lwe_mod_switch_2n_batched(ptr(lwe), ptr(lwe_in), s["n_lwe"], 2, s["base2k"], 2 * n, False, args.batch)
            mod.blind_rotation_execute_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), p, args.batch)
            mod.lwe_from_glwe_batched(ptr(lwe_out), s["n_lwe"], ptr(res), 0, ptr(kpm), kp, args.batch)
        bootstrap()
        mod.sync()
        R.sync_all()
        t0 = time.perf_counter()
        for _ in range(reps):
            bootstrap()
        mod.sync()
        R.sync_all()
        dtb = R.max_seconds((time.perf_counter() - t0) / reps)
        ks_stats = {"gate_bootstraps_per_s": args.batch * R.world / dtb, "ms_per_batch": dtb * 1e3,
                    "steps": "lwe_mod_switch_2n + blind_rotation_execute + lwe_from_glwe (key switch + sample extract), all device-resident"}
    out = {"metric": "CGGI blind rotations/s", "shape": args.shape, **s, "batch": args.batch, "value": args.batch * R.world / dt, "unit": "rotations/s",
           "ms_per_batch": dt * 1e3, "timed_calls": reps, "rounding_margin": margin, "kernel_classes_launches_ms": kstats, "digits_balanced": bool((res.min() >= -half).item() and (res.max() < half).item())}
    # which kernel instantiations ran, and the rotation priced against the three ceilings that can bound it (tools/roofline_models.py)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import roofline_models as rm
    notes = mod.dispatch_notes()
    model = rm.blind_rotation(n, s["n_lwe"], s["rank"], s["block_size"], s["dnum"], s["brk_size"], s["res_size"], args.batch)
    out["dispatch"] = notes
    out["roofline"] = rm.roofline(out["value"] / R.world, model, rm.key_share(notes))   # per GPU
    if ks_stats:
        out["gate_bootstrap"] = ks_stats
    # CPU port beside it (single thread) on a few of the same ciphertexts, and parity on those
    if args.cpu_cts:
        from oracle.ref import RefModule
        from poulpy_amd.layouts import VecZnx
        ref = RefModule(n, fast=True)
        # the oracle needs the key in ITS prepared order: prepare the same matrices with it
        brk_r = np.empty((s["n_lwe"], pm_elems), dtype=np.float64)
        mat_h = mat.cpu().numpy()
        from poulpy_amd.layouts import MatZnx
        for i in range(s["n_lwe"]):
            mz = MatZnx(n, s["dnum"], cols, cols, s["brk_size"], np.ascontiguousarray(np.roll(mat_h, i * 977)))
            pr = ref.vmp_pmat_alloc(s["dnum"], cols, cols, s["brk_size"])
            ref.vmp_prepare(pr, mz)
            brk_r[i] = pr.data.reshape(-1)
        xpa = ref.blind_rotation_x_pow_a()
        lut_h = VecZnx(n, 1, s["res_size"], np.ascontiguousarray(lut.cpu().numpy()))
        lwe_h = lwe[:args.cpu_cts].cpu().numpy()
        got = res[:args.cpu_cts].cpu().numpy()
        t0 = time.perf_counter()
        ok = True
        for b in range(args.cpu_cts):
            r = VecZnx(n, cols, s["res_size"])
            ref.blind_rotation_execute(r, s["base2k"], np.ascontiguousarray(lwe_h[b]), lut_h, brk_r, s["dnum"], s["brk_size"],
                                       s["block_size"], xpa)
            ok = ok and bool(np.array_equal(r.data, got[b]))
        cdt = (time.perf_counter() - t0) / args.cpu_cts
        out["cpu_port_1thread_per_s"] = 1.0 / cdt
        out["parity_on_cpu_sample"] = ok
        if ks_stats:
            # end-to-end parity of the bootstrap chain on the same sample (mod switch and final LWE)
            kmz = MatZnx(n, ksz, 1, cols, ksz, np.ascontiguousarray(kmat.cpu().numpy()))
            kpr = ref.vmp_pmat_alloc(ksz, 1, cols, ksz)
            ref.vmp_prepare(kpr, kmz)
            lin = lwe_in[:args.cpu_cts].cpu().numpy()
            lout = lwe_out[:args.cpu_cts].cpu().numpy()
            ok2 = True
            for b in range(args.cpu_cts):
                ok2 = ok2 and bool(np.array_equal(ref.mod_switch_2n(2 * n, lin[b], s["base2k"], False), lwe_h[b]))
                acc = VecZnx(n, cols, s["res_size"], np.ascontiguousarray(got[b]))
                ok2 = ok2 and bool(np.array_equal(ref.lwe_from_glwe(s["n_lwe"], ksz, s["base2k"], acc, s["base2k"], 0, kpr, 1, s["base2k"]), lout[b]))
            out["gate_bootstrap"]["parity_on_cpu_sample"] = ok2
    # every rank checked its own sample; the line reports the AND, the per-rank block and the job-level fields (tools/multirank.py)
    mine = {"value": args.batch / dt_mine, "ms_per_batch": dt_mine * 1e3, "global_first_index": lo, "parity_ok": out.get("parity_on_cpu_sample"),
            "rounding_margin": margin, "device": R.local_rank}
    per_rank = R.gather(mine)
    all_ok = R.all_true(out.get("parity_on_cpu_sample"))
    if R.distributed:
        out["parity_on_cpu_sample"] = all_ok if args.cpu_cts else None
        out["rounding_margin"] = max(e["rounding_margin"] for e in per_rank)
    out.update(R.line_fields(out["value"], args.ref_value, route, mod))
    out["per_rank"] = per_rank
    if R.rank == 0:
        print(json.dumps(out), flush=True)
    R.finish()
    if all_ok is False:
        raise SystemExit(3)   # a fast wrong answer is not a result


if __name__ == "__main__":
    main()
