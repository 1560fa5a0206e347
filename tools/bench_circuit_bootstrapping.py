#!/usr/bin/env python3
"""Secondary measurement: circuit bootstrappings (LWE -> GGSW, constant mode) / s on one MI355X for a batch of LWE ciphertexts,
at the shape of the reference's own benchmark (poulpy-bench/src/bench_suite/schemes/circuit_bootstrapping.rs:47-85:
n_glwe 1024, n_lwe 574, block 7, base2k 13 everywhere, rank 2; brk / atk / tsk k = 52 (4 limbs), dnum 3; GGSW k = 26
(2 limbs), dnum 2; log_domain 1, extension_factor 1).  Synthetic key material and lookup table (uniform digits); the
oracle's composition is timed single-threaded on a few of the same ciphertexts beside it and compared bit for bit.

    python tools/bench_circuit_bootstrapping.py [--batch 512] [--reps 3] [--n-lwe 574] [--cpu-cts 1] [--gpus N]

--gpus N (round 6; BASELINE configs[3]: "batch = 8192 sharded over 8 MI355X" = 1024 per GPU): one rank per GPU (tools/multirank.py), `--batch` LWEs
per GPU (weak scaling); the blind-rotation key (n_lwe prepared GGSWs), the automorphism keys and the tensor keys are prepared on rank 0 and
broadcast once (broadcast_key_agreed: pz_bcast_key = RCCL inside the C ABI, or torch.distributed); no other collective.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SHAPE = dict(n=1024, n_lwe=574, rank=2, block_size=7, base2k=13, brk_dnum=3, glwe_size=4, atk_dnum=3, atk_size=4, tsk_dnum=3, tsk_size=4,
             res_dnum=2, res_size=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024, help="LWEs per call (BASELINE configs[3]: 8192 over 8 GPUs = 1024 per GPU; 512 was the default of rounds 2 - 4)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--n-lwe", type=int, default=0)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--cpu-cts", type=int, default=1)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import multirank
    multirank.add_arguments(ap)
    args = ap.parse_args()
    R = multirank.enter(__file__, args.gpus, sys.argv[1:])   # --gpus N > 1 started plainly: launches the ranks and exits with their code
    torch, dist = R.init()
    from poulpy_amd.hal import BlindRotationParams, CircuitBootstrappingParams, Module
    s = dict(SHAPE)
    if args.n_lwe:
        s["n_lwe"] = args.n_lwe
    if args.rank:
        s["rank"] = args.rank
    n, rank, cols = s["n"], s["rank"], s["rank"] + 1
    log_n = n.bit_length() - 1
    dev = R.dev
    mod = Module(n, device=R.local_rank)
    half = 1 << (s["base2k"] - 1)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def prepare(mat_i64, rows, cols_in, size):
        pm = torch.empty(mat_i64.numel(), dtype=torch.float64, device=dev)
        if R.rank == 0:   # rank 0 prepares every key; the others receive them below
            mod._ck(mod.lib.pz_vmp_prepare(mod.handle, ptr(pm), ptr(mat_i64), C.c_size_t(rows), C.c_size_t(cols_in), C.c_size_t(cols), C.c_size_t(size)))
            mod.sync()   # stream-ordered on the module's stream: the source must outlive the kernels that read it
        return pm

    def synth(rows, cols_in, size, count):
        base = torch.randint(-half, half, (n * rows * cols_in * cols * size,), dtype=torch.int64, device=dev, generator=g)
        mats = [torch.roll(base, i * 977) for i in range(count)]   # distinct keys (one shared key would let the caches lie)
        torch.cuda.synchronize()   # torch's stream and the module's stream are not ordered: the rolls must have finished
        return mats, [prepare(m, rows, cols_in, size) for m in mats]

    brk_m, brk_p = synth(s["brk_dnum"], cols, s["glwe_size"], s["n_lwe"])
    brk = torch.stack(brk_p)
    del brk_p
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    atk_m, atk_p = synth(s["atk_dnum"], rank, s["atk_size"], len(gals))
    tsk_m, tsk_p = synth(s["tsk_dnum"], rank, s["tsk_size"], rank)
    torch.cuda.synchronize()
    # the only collectives: the blind-rotation key (n_lwe GGSWs in one tensor, several 64 MiB buckets), the log2(N) automorphism keys, the tensor keys
    route = R.broadcast_keys(mod, [brk, *atk_p, *tsk_p], args.bcast, log=lambda m: print(f"[bench_circuit_bootstrapping] {m}", file=sys.stderr, flush=True))
    lut = torch.randint(-half, half, (s["glwe_size"], 1, n), dtype=torch.int64, device=dev, generator=g)
    lo, hi = R.shard(args.batch * R.world)   # this rank's block of the global batch (weak scaling: args.batch per GPU)
    g.manual_seed(0xCB7 + lo)                # a rank's LWEs are drawn from its first global index
    lwe = torch.randint(-n, n, (args.batch, s["n_lwe"] + 1), dtype=torch.int64, device=dev, generator=g)
    gap = 2 * (n // 8)    # lut.drift = n/4 halves for a 1-bit domain with two table entries per bit; any even gap times the same
    res = torch.empty((args.batch, s["res_dnum"], cols, s["res_size"], cols, n), dtype=torch.int64, device=dev)
    p = CircuitBootstrappingParams(
        br=BlindRotationParams(rank=rank, n_lwe=s["n_lwe"], block_size=s["block_size"], dnum=s["brk_dnum"], brk_size=s["glwe_size"],
                               base2k=s["base2k"], res_size=s["glwe_size"], lut_size=s["glwe_size"]),
        atk_dnum=s["atk_dnum"], atk_size=s["atk_size"], tsk_dnum=s["tsk_dnum"], tsk_size=s["tsk_size"], res_dnum=s["res_dnum"],
        res_size=s["res_size"], gap=gap)
    nbytes = mod.circuit_bootstrapping_tmp_bytes(p, args.batch)
    tmp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def run():
        mod.circuit_bootstrapping_execute_to_constant_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), gals, [t.data_ptr() for t in atk_p],
                                                              [t.data_ptr() for t in tsk_p], p, ptr(tmp), nbytes, args.batch)

    mod.dispatch_notes(reset=True)
    run()
    mod.sync()
    R.sync_all()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    t_submit = (time.perf_counter() - t0) / args.reps    # host time to issue one call (asynchronous: before the sync)
    mod.sync()
    R.sync_all()
    dt_mine = (time.perf_counter() - t0) / args.reps
    dt = R.max_seconds(dt_mine)   # the slowest rank's clock around the same barrier-bracketed region
    mod.set_kernel_timing(True)   # one more pass with per-class HIP-event timing (not the timed run)
    run()
    mod.sync()
    kstats = {k: (v[0], round(v[1], 3)) for k, v in mod.kernel_stats().items() if v[0]}
    mod.set_kernel_timing(False)
    margin = mod.rounding_margin_of(run)   # blind rotation, the traces' key switches and the row expansion: the worst of all their roundings
    out = {"metric": "circuit bootstrappings/s (LWE -> GGSW, constant mode)", **s, "batch": args.batch, "value": args.batch * R.world / dt, "unit": "bootstrappings/s",
           "ms_per_batch": dt * 1e3, "rounding_margin": margin, "host_submit_ms_per_call": t_submit * 1e3, "graph_launches": mod.graph_launches(),
           "kernel_classes_launches_ms": kstats,
           "digits_balanced": bool((res.min() >= -half).item() and (res.max() <= half).item())}
    # the composition priced against the three ceilings (tools/roofline_models.py): blind rotation + res_dnum traces of log2(N)
    # automorphism key switches + res_dnum x rank key switches of ggsw_expand_row; the key stream of each part divided by the
    # ciphertexts that share one fetch in the kernel that ran (dispatch notes)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import roofline_models as rm
    notes = mod.dispatch_notes()
    br_m = rm.blind_rotation(n, s["n_lwe"], rank, s["block_size"], s["brk_dnum"], s["glwe_size"], s["glwe_size"], args.batch)
    logn = n.bit_length() - 1
    tr_m = rm.glwe_op(n, rank, cols, s["glwe_size"], s["atk_size"], s["atk_dnum"], args.batch * s["res_dnum"], a_cols=cols, extra_in_polys=s["glwe_size"])
    ex_m = rm.glwe_op(n, rank, cols, s["res_size"], s["tsk_size"], s["tsk_dnum"], args.batch * s["res_dnum"], a_cols=cols)
    br_share = rm.key_share(";".join(x for x in notes.split(";") if "k_br" in x)) if "k_br" in notes else 1
    ks_share = rm.key_share(";".join(x for x in notes.split(";") if "k_mid128" in x)) if "k_mid128" in notes else 1
    n_tr, n_ex = s["res_dnum"] * logn, s["res_dnum"] * rank
    model = {"hbm_bytes": br_m["hbm_bytes"] + n_tr * tr_m["hbm_bytes"] + n_ex * ex_m["hbm_bytes"],
             "flops": br_m["flops"] + n_tr * tr_m["flops"] + n_ex * ex_m["flops"],
             "key_stream_bytes": br_m["key_stream_bytes"] / br_share + (n_tr * tr_m["key_stream_bytes"] + n_ex * ex_m["key_stream_bytes"]) / ks_share}
    out["dispatch"] = notes
    out["roofline"] = rm.roofline(out["value"] / R.world, model, 1)   # per GPU
    out["roofline"]["l2_stream"]["ciphertexts_per_key_fetch"] = {"blind_rotation": br_share, "key_switches": ks_share}
    out["roofline"]["parts"] = {"blind_rotation_flops": br_m["flops"], "trace_flops": n_tr * tr_m["flops"], "expand_row_flops": n_ex * ex_m["flops"]}
    if args.cpu_cts:
        from oracle.ref import RefModule
        from poulpy_amd.layouts import MatZnx, VecZnx
        ref = RefModule(n, fast=True)

        def ref_prepare(mats, rows, cols_in, size):
            outp = []
            for m in mats:
                pr = ref.vmp_pmat_alloc(rows, cols_in, cols, size)
                ref.vmp_prepare(pr, MatZnx(n, rows, cols_in, cols, size, np.ascontiguousarray(m.cpu().numpy())))
                outp.append(pr)
            return outp

        brk_r = np.stack([pr.data.reshape(-1) for pr in ref_prepare(brk_m, s["brk_dnum"], cols, s["glwe_size"])])
        atk_r = ref_prepare(atk_m, s["atk_dnum"], rank, s["atk_size"])
        tsk_r = ref_prepare(tsk_m, s["tsk_dnum"], rank, s["tsk_size"])
        xpa = ref.blind_rotation_x_pow_a()
        lut_h = VecZnx(n, 1, s["glwe_size"], np.ascontiguousarray(lut.cpu().numpy()))
        lwe_h = lwe[:args.cpu_cts].cpu().numpy()
        got = res[:args.cpu_cts].cpu().numpy()
        ok = True
        t0 = time.perf_counter()
        for b in range(args.cpu_cts):
            gg = MatZnx(n, s["res_dnum"], cols, cols, s["res_size"])
            ref.circuit_bootstrap_to_constant(gg, s["base2k"], np.ascontiguousarray(lwe_h[b]), lut_h, brk_r, s["brk_dnum"], s["glwe_size"],
                                              s["glwe_size"], s["block_size"], xpa, gals, atk_r, tsk_r, gap)
            ok = ok and bool(np.array_equal(gg.data, got[b]))
        cdt = (time.perf_counter() - t0) / args.cpu_cts
        out["cpu_port_1thread_per_s"] = 1.0 / cdt
        out["parity_on_cpu_sample"] = ok
    # every rank checked its own sample; the line reports the AND, the per-rank block and the job-level fields (tools/multirank.py)
    per_rank = R.gather({"value": args.batch / dt_mine, "ms_per_batch": dt_mine * 1e3, "global_first_index": lo, "parity_ok": out.get("parity_on_cpu_sample"),
                         "rounding_margin": margin, "device": R.local_rank})
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
