#!/usr/bin/env python3
"""Secondary measurement: GLWE tensorings / s (glwe_tensor_apply, poulpy-core/src/operations/glwe.rs:700-807 - the convolution half of a
CKKS multiplication; the relinearization half is `bench.py --op relinearize`) on one MI355X, at the shape of BASELINE configs[4]:
N = 2^16, rank 1, 16 limbs, base2k 12 (so that FFT64 represents it).  Device-resident pairs of ciphertexts in, 3-column GLWETensors out;
a few outputs are compared bit for bit with the oracle's composition (cnv_prepare_left / right, cnv_apply_dft, cnv_pairwise_apply_dft,
idft, normalize).

    python tools/bench_tensor.py [--batch 256] [--limbs 16] [--n 65536] [--steps 10] [--mode apply|square] [--relin] [--gpus N]

--gpus N (round 6; BASELINE configs[4]: "batch = 2048 sharded over 8 MI355X" = 256 per GPU): one rank per GPU (tools/multirank.py), `--batch`
pairs per GPU (weak scaling); with --relin the tensor key is prepared on rank 0 and broadcast once (broadcast_key_agreed); no other collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--limbs", type=int, default=16)
    ap.add_argument("--base2k", type=int, default=12)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("apply", "square"), default="apply")
    ap.add_argument("--parity-samples", type=int, default=2)
    ap.add_argument("--relin", action="store_true", help="follow every tensoring with glwe_tensor_relinearize (tensor key 1 -> 1, dnum = limbs): "
                    "the whole GLWE multiplication of a CKKS multiply (operations/glwe.rs:541-607 after :700-807)")
    ap.add_argument("--one-call", action="store_true", help="with --relin: tensoring + relinearization as ONE call with the tensor in scratch "
                    "(pz_glwe_tensor_mul_relinearize_batched: poulpy-ckks's ckks_mul_into_default, leveled/default/mul.rs:49-85) instead of the two calls")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import multirank
    multirank.add_arguments(ap)
    args = ap.parse_args()
    R = multirank.enter(__file__, args.gpus, sys.argv[1:])   # --gpus N > 1 started plainly: launches the ranks and exits with their code
    import ctypes as C
    torch, dist = R.init()
    from poulpy_amd.hal import GlweOpParams, GlweTensorParams, Module
    from poulpy_amd.layouts import MatZnx, VecZnx
    n, size, k, rank = args.n, args.limbs, args.base2k, 1
    cols, tcols = rank + 1, (rank + 1) * (rank + 2) // 2
    dev = R.dev
    mod = Module(n, device=R.local_rank)
    g = torch.Generator(device=dev)
    lo, hi = R.shard(args.batch * R.world)   # this rank's block of the global batch (weak scaling: args.batch pairs per GPU)
    g.manual_seed(0x7e50 + lo)               # a rank's operands are drawn from its first global index (rank 0 of any world: the round-5 inputs)
    half = 1 << (k - 1)
    a = torch.randint(-half, half, (args.batch, size, cols, n), dtype=torch.int64, device=dev, generator=g)
    b = a if args.mode == "square" else torch.randint(-half, half, (args.batch, size, cols, n), dtype=torch.int64, device=dev, generator=g)
    res = torch.zeros((args.batch, size, tcols, n), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    cnv_offset = size * k - 20     # the offset of the 16-limb parity test (tests/test_gpu_cnv.py)
    p = GlweTensorParams(rank=rank, a_size=size, b_size=size, ab_base2k=k, a_effective_k=size * k, b_effective_k=size * k, res_size=size,
                         res_base2k=k, cnv_offset=cnv_offset)

    # --relin: a tensor key (GGLWE rank*(rank+1)/2 -> rank: rows = dnum, cols_in = 1, cols_out = 2), synthetic digits, prepared on the device
    out = pmat = mat = rp = route = None
    if args.relin:
        dnum = size
        gk = torch.Generator(device=dev)
        gk.manual_seed(0x7e51)   # the key digits: the same on every rank (rank 0 prepares and broadcasts; every rank's CPU check re-prepares them)
        mat = torch.randint(-half, half, (n * dnum * 1 * cols * size,), dtype=torch.int64, device=dev, generator=gk)
        pmat = torch.empty(mat.numel(), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        if R.rank == 0:
            mod._ck(mod.lib.pz_vmp_prepare(mod.handle, C.c_void_p(pmat.data_ptr()), C.c_void_p(mat.data_ptr()), C.c_size_t(dnum), C.c_size_t(1),
                                           C.c_size_t(cols), C.c_size_t(size)))
            mod.sync()
        route = R.broadcast_keys(mod, [pmat], args.bcast, log=lambda m: print(f"[bench_tensor] {m}", file=sys.stderr, flush=True))   # the only collective
        mod.pin_key(C.c_void_p(pmat.data_ptr()), dnum, 1, cols, size)
        out = torch.zeros((args.batch, size, cols, n), dtype=torch.int64, device=dev)
        rp = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k, rank_out=rank)
        torch.cuda.synchronize()

    if args.one_call and not args.relin:
        raise SystemExit("--one-call goes with --relin")

    def run():
        if args.one_call:
            mod.glwe_tensor_mul_relinearize_batched(C.c_void_p(out.data_ptr()), C.c_void_p(a.data_ptr()), None if args.mode == "square" else C.c_void_p(b.data_ptr()),
                                                    C.c_void_p(pmat.data_ptr()), p, rp, args.mode, args.batch)
            return
        mod.glwe_tensor_apply_batched(C.c_void_p(res.data_ptr()), C.c_void_p(a.data_ptr()), None if args.mode == "square" else C.c_void_p(b.data_ptr()),
                                      p, args.mode, args.batch)
        if args.relin:
            mod.glwe_tensor_relinearize_batched(C.c_void_p(out.data_ptr()), C.c_void_p(res.data_ptr()), C.c_void_p(pmat.data_ptr()), rp, args.batch)

    for _ in range(args.warmup):
        run()
    mod.sync()
    R.sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    mod.sync()
    R.sync_all()
    dt_mine = (time.perf_counter() - t0) / args.steps
    dt = R.max_seconds(dt_mine)   # the slowest rank's clock around the same barrier-bracketed region
    mod.set_kernel_timing(True)
    run()
    mod.sync()
    stats = {kname: [c, round(ms, 3)] for kname, (c, ms) in mod.kernel_stats().items() if c}
    mod.set_kernel_timing(False)
    margin = mod.rounding_margin_of(run)
    # parity of a few outputs against the oracle
    ok = None
    if args.parity_samples:
        from oracle.ref import RefModule
        ref = RefModule(n)   # strict build (-ffp-contract=off): the bits the parity tests pin
        ok = True
        got = res.cpu().numpy()
        for t in sorted(set(np.linspace(0, args.batch - 1, args.parity_samples).astype(int).tolist())):
            av = VecZnx(n, cols, size, a[t].cpu().numpy().copy())
            bv = av if args.mode == "square" else VecZnx(n, cols, size, b[t].cpu().numpy().copy())
            r = VecZnx(n, tcols, size, np.zeros((size, tcols, n), dtype=np.int64))
            if args.mode == "square":
                ref.glwe_tensor_square_apply(cnv_offset, r, k, av, size * k, k)
            else:
                ref.glwe_tensor_apply(cnv_offset, r, k, av, size * k, bv, size * k, k, add_assign=False)
            if not args.one_call:   # (one call: the tensor never leaves the workspace)
                ok = ok and bool(np.array_equal(got[t], r.data))
            if args.relin:
                if t == 0:
                    pm = ref.vmp_pmat_alloc(size, 1, cols, size)
                    ref.vmp_prepare(pm, MatZnx(n, size, 1, cols, size, np.ascontiguousarray(mat.cpu().numpy())))
                want = VecZnx(n, cols, size)
                ref.glwe_tensor_relinearize(want, k, r, k, pm, 1, k)
                ok = ok and bool(np.array_equal(out[t].cpu().numpy(), want.data))
    # algorithmic bytes per tensoring: both operands read, the tensor written; flops: forward transforms of both operands, the limb
    # convolution of cols*(cols+1)/2 column pairs (Karatsuba for the cross column: cnv_pairwise), inverse transforms of the tensor
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import roofline_models as rm
    model = rm.tensoring(n, rank, size, args.mode, args.relin, args.batch, one_call=args.one_call)   # the byte / flop model lives beside the other benched operations'
    nb, flops = model["hbm_bytes"], model["flops"]
    rate = args.batch * R.world / dt
    per_rank = R.gather({"value": args.batch / dt_mine, "ms_per_step": dt_mine * 1e3, "global_first_index": lo, "parity_ok": ok, "rounding_margin": margin,
                         "device": R.local_rank})
    all_ok = R.all_true(ok)
    if R.distributed:
        ok = all_ok if args.parity_samples else None
        margin = max(e["rounding_margin"] for e in per_rank)
    if R.rank == 0:
        print(json.dumps({**R.line_fields(rate, args.ref_value, route, mod), "per_rank": per_rank,
        "metric": ("GLWE multiplications/s (glwe_tensor_%s + glwe_tensor_relinearize)" if args.relin else "GLWE tensorings/s (glwe_tensor_%s)") % ("apply" if args.mode == "apply" else "square_apply"),
        "value": rate, "unit": "multiplications/s" if args.relin else "tensorings/s", "ms_per_step": dt * 1e3, "batch": args.batch, "parity_ok": ok, "rounding_margin": margin,
        "config": {"workload": f"glwe_tensor_{args.mode}" + (" + glwe_tensor_relinearize (tensor key 1 -> 1, dnum = limbs)" if args.relin else "") + (", ONE call, tensor in scratch" if args.one_call else "") + f" (rank 1: 2-column GLWE x GLWE -> 3-column GLWETensor), N={n}, {size} limbs, base2k={k}, cnv_offset={cnv_offset}",
                   "batch_per_gpu": args.batch},
        "kernel_classes_launches_ms": stats,
        "knobs": {k_: v_ for k_, v_ in os.environ.items() if k_.startswith("POULPY_DBG_")},
        "roofline": {"bound": "hbm", "achieved": rate / R.world * nb / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": rate / R.world * nb / 1e9 / 8000.0,   # per GPU
                     "algorithmic_bytes_per_unit": nb, "fp64_tflops": rate / R.world * flops / 1e12, "fp64_frac": rate / R.world * flops / 1e12 / 68.0},
        "dtype": "f64", "data": "synthetic"}), flush=True)
    R.finish()
    if all_ok is False:
        raise SystemExit(3)   # a fast wrong answer is not a result


if __name__ == "__main__":
    main()
