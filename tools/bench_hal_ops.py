#!/usr/bin/env python3
"""Per-op HAL device rates: the reference's own "representative run" times the HAL ops individually
(poulpy-bench/benches/standard.rs:27-58: vec_znx_dft_apply, vec_znx_idft_apply, vmp_apply_dft_to_dft, svp_apply_dft_to_dft,
vec_znx_big_normalize; FFT sweep benches/fft.rs:6-29, m = 2^9 .. 2^15; VMP sweep src/params.rs:72-84).  Here each op runs on a
device-resident batch of containers through the batched C-ABI entry point of the same kernels (pz_*_batched; svp through the
per-container call on one container whose limbs are the batch), timed over K calls, with the op's own algorithmic bytes and the
fraction of the 8 TB/s HBM peak; the per-kernel-class times of one instrumented pass ride along.

    python tools/bench_hal_ops.py --op dft|idft|vmp|svp|normalize [--n 65536] [--limbs 8] [--cols 2] [--batch 1024] [--steps 20]
                                  [--rows R --cols-in CI --cols-out CO]   (vmp)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
HBM_PEAK_GBS = 8000.0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", choices=("dft", "idft", "vmp", "svp", "normalize"), required=True)
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--limbs", type=int, default=8)
    ap.add_argument("--cols", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--base2k", type=int, default=12)
    ap.add_argument("--rows", type=int, default=0, help="vmp: rows of the matrix (default = limbs)")
    ap.add_argument("--cols-in", type=int, default=0, help="vmp: input columns (default = cols)")
    ap.add_argument("--cols-out", type=int, default=0, help="vmp: output columns (default = cols)")
    args = ap.parse_args(argv)
    import torch
    from poulpy_amd.hal import Module
    if not torch.cuda.is_available():
        raise SystemExit("bench_hal_ops.py needs a HIP device (there is no CPU fallback)")
    n, cols, size, batch = args.n, args.cols, args.limbs, args.batch
    dev = torch.device("cuda", 0)
    mod = Module(n, device=0)
    g = torch.Generator(device=dev)
    g.manual_seed(0xA11CE)
    half = 1 << (args.base2k - 1)
    P = C.c_void_p
    poly = n * 8          # bytes of one polynomial in any of the three domains (i64, prepared f64, big i64)

    def i64(shape):
        return torch.randint(-half, half, shape, dtype=torch.int64, device=dev, generator=g)

    if args.op == "dft":
        a = i64((batch, size, cols, n))
        res = torch.empty((batch, size, cols, n), dtype=torch.float64, device=dev)

        def step():
            for c in range(cols):
                mod.vec_znx_dft_apply_batched(batch, 1, 0, P(res.data_ptr()), cols, size, c, P(a.data_ptr()), cols, size, c)
        units, unit = batch * size * cols, "polynomials"
        b_unit = 2 * poly
        what = f"vec_znx_dft_apply (hal_impl.rs:529), batch {batch} x VecZnx(n={n}, cols={cols}, size={size}), one call per column"
    elif args.op == "idft":
        a = i64((batch, size, cols, n))
        buf = torch.empty((batch, size, cols, n), dtype=torch.float64, device=dev)
        for c in range(cols):
            mod.vec_znx_dft_apply_batched(batch, 1, 0, P(buf.data_ptr()), cols, size, c, P(a.data_ptr()), cols, size, c)
        mod.sync()
        keep = buf.clone()

        def step():
            mod.vec_znx_idft_apply_consume_batched(batch, P(buf.data_ptr()), cols, size)   # in place: spectra -> VecZnxBig (i64)
        units, unit = batch * size * cols, "polynomials"
        b_unit = 2 * poly
        what = f"vec_znx_idft_apply_consume (hal_impl.rs:546), batch {batch} x VecZnxDft(n={n}, cols={cols}, size={size})"
    elif args.op == "vmp":
        rows = args.rows or size
        ci, co = args.cols_in or cols, args.cols_out or cols
        a_size = rows
        mat = i64((rows, ci, size, co, n))
        pmat = torch.empty((rows * ci * co * size * n,), dtype=torch.float64, device=dev)
        mod._ck(mod.lib.pz_vmp_prepare(mod.handle, P(pmat.data_ptr()), P(mat.data_ptr()), C.c_size_t(rows), C.c_size_t(ci), C.c_size_t(co), C.c_size_t(size)))
        a = i64((batch, a_size, ci, n))
        a_dft = torch.empty((batch, a_size, ci, n), dtype=torch.float64, device=dev)
        for c in range(ci):
            mod.vec_znx_dft_apply_batched(batch, 1, 0, P(a_dft.data_ptr()), ci, a_size, c, P(a.data_ptr()), ci, a_size, c)
        res = torch.empty((batch, size, co, n), dtype=torch.float64, device=dev)
        mod.sync()

        def step():
            mod.vmp_apply_dft_to_dft_batched(batch, P(res.data_ptr()), co, size, P(a_dft.data_ptr()), ci, a_size, P(pmat.data_ptr()), rows, ci, co, size, 0)
        units, unit = batch, "vector-matrix products"
        b_unit = (a_size * ci + size * co) * poly + rows * ci * co * size * poly / batch
        flops_unit = 8.0 * (a_size * ci) * (size * co) * (n // 2)
        what = (f"vmp_apply_dft_to_dft (hal_impl.rs:653), batch {batch}: VecZnxDft(n={n}, cols={ci}, size={a_size}) x VmpPMat(rows={rows}, cols_in={ci}, "
                f"cols_out={co}, size={size})")
    elif args.op == "svp":
        S = batch * size
        a = i64((S, 1, n))
        a_dft = torch.empty((S, 1, n), dtype=torch.float64, device=dev)
        mod.vec_znx_dft_apply_batched(1, 1, 0, P(a_dft.data_ptr()), 1, S, 0, P(a.data_ptr()), 1, S, 0)
        sc = i64((1, 1, n))
        ppol = torch.empty((n,), dtype=torch.float64, device=dev)
        mod._ck(mod.lib.pz_svp_prepare(mod.handle, P(ppol.data_ptr()), C.c_size_t(1), C.c_size_t(0), P(sc.data_ptr()), C.c_size_t(1), C.c_size_t(0)))
        res = torch.empty((S, 1, n), dtype=torch.float64, device=dev)
        mod.sync()

        def step():
            mod._ck(mod.lib.pz_svp_apply_dft_to_dft(mod.handle, P(res.data_ptr()), C.c_size_t(1), C.c_size_t(S), C.c_size_t(0), P(ppol.data_ptr()),
                                                   C.c_size_t(1), C.c_size_t(0), P(a_dft.data_ptr()), C.c_size_t(1), C.c_size_t(S), C.c_size_t(0)))
        units, unit = S, "polynomials"
        b_unit = 2 * poly
        what = f"svp_apply_dft_to_dft (hal_impl.rs:606), one VecZnxDft(n={n}, cols=1, size={S}) x SvpPPol"
    else:
        big = torch.randint(-(1 << 40), 1 << 40, (batch, size, cols, n), dtype=torch.int64, device=dev, generator=g)
        res = torch.empty((batch, size, cols, n), dtype=torch.int64, device=dev)

        def step():
            for c in range(cols):
                mod.vec_znx_big_normalize_batched(batch, P(res.data_ptr()), cols, size, args.base2k, 0, c, P(big.data_ptr()), cols, size, args.base2k, c)
        units, unit = batch * size * cols, "polynomials"
        b_unit = 2 * poly
        what = f"vec_znx_big_normalize (hal_impl.rs:431), batch {batch} x VecZnxBig(n={n}, cols={cols}, size={size}), base2k {args.base2k}, one call per column"

    for _ in range(args.warmup):
        step()
    mod.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    mod.sync()
    dt = time.perf_counter() - t0
    mod.set_kernel_timing(True)
    step()
    mod.sync()
    stats = {k: [v[0], round(v[1], 4)] for k, v in mod.kernel_stats().items() if v[0]}
    mod.set_kernel_timing(False)
    margin = None    # only the inverse transform rounds (forward transforms, products and the integer normalize have no f64 -> i64 step)
    if args.op == "idft":
        buf.copy_(keep)          # the timed steps ran in place on their own output: the spectra of the inputs again
        torch.cuda.synchronize()
        margin = mod.rounding_margin_of(step)
    value = units * args.steps / dt
    achieved = value * b_unit / 1e9
    line = {"metric": f"HAL op on a device-resident batch: {args.op}", "value": value, "unit": unit + "/s", "n_gpus": 1, "steps": args.steps,
            "ms_per_step": dt / args.steps * 1e3, "dtype": "f64", "data": "synthetic", "rounding_margin": margin, "config": {"workload": what},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_unit": b_unit, "kernel_classes_launches_ms": stats}}
    if args.op == "vmp":
        line["roofline"]["fp64_tflops"] = value * flops_unit / 1e12
        line["roofline"]["fp64_frac_of_68"] = value * flops_unit / 1e12 / 68.0
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
