#!/usr/bin/env python3
"""glwe_automorphism_assign / _add_assign (res == a) timed on device-resident batches: the in-place forms bench.py has no switch for.
   python tools/dbg/auto_inplace.py [--limbs 16 --batch 512 --mode automorphism|add --galois 5]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poulpy_amd.hal import GlweOpParams, Module

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536); ap.add_argument("--limbs", type=int, default=16); ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--base2k", type=int, default=12); ap.add_argument("--mode", default="automorphism"); ap.add_argument("--galois", type=int, default=5)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--out-of-place", action="store_true")
args = ap.parse_args()
n, size, k = args.n, args.limbs, args.base2k
dev = torch.device("cuda", 0)
mod = Module(n, device=0)
half = 1 << (k - 1)
g = torch.Generator(device=dev); g.manual_seed(7)
a = torch.randint(-half, half, (args.batch, size, 2, n), dtype=torch.int64, device=dev, generator=g)
res = torch.empty_like(a) if args.out_of_place else a
mat = torch.randint(-half, half, (n * size * 1 * 2 * size,), dtype=torch.int64, device=dev, generator=g)
pmat = torch.empty(mat.numel(), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
mod._ck(mod.lib.pz_vmp_prepare(mod.handle, C.c_void_p(pmat.data_ptr()), C.c_void_p(mat.data_ptr()), C.c_size_t(size), C.c_size_t(1), C.c_size_t(2), C.c_size_t(size)))
mod.pin_key(C.c_void_p(pmat.data_ptr()), size, 1, 2, size)
p = GlweOpParams(rank=1, dnum=size, dsize=1, key_size=size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k, rank_out=1)
run = lambda: mod.glwe_automorphism_batched(C.c_void_p(res.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(pmat.data_ptr()), p, args.galois, args.mode, args.batch)
for _ in range(3):
    run()
mod.sync()
t0 = time.perf_counter()
for _ in range(args.steps):
    run()
mod.sync()
dt = (time.perf_counter() - t0) / args.steps
mod.set_kernel_timing(True); run(); mod.sync()
stats = {kn: [c, round(ms, 3)] for kn, (c, ms) in mod.kernel_stats().items() if c}
print("%-12s %-12s limbs=%d galois=%d  %9.0f /s  %s" % ("out-of-place" if args.out_of_place else "in-place", args.mode, size, args.galois, args.batch / dt, stats), flush=True)
