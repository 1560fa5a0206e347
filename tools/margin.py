"""Rounding margin of the inverse transform on device: max |x - round(x)| over a batch of external products at the
metric shape (N=2^16, 8 limbs, rank 1, dnum 8) for several base2k (SURVEY.md §7 'exactness margin').  0.5 = failure."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poulpy_amd.hal import GlweOpParams, Module
N, cols, size, dnum, batch = 1 << 16, 2, 8, 8, 8
mod = Module(N, device=0)
for base2k in (12, 14, 16, 17, 18, 19):
    half = 1 << (base2k - 1)
    g = torch.Generator(device="cuda"); g.manual_seed(base2k)
    mat = torch.randint(-half, half, (N * dnum * cols * cols * size,), dtype=torch.int64, device="cuda", generator=g)
    pmat = torch.empty(mat.numel(), dtype=torch.float64, device="cuda")
    a = torch.randint(-half, half, (batch, size, cols, N), dtype=torch.int64, device="cuda", generator=g)
    res = torch.empty_like(a)
    torch.cuda.synchronize()
    mod._ck(mod.lib.pz_vmp_prepare(mod.handle, C.c_void_p(pmat.data_ptr()), C.c_void_p(mat.data_ptr()), C.c_size_t(dnum),
                                   C.c_size_t(cols), C.c_size_t(cols), C.c_size_t(size)))
    p = GlweOpParams(rank=1, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=1)
    mod.set_margin_probe(True)
    mod.glwe_external_product_batched(C.c_void_p(res.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(pmat.data_ptr()), p, batch)
    mod.sync()
    m = mod.get_margin()
    mod.set_margin_probe(False)
    print(f"base2k {base2k:2d}: max |x - round(x)| = {m:.3e}   (uniform inputs, {batch} ciphertexts, {batch*16*N} coefficients)")
