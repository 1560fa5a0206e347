"""Can the HBM-bound stages (pass 1, tail) of one wave of ciphertexts overlap with the latency-bound middle kernel of
another wave when they run on different streams?  Two modules (two streams), stage masks via pz_module_set_debug_stages."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poulpy_amd.hal import GlweOpParams, Module
N, cols, size, dnum, batch = 1 << 16, 2, 8, 8, int(os.environ.get("B", "64"))
A, B = Module(N, device=0), Module(N, device=0)
half = 2048
mat = torch.randint(-half, half, (N * dnum * cols * cols * size,), dtype=torch.int64, device="cuda")
pmat = torch.empty(mat.numel(), dtype=torch.float64, device="cuda")
a = torch.randint(-half, half, (2, batch, size, cols, N), dtype=torch.int64, device="cuda")
res = torch.empty_like(a)
torch.cuda.synchronize()
A._ck(A.lib.pz_vmp_prepare(A.handle, C.c_void_p(pmat.data_ptr()), C.c_void_p(mat.data_ptr()), C.c_size_t(dnum), C.c_size_t(cols), C.c_size_t(cols), C.c_size_t(size)))
A.sync()
p = GlweOpParams(rank=1, dnum=dnum, dsize=1, key_size=size, key_base2k=12, a_size=size, a_base2k=12, res_size=size, res_base2k=12, rank_out=1)
def run(mod, i):
    mod.glwe_external_product_batched(C.c_void_p(res[i].data_ptr()), C.c_void_p(a[i].data_ptr()), C.c_void_p(pmat.data_ptr()), p, batch)
def timed(fn, reps=10):
    fn(); A.sync(); B.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    A.sync(); B.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for m in (A, B): m.lib.pz_module_set_debug_stages(m.handle, 7)
full = timed(lambda: run(A, 0))
A.lib.pz_module_set_debug_stages(A.handle, 2); mid = timed(lambda: run(A, 0))
B.lib.pz_module_set_debug_stages(B.handle, 5); io = timed(lambda: run(B, 1))
both = timed(lambda: (run(A, 0), run(B, 1)))
print(f"batch {batch}: full pipeline {full:.3f} ms | middle only {mid:.3f} | pass1+tail only {io:.3f} | both streams concurrently {both:.3f} (sum {mid+io:.3f})")
