"""placement sweep of T2' (debug knob pz_module_set_ws_shift) at a given ring degree: tail / mid / pass-1 ms per launch"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from poulpy_amd.hal import GlweOpParams, Module
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
size, cols, batch = 8, 2, 1024 * 65536 // n
dev = torch.device("cuda", 0)
mod = Module(n, device=0)
g = torch.Generator(device=dev); g.manual_seed(1)
a = torch.randint(-2048, 2048, (batch, size, cols, n), dtype=torch.int64, device=dev, generator=g)
res = torch.empty_like(a)
mat = torch.randint(-2048, 2048, (size * cols * cols * size * n,), dtype=torch.int64, device=dev, generator=g)
key = torch.empty(mat.numel(), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
mod._ck(mod.lib.pz_vmp_prepare(mod.handle, C.c_void_p(key.data_ptr()), C.c_void_p(mat.data_ptr()), C.c_size_t(size), C.c_size_t(cols), C.c_size_t(cols), C.c_size_t(size)))
mod.sync()
mod.pin_key(C.c_void_p(key.data_ptr()), size, cols, cols, size)
p = GlweOpParams(rank=1, dnum=size, dsize=1, key_size=size, key_base2k=12, a_size=size, a_base2k=12, res_size=size, res_base2k=12, rank_out=1)
ptr = lambda t: C.c_void_p(t.data_ptr())
def run(st2):
    mod.lib.pz_module_set_ws_shift(mod.handle, C.c_size_t(st2))
    mod.glwe_external_product_batched(ptr(res), ptr(a), ptr(key), p, batch)
    mod.sync()
    mod.set_kernel_timing(True)
    for _ in range(3):
        mod.glwe_external_product_batched(ptr(res), ptr(a), ptr(key), p, batch)
    mod.sync()
    ks = {k: round(v[1] / v[0], 3) for k, v in mod.kernel_stats().items() if v[0]}
    mod.set_kernel_timing(False)
    print(n, hex(st2), "tail", ks.get("fused_tail"), "mid", ks.get("fused_mid"), "pass1", ks.get("fwd_pass1"), flush=True)
u = n * 4   # bytes between the two coefficient halves
for st2 in (0, u, 2 * u, 3 * u, 4 * u, 5 * u, 7 * u, 1 << 20, 3 << 18):
    run(st2)
