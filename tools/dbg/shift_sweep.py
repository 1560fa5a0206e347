import ctypes as C, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from poulpy_amd.hal import GlweOpParams, Module
n, size, cols, batch = 65536, 8, 2, 1024
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
print("a", hex(a.data_ptr()), "res", hex(res.data_ptr()))
def run(st2, st):
    mod.lib.pz_module_set_ws_shift(mod.handle, C.c_size_t(st2), C.c_size_t(st))
    mod.glwe_external_product_batched(ptr(res), ptr(a), ptr(key), p, batch)
    mod.sync()
    mod.set_kernel_timing(True)
    for _ in range(3):
        mod.glwe_external_product_batched(ptr(res), ptr(a), ptr(key), p, batch)
    mod.sync()
    ks = {k: round(v[1] / v[0], 3) for k, v in mod.kernel_stats().items() if v[0]}
    mod.set_kernel_timing(False)
    print(hex(st2), hex(st), "tail", ks["fused_tail"], "mid", ks["fused_mid"], "pass1", ks["fwd_pass1"], "sum", round(sum(ks.values()), 3), flush=True)
mod.lib.pz_module_debug_ws.restype = C.c_void_p
run(0, 0)
ws = mod.lib.pz_module_debug_ws(mod.handle)
t2 = ws + 8 * (1 << 30)       # key pinned: Pp not in ws?  (reported for orientation only)
print("ws", hex(ws), "res", hex(res.data_ptr()), "a", hex(a.data_ptr()), "ws-res mod 4M", hex((ws - res.data_ptr()) % (1 << 22)), "ws-a mod 4M", hex((ws - a.data_ptr()) % (1 << 22)))
for st in (0, 0x40000, 0x80000, 0xC0000, 0x140000, 0x1C0000):
    run(0xC0000, st)
