"""Does a MALL-resident working set make the FFT passes faster?  dft_apply_batched + idft_consume_batched on
`batch` ciphertexts (16 polys each, N=2^16), repeated on the same buffers; per-polynomial kernel time by class."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poulpy_amd.hal import Module
N, cols, size = 1 << 16, 2, 8
mod = Module(N, device=0)
for batch in (2, 4, 8, 16, 32, 128, 256):
    a = torch.randint(-2048, 2048, (batch, size, cols, N), dtype=torch.int64, device="cuda")
    d = torch.empty((batch, size, cols, N), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    reps = max(2, 512 // batch)
    def run():
        for c in range(cols):
            mod._ck(mod.lib.pz_vec_znx_dft_apply_batched(mod.handle, C.c_size_t(batch), C.c_size_t(1), C.c_size_t(0), C.c_void_p(d.data_ptr()),
                    C.c_size_t(cols), C.c_size_t(size), C.c_size_t(c), C.c_void_p(a.data_ptr()), C.c_size_t(cols), C.c_size_t(size), C.c_size_t(c)))
        mod._ck(mod.lib.pz_vec_znx_idft_apply_consume_batched(mod.handle, C.c_size_t(batch), C.c_void_p(d.data_ptr()), C.c_size_t(cols), C.c_size_t(size)))
    run(); mod.sync()
    mod.set_kernel_timing(True)
    for _ in range(reps): run()
    st = mod.kernel_stats(); mod.set_kernel_timing(False)
    polys = batch * cols * size * reps
    ws = batch * cols * size * N * 8 / 2**20
    print(f"batch {batch:4d} working set in/T/out {ws:7.0f} MiB each: " + "  ".join(f"{k} {v[1]/polys*1e3:6.3f} us/poly" for k, v in st.items() if v[0]))
