"""Create / use / destroy modules repeatedly: no crash, device memory returns to its starting level."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from poulpy_amd.hal import Module
from poulpy_amd.layouts import VecZnx
free0 = torch.cuda.mem_get_info()[0]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    n = [256, 2048, 4096, 65536][it % 4]
    m = Module(n)
    a = VecZnx(n, 2, 3).fill_uniform(12, np.random.default_rng(it))
    d = m.vec_znx_dft_alloc(2, 3)
    for c in range(2):
        m.vec_znx_dft_apply(1, 0, d, c, a, c)
    big = m.vec_znx_idft_apply_consume(d)
    assert np.array_equal(big.data, a.data)
    buf = m.device_alloc(1 << 20)
    m.pin_key(buf.ptr, 1, 1, 1, 1) if n >= 4096 * 2 else None
    buf2 = m.device_alloc(8 * n * 4)
    m.pin_key(buf2.ptr, 1, 1, 2, 2)
    m.unpin_key(buf2.ptr)
    buf.free(); buf2.free()
    m.close()
free1 = torch.cuda.mem_get_info()[0]
print("device memory drift: %.1f MiB" % ((free0 - free1) / 2**20))
