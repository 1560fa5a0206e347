#!/usr/bin/env python3
"""Host threads over ONE module vs over sibling modules (pz_module_clone): fused GLWE external products on pinned host containers
(N = 2^16, 8 limbs; what the Rust shim's CoreImpl override issues), T threads, each its own ciphertexts.  PCIe-inclusive."""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poulpy_amd.hal import GlweOpParams, Module
from poulpy_amd.layouts import MatZnx


def pinned(mod, shape, dtype):
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = mod.lib.pz_alloc_bytes(C.c_size_t(nbytes))
    return np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype).reshape(shape)


def main():
    n, cols, size, k, per_call, calls = 65536, 2, 8, 12, 4, 12
    parent = Module(n)
    rng = np.random.default_rng(1)
    mat = MatZnx(n, size, cols, cols, size).fill_uniform(k, rng)
    pm = parent.vmp_pmat_alloc(size, cols, cols, size)
    parent.vmp_prepare(pm, mat)
    d_key = parent.device_alloc(pm.data.nbytes).upload(pm.data)
    p = GlweOpParams(rank=1, dnum=size, dsize=1, key_size=size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k, rank_out=1)
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)
    for T in (1, 2, 4, 8):
        for mode in ("one module", "siblings"):
            mods = [parent if mode == "one module" else parent.clone() for _ in range(T)]
            bufs = []
            for t in range(T):
                a = pinned(parent, (per_call, size, cols, n), np.int64)
                a[...] = rng.integers(-2048, 2048, a.shape)
                bufs.append((a, pinned(parent, a.shape, np.int64)))

            def work(t):
                a, r = bufs[t]
                for _ in range(calls):
                    mods[t].glwe_external_product_batched(hp(r), hp(a), d_key.ptr, p, per_call)

            work(0)
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            [x.start() for x in th]
            [x.join() for x in th]
            dt = time.perf_counter() - t0
            print(json.dumps({"threads": T, "mode": mode, "external_products_per_s": T * calls * per_call / dt,
                              "note": f"{per_call} ciphertexts per call, pinned host containers, device-resident key"}), flush=True)
            if mode == "siblings":
                for m in mods:
                    m.close()


if __name__ == "__main__":
    main()
