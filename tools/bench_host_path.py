#!/usr/bin/env python3
"""PCIe-inclusive rate of the per-op host-pointer path (the HalImpl-level boundary): one GLWE (x) GGSW external product at the
metric shape (N = 2^16, 8 limbs, rank 1, dnum 8) written as the reference's op sequence over HOST containers
(external_product/glwe.rs:197-271: dft_apply x cols, vmp_apply_dft_to_dft, idft_apply_consume, big_normalize x cols); every call
stages its operands through the module's device arena (H2D, kernels, D2H).  Not the headline metric: DESIGN.md §1.
The prepared GGSW is uploaded once and passed as a device pointer (as a shim's DeviceBuf would be)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from poulpy_amd.hal import Module
from poulpy_amd.layouts import MatZnx, VecZnx, VmpPMat


def pinned(mod, shape, dtype):
    """numpy view over pz_alloc_bytes memory (hipHostMalloc: what the shim's OwnedBuf is)"""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = mod.lib.pz_alloc_bytes(C.c_size_t(nbytes))
    buf = (C.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def main():
    n, cols, size, dnum, k, reps = 65536, 2, 8, 8, 12, 20
    use_pinned = "--pinned" in sys.argv
    mod = Module(n)
    rng = np.random.default_rng(1)
    mat = MatZnx(n, dnum, cols, cols, size).fill_uniform(k, rng)
    pm = mod.vmp_pmat_alloc(dnum, cols, cols, size)
    mod.vmp_prepare(pm, mat)
    d_key = mod.device_alloc(pm.data.nbytes).upload(pm.data)
    pm_dev = VmpPMat.__new__(VmpPMat)
    a = VecZnx(n, cols, size).fill_uniform(k, rng)
    res = VecZnx(n, cols, size)
    if use_pinned:
        from poulpy_amd.layouts import VecZnxDft
        pa = pinned(mod, a.data.shape, np.int64); pa[...] = a.data; a = VecZnx(n, cols, size, pa)
        res = VecZnx(n, cols, size, pinned(mod, res.data.shape, np.int64))
        ad_p = VecZnxDft(n, cols, size, pinned(mod, (size, cols, n), np.float64))
        rd_p = VecZnxDft(n, cols, size, pinned(mod, (size, cols, n), np.float64))

    def one():
        ad = ad_p if use_pinned else mod.vec_znx_dft_alloc(cols, size)
        for c in range(cols):
            mod.vec_znx_dft_apply(1, 0, ad, c, a, c)
        rd = rd_p if use_pinned else mod.vec_znx_dft_alloc(cols, size)
        mod._ck(mod.lib.pz_vmp_apply_dft_to_dft(mod.handle, rd.data.ctypes.data_as(C.c_void_p), C.c_size_t(cols), C.c_size_t(size),
                                                ad.data.ctypes.data_as(C.c_void_p), C.c_size_t(cols), C.c_size_t(size), d_key.ptr,
                                                C.c_size_t(dnum), C.c_size_t(cols), C.c_size_t(cols), C.c_size_t(size), C.c_size_t(0)))
        big = mod.vec_znx_idft_apply_consume(rd)
        for c in range(cols):
            mod.vec_znx_big_normalize(res, k, 0, c, big, k, c)

    one()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"metric": "external products/s, per-op host-pointer path (PCIe-inclusive, %s host memory, 1 ciphertext per call)" % ("pinned" if use_pinned else "pageable"),
                      "value": 1.0 / dt, "ms_per_product": dt * 1e3, "n": n, "limbs": size}))

    # the CoreImpl-level call on HOST containers (what the Rust shim's `core-fused` override issues): ciphertexts staged, the
    # host-resident prepared key mirrored on the device at first use; 1 ciphertext and 16 ciphertexts per call
    from poulpy_amd.hal import GlweOpParams
    p = GlweOpParams(rank=cols - 1, dnum=dnum, dsize=1, key_size=size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k,
                     rank_out=cols - 1)
    key_host = pinned(mod, pm.data.shape, np.float64) if use_pinned else pm.data
    if use_pinned:
        key_host[...] = pm.data
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)
    for batch in (1, 16):
        shape = (batch, size, cols, n)
        ab = pinned(mod, shape, np.int64) if use_pinned else np.empty(shape, dtype=np.int64)
        rb = pinned(mod, shape, np.int64) if use_pinned else np.empty(shape, dtype=np.int64)
        ab[...] = a.data
        mod.glwe_external_product_batched(hp(rb), hp(ab), hp(key_host), p, batch)      # uploads the key mirror
        assert np.array_equal(rb[0], res.data)
        for _ in range(10):                                                              # steady state (the first calls of a fresh process also pay the
            mod.glwe_external_product_batched(hp(rb), hp(ab), hp(key_host), p, batch)    # driver's first mapping of the pinned pages: 3.9k vs 4.9k/s)
        t0 = time.perf_counter()
        for _ in range(reps):
            mod.glwe_external_product_batched(hp(rb), hp(ab), hp(key_host), p, batch)
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"metric": "external products/s, fused GLWE-level call on host containers (PCIe-inclusive, %s host memory, "
                                    "%d ciphertext(s) per call, host key mirrored)" % ("pinned" if use_pinned else "pageable", batch),
                          "value": batch / dt, "ms_per_call": dt * 1e3, "n": n, "limbs": size}))


if __name__ == "__main__":
    main()
