#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference (Rust) cannot run in this image and ships no golden vectors, so the committed
vectors are produced by exact integer arithmetic only (oracle/exact.py: schoolbook negacyclic
products in Python big ints + big-int digit decomposition) — no FFT, no floating point, nothing
from oracle/fft64_ref.c or the GPU path is involved.  They state what poulpy-cpu-ref must
output whenever its f64 rounding error stays below 1/2 (the property its own FFT64-vs-NTT120
test relies on, poulpy-cpu-ref/src/tests.rs:133-141).

    python tests/golden/make_golden.py          # rewrites the fixtures (deterministic seeds)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import exact  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def uniform(rng, log_bound, shape):
    h = 1 << (log_bound - 1)
    return rng.integers(-h, h, shape, dtype=np.int64)


def gen_vmp(name, seed, n, base2k, rows, cols_in, cols_out, size, a_size, res_size, limb_offset):
    rng = np.random.default_rng(seed)
    a = uniform(rng, base2k, (a_size, cols_in, n))
    mat = uniform(rng, base2k, (rows, cols_in, size, cols_out, n))
    big = exact.vmp_exact(a, mat, limb_offset, res_size)
    norm = np.zeros((res_size, cols_out, n), dtype=np.int64)
    for c in range(cols_out):
        norm[:, c, :] = exact.normalize_exact(big[:, c, :], base2k, res_size)
    np.savez_compressed(os.path.join(HERE, name), kind="vmp", n=n, base2k=base2k, limb_offset=limb_offset, a=a, mat=mat,
                        res_big=big.astype(np.int64), res_norm=norm)


def gen_svp(name, seed, n, base2k, cols, size):
    rng = np.random.default_rng(seed)
    s = uniform(rng, base2k, (1, cols, n))
    b = uniform(rng, base2k, (size, cols, n))
    big = np.zeros((size, cols, n), dtype=np.int64)
    for c in range(cols):
        for j in range(size):
            big[j, c] = exact.negacyclic_mul(s[0, c], b[j, c]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, name), kind="svp", n=n, base2k=base2k, s=s, b=b, res_big=big)


def gen_normalize(name, seed, n, base2k, a_size, res_size, log_bound):
    rng = np.random.default_rng(seed)
    a = uniform(rng, log_bound, (a_size, 1, n))
    res = exact.normalize_exact(a[:, 0, :], base2k, res_size)[:, None, :]
    np.savez_compressed(os.path.join(HERE, name), kind="normalize", n=n, base2k=base2k, a=a, res=res)


def gen_external_product(name, seed, n, base2k, rank, a_size, dnum, key_size, res_size):
    """dsize = 1 GLWE (x) GGSW: res_big = vmp(a, ggsw) ; res = normalize(res_big) per column
    (poulpy-core/src/external_product/glwe.rs:99-141,197-271)."""
    rng = np.random.default_rng(seed)
    cols = rank + 1
    a = uniform(rng, base2k, (a_size, cols, n))
    mat = uniform(rng, base2k, (dnum, cols, key_size, cols, n))
    big = exact.vmp_exact(a, mat, 0, key_size)
    res = np.zeros((res_size, cols, n), dtype=np.int64)
    for c in range(cols):
        res[:, c, :] = exact.normalize_exact(big[:, c, :], base2k, res_size)
    np.savez_compressed(os.path.join(HERE, name), kind="external_product", n=n, base2k=base2k, rank=rank, a=a, mat=mat, res=res)


def gen_glwe_automorphism(name, seed, n, base2k, rank, a_size, dnum, key_size, res_size, p):
    """dsize = 1 glwe_automorphism / glwe_automorphism_add (poulpy-core/src/automorphism/glwe_ct.rs:51-72, :96-140):
    big = mask x key + body (keyswitching/glwe.rs:207-239); res_auto = phi(normalize(big)); res_add = normalize(phi(big) + a)."""
    rng = np.random.default_rng(seed)
    cols = rank + 1
    a = uniform(rng, base2k, (a_size, cols, n))
    mat = uniform(rng, base2k, (dnum, rank, key_size, cols, n))
    big = exact.vmp_exact(np.ascontiguousarray(a[:, 1:, :]), mat, 0, key_size)
    big[:a_size, 0, :] += a[:, 0, :].astype(object)
    res_auto = np.zeros((res_size, cols, n), dtype=np.int64)
    res_add = np.zeros((res_size, cols, n), dtype=np.int64)
    for c in range(cols):
        res_auto[:, c, :] = exact.automorphism_exact(exact.normalize_exact(big[:, c, :], base2k, res_size), p)
        v = exact.automorphism_exact(big[:, c, :], p)
        v[:a_size] += a[:, c, :].astype(object)
        res_add[:, c, :] = exact.normalize_exact(v, base2k, res_size)
    np.savez_compressed(os.path.join(HERE, name), kind="glwe_automorphism", n=n, base2k=base2k, rank=rank, p=p, a=a, mat=mat,
                        res_auto=res_auto, res_add=res_add)


def main():
    # shape grid of poulpy-hal/src/test_suite/vmp.rs (sizes 1..4, cols 1..2, limb_offset) at small N
    gen_vmp("vmp_n32_b12.npz", 1, 32, 12, rows=3, cols_in=2, cols_out=2, size=4, a_size=3, res_size=4, limb_offset=0)
    gen_vmp("vmp_n64_b17_off1.npz", 2, 64, 17, rows=2, cols_in=1, cols_out=2, size=3, a_size=2, res_size=3, limb_offset=1)
    gen_vmp("vmp_n256_b19.npz", 3, 256, 19, rows=4, cols_in=2, cols_out=1, size=2, a_size=4, res_size=2, limb_offset=0)
    gen_svp("svp_n64_b17.npz", 4, 64, 17, cols=2, size=3)
    gen_normalize("normalize_n64_b12.npz", 5, 64, 12, a_size=5, res_size=3, log_bound=50)
    gen_normalize("normalize_n64_b19.npz", 6, 64, 19, a_size=2, res_size=4, log_bound=58)
    # config 1 of BASELINE.json: N=2^10, 2 limbs (DFT + SVP plumbing)
    gen_svp("config1_svp_n1024_b17.npz", 7, 1024, 17, cols=2, size=2)
    gen_external_product("extprod_n256_b12_rank1.npz", 8, 256, 12, rank=1, a_size=4, dnum=4, key_size=4, res_size=4)
    gen_external_product("extprod_n128_b14_rank2.npz", 9, 128, 14, rank=2, a_size=3, dnum=3, key_size=4, res_size=3)
    gen_glwe_automorphism("glwe_automorphism_n128_b13_rank1.npz", 10, 128, 13, rank=1, a_size=3, dnum=3, key_size=4, res_size=4, p=-5)
    gen_glwe_automorphism("glwe_automorphism_n64_b12_rank2.npz", 11, 64, 12, rank=2, a_size=4, dnum=4, key_size=4, res_size=3, p=25)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
