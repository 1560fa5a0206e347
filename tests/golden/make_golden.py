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


def gen_glwe_batched(name, seed, n, base2k, rank, a_size, dnum, key_size, res_size, batch, keyswitch):
    """A batch of dsize = 1 GLWE (x) GGSW external products or GLWE key switches sharing one key — what the batched device entry
    points (pz_glwe_external_product_batched / pz_glwe_keyswitch_batched: the fused pipelines) must return.  Key switch: big = mask x key
    + body on column 0 (keyswitching/glwe.rs:207-239)."""
    rng = np.random.default_rng(seed)
    cols = rank + 1
    cols_in = rank if keyswitch else cols
    a = uniform(rng, base2k, (batch, a_size, cols, n))
    mat = uniform(rng, base2k, (dnum, cols_in, key_size, cols, n))
    res = np.zeros((batch, res_size, cols, n), dtype=np.int64)
    for b in range(batch):
        big = exact.vmp_exact(np.ascontiguousarray(a[b][:, (1 if keyswitch else 0):, :]), mat, 0, key_size)
        if keyswitch:
            big[:min(a_size, key_size), 0, :] += a[b][:min(a_size, key_size), 0, :].astype(object)
        for c in range(cols):
            res[b, :, c, :] = exact.normalize_exact(big[:, c, :], base2k, res_size)
    np.savez_compressed(os.path.join(HERE, name), kind="glwe_batched", n=n, base2k=base2k, rank=rank, keyswitch=int(keyswitch), a=a, mat=mat, res=res)


def gen_lwe_keyswitch(name, seed, n, base2k, n_lwe_in, n_lwe_out, size, dnum, key_size, batch, n2):
    """mod_switch_2n (poulpy-bin-fhe blind_rotation/algorithms/mod.rs:136-176, the one-limb branch: round(limb0 * n2 / 2^base2k), half up)
    and lwe_keyswitch (poulpy-core keyswitching/lwe.rs:49-94): embed b -> X^0 of column 0 and a_i -> the first coefficients of column 1,
    rank-1 key switch on exact integers, extract."""
    rng = np.random.default_rng(seed)
    lwe = uniform(rng, base2k, (batch, size, n_lwe_in + 1))
    mat = uniform(rng, base2k, (dnum, 1, key_size, 2, n))
    log2n = (n2 - 1).bit_length() + 1
    assert base2k > log2n
    diff = base2k - (log2n - 1)
    ms = (lwe[:, 0, :] + (1 << (diff - 1))) >> diff
    out = np.zeros((batch, size, n_lwe_out + 1), dtype=np.int64)
    for b in range(batch):
        glwe = np.zeros((size, 2, n), dtype=np.int64)
        glwe[:, 0, 0] = lwe[b, :, 0]
        glwe[:, 1, :n_lwe_in] = lwe[b, :, 1:]
        big = exact.vmp_exact(np.ascontiguousarray(glwe[:, 1:, :]), mat, 0, key_size)
        big[:min(size, key_size), 0, :] += glwe[:min(size, key_size), 0, :].astype(object)
        for c in range(2):
            d = exact.normalize_exact(big[:, c, :], base2k, size)
            if c == 0:
                out[b, :, 0] = d[:, 0]
            else:
                out[b, :, 1:] = d[:, :n_lwe_out]
    np.savez_compressed(os.path.join(HERE, name), kind="lwe_keyswitch", n=n, base2k=base2k, n2=n2, lwe=lwe, mat=mat, mod_switched=ms, res=out)


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
    # the batched GLWE entry points at ring degrees with a device pipeline of their own (two-kernel: N = 1024; three-kernel: N = 8192)
    gen_glwe_batched("glwe_batched_extprod_n1024_b17.npz", 12, 1024, 17, rank=1, a_size=3, dnum=3, key_size=3, res_size=3, batch=3, keyswitch=False)
    gen_glwe_batched("glwe_batched_keyswitch_n1024_b14.npz", 13, 1024, 14, rank=1, a_size=2, dnum=2, key_size=3, res_size=2, batch=2, keyswitch=True)
    gen_glwe_batched("glwe_batched_extprod_n4096_b12.npz", 14, 4096, 12, rank=1, a_size=2, dnum=2, key_size=2, res_size=2, batch=2, keyswitch=False)
    gen_lwe_keyswitch("lwe_keyswitch_n256_b17.npz", 15, 256, 17, n_lwe_in=100, n_lwe_out=77, size=2, dnum=2, key_size=3, batch=3, n2=512)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
