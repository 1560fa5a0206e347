"""ORACLE — TEST INFRASTRUCTURE ONLY.  Exact-integer (Python big-int / object arrays) statements
of what the FFT64 path must compute, independent of any FFT.  These pin the C restatement
(oracle/fft64_ref.c) and generate tests/golden/*.npz (tests/golden/make_golden.py).

P1  IDFT(VMP(DFT(a), prepare(M))) rounded == the exact bivariate negacyclic product
    sum_r a_r * M[r, c] mod (X^N + 1)        (what FFT64Ref == NTT120Ref demonstrates,
                                              poulpy-cpu-ref/src/tests.rs:133-141)
P3  normalize preserves the torus value sum_j limb_j * 2^{-(j+1)k} (mod 1) and yields
    balanced digits                           (reference/vec_znx/normalize.rs:428-540)
"""
from __future__ import annotations

import numpy as np


def negacyclic_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Exact product in Z[X]/(X^N+1); inputs int64 arrays, output Python-int object array."""
    n = len(a)
    ao = a.astype(object)
    bo = b.astype(object)
    full = np.zeros(2 * n, dtype=object)
    for i in range(n):
        if ao[i] != 0:
            full[i:i + n] += ao[i] * bo
    return full[:n] - full[n:]


def negacyclic_mul_fast(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Same as negacyclic_mul via exact int64 numpy convolution when it cannot overflow
    (|a|,|b| < 2^20 and N <= 2^16 keeps every partial sum below 2^57)."""
    n = len(a)
    assert np.abs(a).max(initial=0) < (1 << 20) and np.abs(b).max(initial=0) < (1 << 20) and n <= (1 << 16)
    full = np.zeros(2 * n, dtype=np.int64)
    for i in range(n):
        if a[i] != 0:
            full[i:i + n] += a[i] * b
    return full[:n] - full[n:]


def vmp_exact(a: np.ndarray, mat: np.ndarray, limb_offset: int = 0, res_size: int | None = None) -> np.ndarray:
    """Exact res_big of vmp_apply: a (a_size, cols_in, n), mat (rows, cols_in, size, cols_out, n)
    -> (res_size, cols_out, n) object array; row index = limb_in."""
    a_size, cols_in, n = a.shape
    rows, _, size, cols_out, _ = mat.shape
    if res_size is None:
        res_size = size
    out = np.zeros((res_size, cols_out, n), dtype=object)
    for limb_out in range(res_size):
        src = limb_out + limb_offset
        if src >= size:
            continue
        for co in range(cols_out):
            acc = np.zeros(n, dtype=object)
            for limb_in in range(min(a_size, rows)):
                for ci in range(cols_in):
                    acc = acc + negacyclic_mul_fast(a[limb_in, ci], mat[limb_in, ci, src, co]).astype(object)
            out[limb_out, co] = acc
    return out


def torus_value(limbs, base2k: int, total_bits: int):
    """sum_j limb_j * 2^{total_bits-(j+1)k} as exact integers (numerator over 2^total_bits)."""
    size = limbs.shape[0]
    acc = np.zeros(limbs.shape[1:], dtype=object)
    for j in range(size):
        sh = total_bits - (j + 1) * base2k
        lj = limbs[j].astype(object)
        if sh >= 0:
            acc = acc + lj * (1 << sh)
        else:
            raise ValueError("total_bits too small")
    return acc


def normalize_exact(a_limbs: np.ndarray, base2k: int, res_size: int) -> np.ndarray:
    """Same-base, res_offset = 0 normalize as a big-int digit decomposition.

    The value sum_j a_j * 2^{(a_size-1-j)k} is split from the least significant limb into
    balanced digits in [-2^(k-1), 2^(k-1)); limbs of `a` beyond res_size only contribute
    their carry, limbs of res beyond a_size are zero, the carry out of limb 0 is dropped
    (torus, mod 1).  This is what the reference's carry chain computes when no i64 overflow
    occurs (normalize.rs:50-144 with lsh = 0), stated without any step function."""
    a_size = a_limbs.shape[0]
    k = base2k
    half = 1 << (k - 1)
    v = np.zeros(a_limbs.shape[1:], dtype=object)
    for j in range(a_size):
        v = v * (1 << k) + a_limbs[j].astype(object)
    out = np.zeros((res_size,) + a_limbs.shape[1:], dtype=np.int64)
    for j in range(a_size - 1, -1, -1):
        d = v % (1 << k)
        d = np.where(d >= half, d - (1 << k), d)
        v = (v - d) >> k
        if j < res_size:
            out[j] = d.astype(np.int64)
    return out


def torus_equal(a_limbs: np.ndarray, a_base2k: int, res_limbs: np.ndarray, res_base2k: int, res_offset: int = 0,
                slack_bits: int = 1) -> bool:
    """Criterion of test_vec_znx_normalize_cross_base2k (reference/vec_znx/normalize.rs:428-540):
    with want = value(a) * 2^res_offset and have = value(res), both reduced mod 1,
    |have - want| <= 2^(-min_prec + slack_bits), min_prec = min(a bits, res bits).  Exact big-int
    arithmetic (the reference uses 128-bit floats)."""
    a_bits = a_limbs.shape[0] * a_base2k
    r_bits = res_limbs.shape[0] * res_base2k
    min_prec = min(a_bits, r_bits)
    tot = max(a_bits, r_bits) + abs(res_offset) + 4
    va = torus_value(a_limbs, a_base2k, tot)      # numerators over 2^tot
    vr = torus_value(res_limbs, res_base2k, tot)
    if res_offset >= 0:
        va = va * (1 << res_offset)
    else:
        va = va >> (-res_offset)                   # floor; error < 2^-tot
    mod = 1 << tot
    diff = (va - vr) % mod
    diff = np.where(diff > mod // 2, mod - diff, diff)
    bound = (1 << (tot - min_prec + slack_bits)) + 1
    return bool(np.all(diff <= bound))


def automorphism_exact(a: np.ndarray, p: int) -> np.ndarray:
    """X -> X^p on the last axis (length n), stated directly: coefficient i moves to i*p mod 2n and changes sign when
    that index is >= n (X^n = -1).  p odd."""
    a = np.asarray(a)
    n = a.shape[-1]
    out = np.zeros_like(a)
    for i in range(n):
        k = (i * p) % (2 * n)
        if k < n:
            out[..., k] = a[..., i]
        else:
            out[..., k - n] = -a[..., i]
    return out


def rotate_exact(a: np.ndarray, p: int) -> np.ndarray:
    """X^p * a on the last axis, stated directly: coefficient i moves to (i + p) mod 2n, negated when >= n."""
    a = np.asarray(a)
    n = a.shape[-1]
    out = np.zeros_like(a)
    for i in range(n):
        k = (i + p) % (2 * n)
        if k < n:
            out[..., k] = a[..., i]
        else:
            out[..., k - n] = -a[..., i]
    return out
