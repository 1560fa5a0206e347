"""Structured (coherent) test inputs for the FFT64 path, and an FFT-independent exact product to judge them with.

The reference's tests - and until round 6 every margin figure of this repo - draw uniform digits (`fill_uniform`,
poulpy-hal/src/layouts/vec_znx.rs:283-295).  Uniform digits spread their energy over all frequency bins; a coherent input puts N 2^(k-1) into
ONE bin instead of sqrt(N) 2^(k-1), and the f64 error of what comes back scales with it.  These are the inputs an FFT likes least:

    min        every digit -2^(k-1)                       (the largest magnitude a balanced digit takes; all energy in the bins around DC)
    alt        +(2^(k-1) - 1), -2^(k-1), +, -, ...        (all energy at the Nyquist end)
    tone:f     round((2^(k-1) - 1) cos(2 pi f j / N)),  f in {1, N/4, N/2 - 1}
    delta      -2^(k-1) at j = 0, zero elsewhere          (flat spectrum: the easy extreme, as a control)

`exact_vmp_big(a, mat)` computes sum_r a_r * M[r, c] mod (X^N + 1) EXACTLY (7-bit pieces through numpy's complex FFT: every partial sum stays
below 2^31, five decimal orders under the 2^53 an f64 holds exactly, so the rounded convolution of the pieces is the integer convolution) - the
"big" value the GPU's and the oracle's inverse transforms must round to; `normalize_exact` (oracle/exact.py) then gives the digits.
Test infrastructure only (tools/margin.py --inputs structured, tests/test_gpu_structured.py)."""
from __future__ import annotations

import numpy as np

PATTERNS = ("min", "alt", "tone:1", "tone:N/4", "tone:N/2-1", "delta")


def pattern(name: str, n: int, base2k: int) -> np.ndarray:
    """One polynomial of n balanced base2k-bit digits following `name` (int64)."""
    half = 1 << (base2k - 1)
    j = np.arange(n, dtype=np.int64)
    if name == "min":
        return np.full(n, -half, dtype=np.int64)
    if name == "alt":
        return np.where(j % 2 == 0, half - 1, -half).astype(np.int64)
    if name == "delta":
        out = np.zeros(n, dtype=np.int64)
        out[0] = -half
        return out
    if name.startswith("tone:"):
        f = {"1": 1, "N/4": n // 4, "N/2-1": n // 2 - 1}[name.split(":", 1)[1]]
        return np.rint((half - 1) * np.cos(2.0 * np.pi * f * j.astype(np.float64) / n)).astype(np.int64)
    raise ValueError(f"unknown pattern {name!r}")


def fill(shape_prefix, name: str, n: int, base2k: int) -> np.ndarray:
    """An array [*shape_prefix, n] whose every polynomial is the pattern (every limb, column, row alike: the coherent worst case)."""
    return np.broadcast_to(pattern(name, n, base2k), tuple(shape_prefix) + (n,)).copy()


def _pieces(x: np.ndarray, bits: int, count: int):
    """x = sum_s piece_s 2^(bits s) with BALANCED pieces in [-2^(bits-1), 2^(bits-1)) (the last one takes what is left)."""
    out = []
    v = x.astype(np.int64).copy()
    half = 1 << (bits - 1)
    for s in range(count - 1):
        d = ((v + half) & ((1 << bits) - 1)) - half
        out.append(d)
        v = (v - d) >> bits
    out.append(v)
    return out


def exact_negacyclic_sum(a_rows: np.ndarray, b_rows: np.ndarray) -> np.ndarray:
    """sum_r a_rows[r] * b_rows[r] mod (X^n + 1), exact, as int64 (the caller keeps |result| < 2^62).  a_rows, b_rows: [rows][n] int64 digits
    of at most 21 bits."""
    rows, n = a_rows.shape
    bits, cnt = 7, 3
    assert np.abs(a_rows).max(initial=0) < (1 << 20) and np.abs(b_rows).max(initial=0) < (1 << 20)
    assert rows * n * (1 << (2 * bits - 2)) < (1 << 40), "piece products must stay far below 2^53"
    fa = [np.fft.rfft(np.stack(p).astype(np.float64), 2 * n, axis=-1) for p in zip(*[_pieces(a_rows[r], bits, cnt) for r in range(rows)])]
    fb = [np.fft.rfft(np.stack(p).astype(np.float64), 2 * n, axis=-1) for p in zip(*[_pieces(b_rows[r], bits, cnt) for r in range(rows)])]
    total = np.zeros(n, dtype=np.int64)
    for s in range(cnt):
        for t in range(cnt):
            full = np.fft.irfft((fa[s] * fb[t]).sum(axis=0), 2 * n)
            r = np.rint(full)
            assert np.abs(full - r).max() < 1e-3, "piece convolution not exact"
            c = r.astype(np.int64)
            total += (c[:n] - c[n:]) << (bits * (s + t))
    return total


def exact_vmp_big(a: np.ndarray, mat: np.ndarray) -> np.ndarray:
    """The exact VecZnxBig of vmp_apply (limb_offset 0, dsize 1): a [a_size][cols_in][n] digits, mat [rows][cols_in][out_size][cols_out][n]
    digits (MatZnx order: row r, input column ci -> VecZnx(cols_out, out_size)); rows = a_size.  Returns [out_size][cols_out][n] int64:
    big[l][c] = sum_{r, ci} a[r][ci] * mat[r][ci][l][c]  (reference/fft64/vmp.rs:144-264)."""
    a_size, cols_in, n = a.shape
    rows, ci2, out_size, cols_out, n2 = mat.shape
    assert ci2 == cols_in and n2 == n
    nr = min(rows, a_size)
    ar = a[:nr].reshape(nr * cols_in, n)
    out = np.empty((out_size, cols_out, n), dtype=np.int64)
    for l in range(out_size):
        for c in range(cols_out):
            out[l, c] = exact_negacyclic_sum(ar, mat[:nr, :, l, c].reshape(nr * cols_in, n))
    return out


def exact_external_product(a: np.ndarray, mat: np.ndarray, base2k: int) -> np.ndarray:
    """GLWE (x) GGSW with one base2k, dsize 1, res_size = a_size = key size: the digits the reference must produce when nothing rounds wrongly
    (external_product/glwe.rs:197-271: vmp -> idft -> normalize per column).  a [size][cols][n], mat as exact_vmp_big; returns [size][cols][n]."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle.exact import normalize_exact
    big = exact_vmp_big(a, mat)
    size, cols, n = a.shape
    res = np.empty((size, mat.shape[3], n), dtype=np.int64)
    for c in range(mat.shape[3]):
        res[:, c] = normalize_exact(big[:, c], base2k, size)
    return res
