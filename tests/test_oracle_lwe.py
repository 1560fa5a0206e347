"""CPU: pins of the LWE glue restated in oracle/fft64_ref.c (the gate-bootstrap path of BASELINE configs[3]:
mod_switch_2n -> blind rotation -> sample extract / LWE key switch).  The reference holds no vectors for these, so:
  P19  mod_switch_2n == the rounding / truncation it documents, stated on exact rationals and Python integers
       (poulpy-bin-fhe/src/blind_rotation/algorithms/mod.rs:136-176)
  P20  lwe_sample_extract and the LWE -> GLWE embedding are index statements (poulpy-core/src/api/conversion.rs:15-40)
  P21  lwe_keyswitch / glwe_from_lwe / lwe_from_glwe == embedding / rotation + the EXACT key-switch value (bivariate product of the
       mask with the key + body, big-int normalize) + extraction (keyswitching/lwe.rs:49-94, conversion/lwe_to_glwe.rs:46-121,
       conversion/glwe_to_lwe.rs:42-90); the cross-base embedding against an independent composition of the pinned normalize (P3)
"""
from fractions import Fraction

import numpy as np
import pytest

from oracle import exact
from oracle.ref import RefModule
from poulpy_amd.layouts import MatZnx, VecZnx
from tests.helpers import seeded


def rand_lwe(rng, size, n_lwe, base2k):
    return rng.integers(-(1 << (base2k - 1)), 1 << (base2k - 1), size=(size, n_lwe + 1), dtype=np.int64)


@pytest.mark.parametrize("n2", [64, 1024, 2048, 1 << 15])
@pytest.mark.parametrize("base2k", [5, 7, 12, 13, 17, 19])
@pytest.mark.parametrize("negate", [False, True])
def test_p19_mod_switch_2n(n2, base2k, negate):
    R = RefModule(64)
    rng = seeded(n2 + base2k)
    log2n = (n2 - 1).bit_length() + 1
    size = max(1, -(-log2n // base2k)) + 1
    lwe = rand_lwe(rng, size, 37, base2k)
    got = R.mod_switch_2n(n2, lwe, base2k, negate)
    s = -1 if negate else 1
    want = []
    for j in range(lwe.shape[1]):
        l = [int(v) for v in lwe[:, j]]
        if base2k > log2n:
            # one limb, rounded half up: round(s * l0 * n2 / 2^base2k)
            x = Fraction(s * l[0] * n2, 1 << base2k)
            want.append((x + Fraction(1, 2)).__floor__())
        else:
            # the first ceil(log2n / base2k) limbs concatenated and cut to log2n bits; only limb 0 carries the sign flip
            nl = -(-log2n // base2k)
            y = s * l[0]
            bits = base2k
            for i in range(1, nl):
                take = min(base2k, log2n - bits)
                y = (y << take) + (l[i] >> (base2k - take))
                bits += take
            want.append(y)
    assert [int(v) for v in got] == want


def test_p19_mod_switch_2n_maps_the_torus_to_z_2n():
    """for a normalized LWE (balanced digits) and base2k above log2(2n): the result is the nearest multiple of 1 / n2 of limb 0's value"""
    R = RefModule(64)
    rng = seeded(5)
    n2, base2k = 2048, 17
    lwe = rand_lwe(rng, 2, 100, base2k)
    got = R.mod_switch_2n(n2, lwe, base2k, False)
    x = lwe[0].astype(np.float64) / (1 << base2k)
    assert np.all(np.abs(got / n2 - x) <= 0.5 / n2 + 1e-12)
    assert got.min() >= -n2 // 2 and got.max() <= n2 // 2


@pytest.mark.parametrize("n,n_lwe", [(64, 20), (64, 63), (256, 64)])
def test_p20_sample_extract_is_an_index_statement(n, n_lwe):
    R = RefModule(n)
    rng = seeded(n + n_lwe)
    for a_size, res_size in ((3, 3), (4, 2), (2, 4)):
        a = VecZnx(n, 2, a_size).fill_uniform(12, rng)
        got = R.lwe_sample_extract(n_lwe, res_size, a)
        want = np.zeros((res_size, n_lwe + 1), dtype=np.int64)
        for i in range(min(a_size, res_size)):
            want[i, 0] = a.data[i, 0, 0]
            want[i, 1:] = a.data[i, 1, :n_lwe]
        assert np.array_equal(got, want)


def exact_keyswitch(a_data, mat_data, base2k, res_size):
    """rank_in -> rank_out key switch on exact integers: mask columns x key + body on column 0, big-int normalize (dsize 1, one base)"""
    key_size, cols_out, n = mat_data.shape[2], mat_data.shape[3], a_data.shape[2]
    big = exact.vmp_exact(a_data[:, 1:, :], mat_data)                       # (key_size, cols_out, n)
    for j in range(min(a_data.shape[0], key_size)):
        big[j, 0] = big[j, 0] + a_data[j, 0].astype(object)
    out = np.zeros((res_size, cols_out, n), dtype=np.int64)
    for c in range(cols_out):
        out[:, c, :] = exact.normalize_exact(big[:, c, :], base2k, res_size)
    return out


@pytest.mark.parametrize("n_lwe_in,n_lwe_out", [(20, 20), (31, 12), (12, 32)])
def test_p21_lwe_keyswitch_matches_exact(n_lwe_in, n_lwe_out):
    n, base2k, size, dnum, key_size = 32, 12, 3, 3, 4
    R = RefModule(n)
    rng = seeded(n_lwe_in * 64 + n_lwe_out)
    lwe = rand_lwe(rng, size, n_lwe_in, base2k)
    mat = MatZnx(n, dnum, 1, 2, key_size).fill_uniform(base2k, rng)
    pm = R.vmp_pmat_alloc(dnum, 1, 2, key_size)
    R.vmp_prepare(pm, mat)
    got = R.lwe_keyswitch(n_lwe_out, size, base2k, lwe, base2k, pm, 1, base2k)
    glwe = np.zeros((size, 2, n), dtype=np.int64)
    glwe[:, 0, 0] = lwe[:, 0]
    glwe[:, 1, :n_lwe_in] = lwe[:, 1:]
    out = exact_keyswitch(glwe, mat.data, base2k, size)
    want = np.zeros((size, n_lwe_out + 1), dtype=np.int64)
    want[:, 0] = out[:, 0, 0]
    want[:, 1:] = out[:, 1, :n_lwe_out]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("rank_out", [1, 2])
def test_p21_glwe_from_lwe_same_base_matches_exact(rank_out):
    n, base2k, size, dnum, key_size, n_lwe = 32, 13, 3, 3, 4, 17
    R = RefModule(n)
    rng = seeded(77 + rank_out)
    lwe = rand_lwe(rng, size, n_lwe, base2k)
    mat = MatZnx(n, dnum, 1, rank_out + 1, key_size).fill_uniform(base2k, rng)
    pm = R.vmp_pmat_alloc(dnum, 1, rank_out + 1, key_size)
    R.vmp_prepare(pm, mat)
    res = VecZnx(n, rank_out + 1, size)
    res.data[...] = 0x5A5A
    R.glwe_from_lwe(res, base2k, lwe, base2k, size, pm, 1, base2k)
    glwe = np.zeros((size, 2, n), dtype=np.int64)
    glwe[:, 0, 0] = lwe[:, 0]
    glwe[:, 1, :n_lwe] = lwe[:, 1:]
    assert np.array_equal(res.data, exact_keyswitch(glwe, mat.data, base2k, size))


def test_p21_glwe_from_lwe_cross_base_is_normalize_then_keyswitch():
    """lwe_to_glwe.rs:82-116: each column embedded into a one-column VecZnx, normalized to the key's base — composed here from the
    pinned normalize (P3) and key switch and compared with the C restatement"""
    n, lwe_base2k, key_base2k, lwe_size, n_lwe = 64, 17, 12, 2, 40
    glwe_size = -(-lwe_size * lwe_base2k // key_base2k)
    dnum, key_size, res_size = glwe_size, glwe_size + 1, glwe_size
    R = RefModule(n)
    rng = seeded(3)
    lwe = rand_lwe(rng, lwe_size, n_lwe, lwe_base2k)
    mat = MatZnx(n, dnum, 1, 2, key_size).fill_uniform(key_base2k, rng)
    pm = R.vmp_pmat_alloc(dnum, 1, 2, key_size)
    R.vmp_prepare(pm, mat)
    res = VecZnx(n, 2, res_size)
    R.glwe_from_lwe(res, key_base2k, lwe, lwe_base2k, glwe_size, pm, 1, key_base2k)
    glwe = VecZnx(n, 2, glwe_size)
    for col in range(2):
        conv = VecZnx(n, 1, lwe_size)
        if col == 0:
            conv.data[:, 0, 0] = lwe[:, 0]
        else:
            conv.data[:, 0, :n_lwe] = lwe[:, 1:]
        R.vec_znx_normalize(glwe, key_base2k, 0, col, conv, lwe_base2k, 0)
        assert exact.torus_equal(conv.data[:, 0, :], lwe_base2k, glwe.data[:, col, :], key_base2k)
    want = VecZnx(n, 2, res_size)
    R.glwe_keyswitch(want, key_base2k, glwe, key_base2k, pm, 1, key_base2k)
    assert np.array_equal(res.data, want.data)


@pytest.mark.parametrize("rank_in,a_idx", [(1, 0), (1, 5), (2, 0), (2, 31)])
def test_p21_lwe_from_glwe_matches_exact(rank_in, a_idx):
    n, base2k, size, dnum, key_size, n_lwe = 32, 12, 3, 3, 4, 20
    R = RefModule(n)
    rng = seeded(rank_in * 100 + a_idx)
    a = VecZnx(n, rank_in + 1, size).fill_uniform(base2k, rng)
    mat = MatZnx(n, dnum, rank_in, 2, key_size).fill_uniform(base2k, rng)
    pm = R.vmp_pmat_alloc(dnum, rank_in, 2, key_size)
    R.vmp_prepare(pm, mat)
    got = R.lwe_from_glwe(n_lwe, size, base2k, a, base2k, a_idx, pm, 1, base2k)
    rot = np.zeros_like(a.data)
    for l in range(size):
        for c in range(rank_in + 1):
            rot[l, c] = exact.rotate_exact(a.data[l, c], -a_idx)
    out = exact_keyswitch(rot, mat.data, base2k, size)
    want = np.zeros((size, n_lwe + 1), dtype=np.int64)
    want[:, 0] = out[:, 0, 0]
    want[:, 1:] = out[:, 1, :n_lwe]
    assert np.array_equal(got, want)
