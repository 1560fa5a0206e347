"""Pins of the oracle's convolution / tensoring restatement (oracle/fft64_ref.c) on exact integers.

The reference's own tests define the expected result of the bivariate convolution with a naive schoolbook statement
(poulpy-hal/src/test_suite/convolution.rs:254-302, `bivariate_convolution_naive`): limb `a_limb + b_limb + 1 - k` of the result
accumulates the negacyclic product of limb a_limb of a and limb b_limb of b, k = cnv_offset + 1, and the sum is normalized
in place.  Restated here on Python big integers (oracle/exact.py), independent of any FFT:

P13  cnv_apply_dft(offset)          -> idft -> normalize  ==  naive bivariate convolution            (test_convolution, :91-157)
P14  cnv_pairwise_apply_dft(i, j)   -> idft -> normalize  ==  naive convolution of (a_i + a_j), (b_i + b_j)   (:159-252)
P15  cnv_by_const_apply(offset)     -> normalize          ==  naive convolution with a constant polynomial    (:22-89)
P16  cnv_prepare_*: mask only touches the last active limb; prepare_self == (prepare_left, prepare_right)
P17  glwe_tensor_apply / _square / _add_assign: every column of the tensor carries the exact product value
     sum a_i b_j 2^(cnv_offset) (mod 1) up to the rounding of its three truncating normalizations
P18  glwe_tensor_relinearize == key switch of the pair columns by the tensor key + the GLWE part (exact statement of
     operations/glwe.rs:541-607 built from the pinned pieces P1 / P3)
"""
import numpy as np
import pytest

from oracle import exact
from oracle.ref import RefModule
from poulpy_amd.layouts import CnvPVecL, CnvPVecR, MatZnx, VecZnx, VecZnxBig, VecZnxDft
from tests.helpers import seeded


def naive_bivariate(n, base2k, k, res_size, a_limbs, b_limbs):
    """bivariate_convolution_naive (test_suite/convolution.rs:254-302) before its final normalize_assign: object limbs."""
    res = np.zeros((res_size, n), dtype=object)
    for al in range(a_limbs.shape[0]):
        for bl in range(b_limbs.shape[0]):
            rl = al + bl + 1
            if k <= 0:
                rl += abs(k)
            elif rl >= k:
                rl -= k
            else:
                continue
            if rl < res_size:
                res[rl] = res[rl] + exact.negacyclic_mul_fast(a_limbs[al], b_limbs[bl]).astype(object)
    return res


def normalize_assign_exact(limbs, base2k):
    return exact.normalize_exact(limbs, base2k, limbs.shape[0])


@pytest.mark.parametrize("n,a_size,b_size,base2k", [(32, 5, 4, 12), (64, 3, 6, 17), (16, 15, 15, 12)])
def test_p13_cnv_apply_dft_is_the_naive_bivariate_convolution(n, a_size, b_size, base2k):
    ref = RefModule(n)
    rng = seeded(n * 100 + a_size)
    cols = 2
    res_size = a_size + b_size
    a = VecZnx(n, cols, a_size).fill_uniform(17, rng)
    b = VecZnx(n, cols, b_size).fill_uniform(17, rng)
    ap, bp = CnvPVecL(n, cols, a_size), CnvPVecR(n, cols, b_size)
    ref.cnv_prepare_left(ap, a)
    ref.cnv_prepare_right(bp, b)
    for a_col in range(cols):
        for b_col in range(cols):
            for off in range(0, res_size, 1 if n <= 32 else 3):
                d = VecZnxDft(n, 1, res_size)
                d.data[...] = 123.0
                ref.cnv_apply_dft(off, d, 0, ap, a_col, bp, b_col)
                big = VecZnxBig(n, 1, res_size)
                ref.vec_znx_idft_apply_tmpa(big, 0, d, 0)
                have = VecZnx(n, 1, res_size)
                ref.vec_znx_big_normalize(have, base2k, 0, 0, big, base2k, 0)
                want = normalize_assign_exact(naive_bivariate(n, base2k, off + 1, res_size, a.data[:, a_col], b.data[:, b_col]), base2k)
                assert np.array_equal(have.data[:, 0], want), (a_col, b_col, off)


@pytest.mark.parametrize("n,a_size,b_size,res_size", [(32, 4, 4, 8), (32, 5, 3, 4), (16, 2, 6, 11)])
def test_p14_pairwise_and_shorter_results(n, a_size, b_size, res_size):
    """(a_i + a_j) x (b_i + b_j); res_size below / above a_size + b_size - 1 (limbs beyond the bound are zero, limbs beyond
    res_size are dropped)."""
    ref = RefModule(n)
    rng = seeded(n + 7 * a_size + res_size)
    cols, base2k = 3, 14
    a = VecZnx(n, cols, a_size).fill_uniform(15, rng)
    b = VecZnx(n, cols, b_size).fill_uniform(15, rng)
    ap, bp = CnvPVecL(n, cols, a_size), CnvPVecR(n, cols, b_size)
    ref.cnv_prepare_left(ap, a)
    ref.cnv_prepare_right(bp, b)
    for i in range(cols):
        for j in range(cols):
            for off in (0, 1, 2, a_size + b_size - 2, a_size + b_size + 3):
                d = VecZnxDft(n, 1, res_size)
                ref.cnv_pairwise_apply_dft(off, d, 0, ap, bp, i, j)
                big = VecZnxBig(n, 1, res_size)
                ref.vec_znx_idft_apply_tmpa(big, 0, d, 0)
                ta = a.data[:, i] + (a.data[:, j] if i != j else 0)
                tb = b.data[:, i] + (b.data[:, j] if i != j else 0)
                # the raw (un-normalized) limbs must already be the exact sums: res limb kk = sum_{al + bl = kk + min(off, bound)}
                bound = a_size + b_size - 1
                full = naive_bivariate(n, base2k, min(off, bound) + 1, max(res_size, 1), ta, tb)
                # naive indexes limbs by al + bl + 1 - k = al + bl - off: identical to the reference's kk
                assert np.array_equal(big.data[:, 0].astype(object), full), (i, j, off)


@pytest.mark.parametrize("n,a_size,b_size", [(32, 6, 4), (64, 15, 15)])
def test_p15_cnv_by_const(n, a_size, b_size):
    ref = RefModule(n)
    rng = seeded(n + a_size)
    res_size = a_size + b_size
    base2k = 12
    a = VecZnx(n, 2, a_size).fill_uniform(17, rng)
    b_const = rng.integers(-(1 << 16), 1 << 16, b_size, dtype=np.int64)
    b_poly = np.zeros((b_size, n), dtype=np.int64)
    b_poly[:, 0] = b_const                               # constant polynomials (convolution.rs:47-53 of the test)
    for a_col in range(2):
        for off in range(0, res_size, 2):
            big = VecZnxBig(n, 2, res_size)
            big.data[...] = 99
            ref.cnv_by_const_apply(off, big, 1, a, a_col, b_const)
            assert (big.data[:, 0] == 99).all()           # only res_col is written
            want = naive_bivariate(n, base2k, off + 1, res_size, a.data[:, a_col], b_poly)
            assert np.array_equal(big.data[:, 1].astype(object), want), (a_col, off)


def test_p16_prepare_mask_and_self():
    n, cols, a_size = 64, 2, 5
    ref = RefModule(n)
    rng = seeded(16)
    a = VecZnx(n, cols, a_size).fill_uniform(20, rng)
    mask = ref.msb_mask_bottom_limb(12, 12 * a_size - 5)   # clears the 5 low bits of the last limb
    assert mask == -(1 << 5) and ref.msb_mask_bottom_limb(12, 24) == -1
    for res_size in (3, 5, 7):
        pl, pr = CnvPVecL(n, cols, res_size), CnvPVecR(n, cols, res_size)
        ref.cnv_prepare_left(pl, a, mask)
        ref.cnv_prepare_right(pr, a, mask)
        assert np.array_equal(pl.data, pr.data)
        sl, sr = CnvPVecL(n, cols, res_size), CnvPVecR(n, cols, res_size)
        ref.cnv_prepare_self(sl, sr, a, mask)
        assert np.array_equal(sl.data, pl.data) and np.array_equal(sr.data, pl.data)
        # the same as preparing a copy of `a` whose last ACTIVE limb is masked by hand (and nothing else)
        am = a.copy()
        last = min(res_size, a_size) - 1
        am.data[last] &= mask
        pm = CnvPVecL(n, cols, res_size)
        ref.cnv_prepare_left(pm, am, -1)
        assert np.array_equal(pm.data, pl.data)


def _torus(limbs, base2k, tot):
    return exact.torus_value(limbs, base2k, tot)


@pytest.mark.parametrize("rank,square,add_assign", [(1, False, False), (1, True, False), (2, False, False), (1, False, True), (2, True, False)])
@pytest.mark.parametrize("cnv_offset", [5, 12, 30])
def test_p17_glwe_tensor_columns_carry_the_exact_products(rank, square, add_assign, cnv_offset):
    """Column (i, j) of the tensor = normalized (sum over the pair) a_i * b_j scaled by 2^cnv_offset relative to the product of the two
    torus values (operations/glwe.rs:700-807).  The reference normalizes each product term separately into res_size limbs and
    combines digits afterwards, so the check is on the torus value: |have - want| <= 4 units of the last res limb (three
    truncating normalizations per off-diagonal column, one per diagonal one), mod 1, where `want` sums exactly the product limbs the
    reference evaluates (it drops those below the precision of res: the tensor is an approximate, CKKS-style product)."""
    n, base2k, a_size, b_size, res_size = 32, 12, 4, 3, 5
    if square:
        b_size = a_size
    ref = RefModule(n)
    rng = seeded(17 + rank + 2 * square + cnv_offset)
    cols = rank + 1
    tcols = cols * (cols + 1) // 2
    a = VecZnx(n, cols, a_size).fill_uniform(base2k, rng)
    b = a if square else VecZnx(n, cols, b_size).fill_uniform(base2k, rng)
    a_k, b_k = base2k * a_size - 3, base2k * b_size            # a: 3 masked low bits
    if square:
        b_k = a_k
    res = VecZnx(n, tcols, res_size)
    prev = None
    if add_assign:
        res.fill_uniform(base2k, rng)
        prev = res.data.copy()
    if square:
        ref.glwe_tensor_square_apply(cnv_offset, res, base2k, a, a_k, base2k)
    else:
        ref.glwe_tensor_apply(cnv_offset, res, base2k, a, a_k, b, b_k, base2k, add_assign=add_assign)
    am = a.data.copy()
    am[a_size - 1] &= ref.msb_mask_bottom_limb(base2k, a_k)
    bm = am if square else b.data.copy()
    # (cnv_offset_hi, cnv_offset_lo) and the number of product limbs the reference evaluates (operations/glwe.rs:733-760)
    if cnv_offset < base2k:
        hi, lo = 0, -(base2k - cnv_offset % base2k)
    else:
        hi, lo = max(cnv_offset // base2k - 1, 0), cnv_offset % base2k
    dft_size = min(a_size + b_size - hi, -(-(res_size * base2k + (lo % base2k)) // base2k))
    tot = base2k * (a_size + b_size + res_size) + cnv_offset + 8
    mod = 1 << tot
    ulp = 1 << (tot - base2k * res_size)
    for i in range(cols):
        for j in range(i, cols):
            col = i * cols - (i * (i + 1) // 2) + j
            want = np.zeros(n, dtype=object)
            for (x, y) in ([(i, i)] if i == j else [(i, j), (j, i)]):
                for al in range(a_size):
                    for bl in range(b_size):
                        if al + bl - hi >= dft_size:
                            continue       # product limbs the reference does not compute (below the precision of res, :748-760)
                        # value(a limb al) * value(b limb bl) = prod * 2^-((al+1) + (bl+1)) k, times 2^cnv_offset
                        sh = tot - (al + bl + 2) * base2k + cnv_offset
                        prod = exact.negacyclic_mul_fast(am[al, x], bm[bl, y]).astype(object)
                        want = want + (prod * (1 << sh) if sh >= 0 else prod >> (-sh))
            have = _torus(res.data[:, col], base2k, tot)
            if add_assign:
                have = have - _torus(prev[:, col], base2k, tot)
            diff = (have - want) % mod
            diff = np.where(diff > mod // 2, mod - diff, diff)
            assert np.all(diff <= 4 * ulp), (i, j, int(max(diff)) / ulp)


@pytest.mark.parametrize("rank,dsize,bases", [(1, 1, (12, 12, 12)), (2, 1, (13, 13, 13)), (1, 2, (12, 12, 12)), (1, 1, (15, 12, 13))])
def test_p18_relinearize_is_keyswitch_of_the_pair_columns_plus_the_glwe_part(rank, dsize, bases):
    """operations/glwe.rs:541-607 stated with the pinned primitives: big = sum_pairs a[cols + p] (x) tsk row p  (exact bivariate product,
    P1) + a[0..cols], then same-base normalize (P3).  Equal bases and dsize = 1 are checked exactly; the other cases on the torus
    value (cross-base normalizations round)."""
    n, a_size, key_size, res_size, dnum = 32, 4, 5, 4, 4
    a_b, k_b, r_b = bases
    ref = RefModule(n)
    rng = seeded(18 + rank + dsize + a_b)
    cols, pairs = rank + 1, rank * (rank + 1) // 2
    a = VecZnx(n, cols + pairs, a_size).fill_uniform(a_b, rng)
    mat = MatZnx(n, dnum, pairs, cols, key_size).fill_uniform(k_b, rng)
    pm = ref.vmp_pmat_alloc(dnum, pairs, cols, key_size)
    ref.vmp_prepare(pm, mat)
    res = VecZnx(n, cols, res_size)
    res.data[...] = 77
    ref.glwe_tensor_relinearize(res, r_b, a, a_b, pm, dsize, k_b)
    if a_b == k_b == r_b and dsize == 1:
        big = exact.vmp_exact(np.ascontiguousarray(a.data[:, cols:]), mat.data)            # (key_size, cols, n) object
        for c in range(cols):
            for l in range(min(a_size, key_size)):
                big[l, c] = big[l, c] + a.data[l, c].astype(object)
            want = exact.normalize_exact(big[:, c], k_b, res_size)
            assert np.array_equal(res.data[:, c], want), c
    else:
        # the same pipeline from the per-op oracle entry points (each pinned on its own): DFT of the pair columns, product, idft, + a, normalize
        ks = VecZnx(n, cols, res_size)
        # a GLWE (rank_in = pairs) whose mask columns are the pair columns and whose body is zero gives the product part ...
        fake = VecZnx(n, pairs + 1, a_size)
        fake.data[:, 1:] = a.data[:, cols:]
        ref.glwe_keyswitch(ks, r_b, fake, a_b, pm, dsize, k_b)
        # ... so relinearize - keyswitch must be (a's GLWE part) on the torus, up to rounding of the cross-base normalizations
        tot = max(a_size * a_b, res_size * r_b) + 16
        mod = 1 << tot
        for c in range(cols):
            have = (_torus(res.data[:, c], r_b, tot) - _torus(ks.data[:, c], r_b, tot)) % mod
            want = _torus(a.data[:, c], a_b, tot) % mod
            diff = (have - want) % mod
            diff = np.where(diff > mod // 2, mod - diff, diff)
            assert np.all(diff <= (1 << (tot - min(res_size * r_b, a_size * a_b) + 3))), c
