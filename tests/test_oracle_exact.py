"""CPU: pin the C restatement (oracle/fft64_ref.c) with exact-integer statements (oracle/exact.py).

The reference ships no golden vectors and cannot be built here, so these
properties are what anchors the oracle:
  P1  idft(vmp(dft(a), prepare(M))) == exact bivariate negacyclic product       (tests.rs:133-141)
  P2  idft(svp(dft(b), prepare(s))) == exact negacyclic product                  (test_suite/svp.rs)
  P3  normalize == big-int balanced digit decomposition / torus value preserved  (normalize.rs:428-540)
  P5  automorphism family == X -> X^p stated directly on exact integers            (automorphism/glwe_ct.rs)
  P6  CGGI blind rotation (block-binary and standard) == the same recurrence on exact integers  (algorithm.rs:265-440)
  P7  vec_znx_rsh_assign (k <= base2k) == exact shift with round-to-nearest; glwe_trace == its recurrence on exact integers
  A1  spectral identity of fft_ref (slot k <-> root exp(2 pi i (4 bitrev(k)+1)/(4m)))
"""
import numpy as np
import pytest

from oracle import exact
from oracle.ref import RefModule
from poulpy_amd.layouts import MatZnx, ScalarZnx, SvpPPol, VecZnx, VecZnxBig, VecZnxDft
from tests.helpers import normalize_all, seeded


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64, 128, 256, 1024])
def test_fft_spectral_identity(n):
    R = RefModule(n)
    m = n // 2
    rng = seeded(n)
    a = rng.integers(-1000, 1000, n).astype(np.float64)
    d = a.copy()
    R.fft(d)
    z = a[:m] + 1j * a[m:]
    lg = m.bit_length() - 1
    rev = [int(format(k, "0%db" % lg)[::-1], 2) if lg else 0 for k in range(m)]
    want = np.array([np.sum(z * np.exp(2j * np.pi * np.arange(m) * (4 * rev[k] + 1) / (4 * m))) for k in range(m)])
    got = d[:m] + 1j * d[m:]
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("n", [4, 16, 64, 2048, 4096, 8192, 65536])
def test_fft_ifft_roundtrip(n):
    R = RefModule(n)
    rng = seeded(n + 1)
    a = rng.integers(-(1 << 40), 1 << 40, n).astype(np.float64)
    d = a.copy()
    R.fft(d)
    R.ifft(d)
    assert np.array_equal(np.round(d / (n // 2)), a)


@pytest.mark.parametrize("n,base2k", [(8, 12), (16, 17), (64, 12), (256, 19)])
def test_P1_vmp_equals_exact_product(n, base2k):
    """Shape grid of poulpy-hal/src/test_suite/vmp.rs:180-308 against the exact product."""
    R = RefModule(n)
    rng = seeded(n * 31 + base2k)
    for cols_in in (1, 2):
        for cols_out in (1, 2):
            for size_in in range(1, 4):
                for size_out in range(1, 4):
                    rows = size_in
                    a = VecZnx(n, cols_in, size_in).fill_uniform(base2k, rng)
                    mat = MatZnx(n, rows, cols_in, cols_out, size_out).fill_uniform(base2k, rng)
                    ad = R.vec_znx_dft_alloc(cols_in, size_in)
                    for j in range(cols_in):
                        R.vec_znx_dft_apply(1, 0, ad, j, a, j)
                    pm = R.vmp_pmat_alloc(rows, cols_in, cols_out, size_out)
                    R.vmp_prepare(pm, mat)
                    for limb_offset in range(0, size_out):
                        rd = R.vec_znx_dft_alloc(cols_out, size_out)  # zeroed, like test_suite/vmp.rs:260-262
                        R.vmp_apply_dft_to_dft(rd, ad, pm, limb_offset)
                        big = R.vec_znx_idft_apply_consume(rd)
                        want = exact.vmp_exact(a.data, mat.data, limb_offset, size_out)
                        assert np.array_equal(big.data.astype(object), want), (cols_in, cols_out, size_in, size_out, limb_offset)


def test_P2_svp_equals_exact_product():
    n, base2k = 64, 17
    R = RefModule(n)
    rng = seeded(5)
    s = ScalarZnx(n, 2).fill_uniform(base2k, rng)
    b = VecZnx(n, 2, 3).fill_uniform(base2k, rng)
    pp = SvpPPol(n, 2)
    for c in range(2):
        R.svp_prepare(pp, c, s, c)
    d = R.vec_znx_dft_alloc(2, 4)
    d.data[...] = 123.0
    for c in range(2):
        R.svp_apply_dft(d, c, pp, 1 - c, b, c)
    big = R.vec_znx_idft_apply_consume(d)
    for c in range(2):
        for j in range(3):
            want = exact.negacyclic_mul(s.at(1 - c, 0), b.at(c, j))
            assert np.array_equal(big.at(c, j).astype(object), want)
        assert not big.at(c, 3).any()


@pytest.mark.parametrize("base2k", [1, 2, 12, 17, 19, 31, 52])
def test_P3_normalize_same_base_exact(base2k):
    n = 32
    R = RefModule(n)
    rng = seeded(base2k)
    for a_size in range(1, 6):
        for res_size in range(1, 6):
            # keep |value| small enough that the reference's i64 carries cannot overflow
            a = VecZnxBig(n, 2, a_size).fill_uniform(min(62, 2 * base2k + 20), rng)
            res = VecZnx(n, 2, res_size)
            res.data[...] = -1
            R.vec_znx_big_normalize(res, base2k, 0, 1, a, base2k, 0)
            want = exact.normalize_exact(a.data[:, 0, :], base2k, res_size)
            assert np.array_equal(res.data[:, 1, :], want), (a_size, res_size)
            h = 1 << (base2k - 1)
            assert res.data[:, 1, :].min() >= -h and res.data[:, 1, :].max() < h
            assert np.all(res.data[:, 0, :] == -1)  # other column untouched


@pytest.mark.parametrize("prec", [128])
def test_P3_normalize_cross_base_value_preserved(prec):
    """Mirror of test_vec_znx_normalize_cross_base2k (reference/vec_znx/normalize.rs:428-540): all
    in/out base2k pairs, the reference's 13 offsets, 60-bit un-normalized inputs, output sized so that
    no precision is lost; the torus value must be preserved to 2^(-min_prec+1)."""
    n = 8
    R = RefModule(n)
    rng = seeded(prec)
    bases = range(1, 52)
    for ak in bases:
        for rk in bases:
            in_size = -(-prec // ak)
            in_prec = in_size * ak
            out_size = -(-in_prec // rk)
            a = VecZnx(n, 1, in_size).fill_uniform(60, rng)
            for off in (-prec, -(prec - 1), -(prec - ak), -(ak + 1), ak, -(ak - 1), 0, ak - 1, ak, ak + 1, prec - ak, prec - 1, prec):
                res = VecZnx(n, 1, out_size).fill_uniform(60, rng)
                R.vec_znx_big_normalize(res, rk, off, 0, a, ak, 0)
                assert exact.torus_equal(a.data[:, 0, :], ak, res.data[:, 0, :], rk, off), (ak, rk, off)


def test_dft_apply_step_offset_semantics():
    """vec_znx_dft.rs:160-200: limb selection, limbs past a.size left untouched, tail zeroed."""
    n = 32
    R = RefModule(n)
    rng = seeded(3)
    a = VecZnx(n, 1, 5).fill_uniform(12, rng)
    for step, offset in ((1, 0), (1, 2), (2, 2), (2, 1), (3, 1)):
        for res_size in (1, 3, 6):
            d = VecZnxDft(n, 1, res_size)
            d.data[...] = 7.5
            R.vec_znx_dft_apply(step, offset, d, 0, a, 0)
            steps = -(-5 // step)
            min_steps = min(res_size, steps)
            for j in range(res_size):
                limb = offset + j * step
                if j >= min_steps:
                    assert not d.data[j].any()
                elif limb >= 5:
                    assert np.all(d.data[j] == 7.5)
                else:
                    single = VecZnxDft(n, 1, 1)
                    R.vec_znx_dft_apply(1, 0, single, 0, VecZnx(n, 1, 1, a.data[limb].copy()), 0)
                    assert np.array_equal(single.data[0], d.data[j])


@pytest.mark.parametrize("dsize", [1, 2, 3, 4, 5])
def test_external_product_and_keyswitch_match_exact(dsize):
    """poulpy-core glue restated in the oracle (external_product/glwe.rs, keyswitching/glwe.rs) vs the
    exact bivariate product + big-int normalize, incl. dsize > 1 folding (zero-tail semantics)."""
    n, base2k, rank = 32, 12, 1
    cols = rank + 1
    R = RefModule(n)
    rng = seeded(40 + dsize)
    a_size, dnum = 4, 2 if dsize > 1 else 4
    key_size = 5
    a = VecZnx(n, cols, a_size).fill_uniform(base2k, rng)
    mat = MatZnx(n, dnum, cols, cols, key_size).fill_uniform(base2k, rng)
    pm = R.vmp_pmat_alloc(dnum, cols, cols, key_size)
    R.vmp_prepare(pm, mat)
    res = VecZnx(n, cols, 4)
    R.glwe_external_product(res, base2k, a, base2k, pm, dsize, base2k)
    # exact: sum over digits di of (limbs offset+j*dsize of a) x (mat shifted by di limbs)
    big = np.zeros((key_size, cols, n), dtype=object)
    for di in range(dsize):
        sel = a.data[dsize - 1 - di::dsize][: (a_size + di) // dsize]
        drop = max(dsize - di - 2, 0)
        part = exact.vmp_exact(sel, mat.data, di, key_size - drop)
        big[: key_size - drop] += part
    want = np.zeros((4, cols, n), dtype=np.int64)
    for c in range(cols):
        want[:, c, :] = exact.normalize_exact(big[:, c, :], base2k, 4)
    assert np.array_equal(res.data, want)


@pytest.mark.parametrize("rank", [1, 2, 3])
def test_P8_ggsw_expand_row_matches_exact(rank):
    """conversion/gglwe_to_ggsw.rs:116-268 restated in the oracle vs exact integers: res.at(row, col) = normalize(mask of
    res.at(row, 0) x tsk.at(col-1) + body of res.at(row, 0) on column col); column 0 untouched."""
    n, base2k = 32, 13
    cols = rank + 1
    R = RefModule(n)
    rng = seeded(800 + rank)
    dnum, size, key_dnum, key_size = 2, 3, 3, 4
    ggsw = MatZnx(n, dnum, cols, cols, size).fill_uniform(base2k, rng)
    before = ggsw.data.copy()
    mats, keys = [], []
    for c in range(rank):
        mat = MatZnx(n, key_dnum, rank, cols, key_size).fill_uniform(base2k, rng)
        pm = R.vmp_pmat_alloc(key_dnum, rank, cols, key_size)
        R.vmp_prepare(pm, mat)
        mats.append(mat)
        keys.append(pm)
    R.ggsw_expand_row(ggsw, base2k, keys, 1, base2k)
    for row in range(dnum):
        a = before[row, 0]  # (size, cols, n)
        assert np.array_equal(ggsw.data[row, 0], a)
        for col in range(1, cols):
            big = exact.vmp_exact(np.ascontiguousarray(a[:, 1:, :]), mats[col - 1].data, 0, key_size)
            big[:size, col, :] += a[:, 0, :].astype(object)
            for c in range(cols):
                want = exact.normalize_exact(big[:, c, :], base2k, size)
                assert np.array_equal(ggsw.data[row, col, :, c, :], want), (rank, row, col, c)


@pytest.mark.parametrize("n", [8, 64, 1024])
def test_P5_vec_znx_automorphism_direct(n):
    """reference/znx/automorphism.rs restated (sequential index walk) vs the direct statement of X -> X^p, incl. the
    reference's own test values p = -5, 5 (poulpy-hal/src/test_suite/vec_znx.rs), negative and > 2n elements, zeroed
    tail limbs, and the composition law phi_p(phi_q(a)) = phi_{pq}(a)."""
    R = RefModule(n)
    rng = seeded(n)
    a = VecZnx(n, 2, 3).fill_uniform(40, rng)
    for p in (-5, 5, 3, -1, 2 * n - 1, 2 * n + 3, 5 ** 9):
        r = VecZnx(n, 2, 4).fill_uniform(10, rng)
        R.vec_znx_automorphism(p, r, 1, a, 0)
        assert np.array_equal(r.data[:3, 1], exact.automorphism_exact(a.data[:, 0], p))
        assert not r.data[3, 1].any()
        q = 7
        t = a.copy()
        R.vec_znx_automorphism_assign(q, t, 0)
        R.vec_znx_automorphism_assign(p, t, 0)
        assert np.array_equal(t.data[:, 0], exact.automorphism_exact(a.data[:, 0], p * q))
        assert np.array_equal(t.data[:, 1], a.data[:, 1])


@pytest.mark.parametrize("mode", ["automorphism", "add", "sub", "sub_negate"])
@pytest.mark.parametrize("rank", [1, 2])
def test_P5_glwe_automorphism_family_matches_exact(mode, rank):
    """automorphism/glwe_ct.rs:51-275 restated in the oracle vs exact integers: big = exact key-switch value (mask x key
    + body), then phi / +-a / big-int normalize in the order the reference applies them."""
    n, base2k = 32, 13
    cols = rank + 1
    R = RefModule(n)
    rng = seeded(500 + rank)
    a_size, dnum, key_size, res_size = 3, 3, 4, 4
    a = VecZnx(n, cols, a_size).fill_uniform(base2k, rng)
    mat = MatZnx(n, dnum, rank, cols, key_size).fill_uniform(base2k, rng)
    pm = R.vmp_pmat_alloc(dnum, rank, cols, key_size)
    R.vmp_prepare(pm, mat)
    big = exact.vmp_exact(np.ascontiguousarray(a.data[:, 1:, :]), mat.data, 0, key_size)
    big[:a_size, 0, :] += a.data[:, 0, :].astype(object)
    for p in (-5, 3, 2 * n - 1):
        res = VecZnx(n, cols, res_size)
        R.glwe_automorphism(res, base2k, a, base2k, pm, 1, base2k, p, mode)
        want = np.zeros((res_size, cols, n), dtype=np.int64)
        for c in range(cols):
            if mode == "automorphism":
                want[:, c, :] = exact.automorphism_exact(exact.normalize_exact(big[:, c, :], base2k, res_size), p)
                continue
            v = exact.automorphism_exact(big[:, c, :], p)
            av = np.zeros_like(v)
            av[:a_size] = a.data[:, c, :].astype(object)
            v = {"add": v + av, "sub": v - av, "sub_negate": av - v}[mode]
            want[:, c, :] = exact.normalize_exact(v, base2k, res_size)
        assert np.array_equal(res.data, want), (mode, rank, p)


def _blind_rotation_inputs(n, rank, n_lwe, dnum, brk_size, res_size, base2k, seed):
    rng = seeded(seed)
    cols = rank + 1
    lut = VecZnx(n, 1, res_size).fill_uniform(base2k, rng)
    mats = [MatZnx(n, dnum, cols, cols, brk_size).fill_uniform(base2k, rng) for _ in range(n_lwe)]
    lwe_2n = rng.integers(-n, n, n_lwe + 1, dtype=np.int64)   # what mod_switch_2n produces: values in [-n, n)
    return lut, mats, lwe_2n


@pytest.mark.parametrize("n", [16, 64])
def test_P6_rotate_and_normalize_assign(n):
    R = RefModule(n)
    rng = seeded(n + 3)
    a = VecZnx(n, 2, 3).fill_uniform(40, rng)
    for p in (0, 1, -1, 5, n - 1, n, n + 3, 2 * n - 1, -n, 3 * n + 2):
        r = VecZnx(n, 2, 4).fill_uniform(20, rng)
        R.vec_znx_rotate(p, r, 1, a, 0)
        assert np.array_equal(r.data[:3, 1], exact.rotate_exact(a.data[:, 0], p))
        assert not r.data[3, 1].any()
        t = a.copy()
        R.vec_znx_mul_xp_minus_one_assign(p, t, 1)
        assert np.array_equal(t.data[:, 1], exact.rotate_exact(a.data[:, 1], p) - a.data[:, 1])
    # in-place same-base normalize == the out-of-place one == big-int digits
    for base2k in (7, 12, 19):
        x = VecZnx(n, 1, 4).fill_uniform(55, rng)
        want = exact.normalize_exact(x.data[:, 0, :].astype(object), base2k, 4)
        y = x.copy()
        R.vec_znx_normalize_assign(base2k, y, 0)
        assert np.array_equal(y.data[:, 0, :], want)


@pytest.mark.parametrize("block_size", [1, 3])
@pytest.mark.parametrize("rank", [1, 2])
def test_P6_blind_rotation_matches_exact(block_size, rank):
    """algorithm.rs:265-368 (block binary) / :370-440 (standard) restated in the oracle vs the same recurrence on
    exact integers: acc <- normalize(acc + sum_i (X^a_i - 1) * (acc (x) BRK_i))."""
    n, base2k, n_lwe, dnum, brk_size, res_size = 32, 10, 6, 2, 3, 2
    cols = rank + 1
    R = RefModule(n)
    lut, mats, lwe_2n = _blind_rotation_inputs(n, rank, n_lwe, dnum, brk_size, res_size, base2k, 900 + rank)
    brk = np.stack([R.vmp_pmat_alloc(dnum, cols, cols, brk_size).data for _ in range(n_lwe)])
    for i, mt in enumerate(mats):
        pm = R.vmp_pmat_alloc(dnum, cols, cols, brk_size)
        R.vmp_prepare(pm, mt)
        brk[i] = pm.data
    brk = np.ascontiguousarray(brk.reshape(n_lwe, -1))
    xpa = R.blind_rotation_x_pow_a()
    res = VecZnx(n, cols, res_size)
    res.data[...] = 77
    R.blind_rotation_execute(res, base2k, lwe_2n, lut, brk, dnum, brk_size, block_size, xpa)

    acc = np.zeros((res_size, cols, n), dtype=object)
    acc[:, 0, :] = exact.rotate_exact(lut.data[:, 0, :].astype(object), int(lwe_2n[0]))
    a = [int(v) for v in lwe_2n[1:]]
    if block_size > 1:
        for blk in range(0, n_lwe - block_size + 1, block_size):
            add = np.zeros((brk_size, cols, n), dtype=object)
            acc_in = np.array(acc[:min(dnum, res_size)], dtype=np.int64)
            for k in range(blk, blk + block_size):
                v = exact.vmp_exact(acc_in, mats[k].data, 0, brk_size)
                add += exact.rotate_exact(v, a[k]) - v
            add[:min(brk_size, res_size)] += acc[:min(brk_size, res_size)]
            for c in range(cols):
                acc[:, c, :] = exact.normalize_exact(add[:, c, :], base2k, res_size)
    else:
        for k in range(n_lwe):
            v = exact.vmp_exact(np.array(acc, dtype=np.int64), mats[k].data, 0, brk_size)
            tmp = np.zeros((res_size, cols, n), dtype=object)
            for c in range(cols):
                tmp[:, c, :] = exact.normalize_exact(v[:, c, :], base2k, res_size)
            acc = acc + exact.rotate_exact(tmp, a[k]) - tmp
        for c in range(cols):
            acc[:, c, :] = exact.normalize_exact(acc[:, c, :], base2k, res_size)
    assert np.array_equal(res.data, np.array(acc, dtype=np.int64))


def _val(limbs, base2k):
    tot = base2k * len(limbs)
    return sum(int(v) << (tot - (j + 1) * base2k) for j, v in enumerate(limbs))


def test_P7_rsh_assign_is_a_rounded_shift():
    """reference/vec_znx/shift.rs:186-243 restated: for k <= base2k (one limb falls off; glwe_trace uses k = 1) the result is the
    value shifted by k bits, rounded to nearest at the precision of res, in balanced digits, other columns untouched."""
    n = 16
    R = RefModule(n)
    rng = seeded(21)
    for base2k in (5, 12, 17):
        for k in (1, 2, base2k - 1, base2k):
            for size in (1, 3, 4):
                a = VecZnx(n, 2, size).fill_uniform(base2k, rng)
                b = a.copy()
                R.vec_znx_rsh_assign(base2k, k, b, 1)
                assert np.array_equal(a.data[:, 0], b.data[:, 0])
                assert np.abs(b.data[:, 1]).max() <= 1 << (base2k - 1)
                for i in range(n):
                    va, vb = _val(a.data[:, 1, i], base2k), _val(b.data[:, 1, i], base2k)
                    assert abs(va - (vb << k)) <= 1 << (k - 1), (base2k, k, size)


def test_P7_glwe_trace_matches_exact():
    """glwe_trace.rs:164-174 restated in the oracle vs exact integers: per step res <- normalize(phi_p(KS_p(rsh(res))) + rsh(res))
    with the exact key-switch value (mask x key + body) and the oracle's own rsh (pinned above)."""
    n, base2k, rank = 32, 13, 1
    cols = rank + 1
    R = RefModule(n)
    rng = seeded(77)
    size, dnum, key_size = 3, 3, 4
    gals = [-1, 5, 25 % (2 * n)]
    mats = [MatZnx(n, dnum, rank, cols, key_size).fill_uniform(base2k, rng) for _ in gals]
    pms = []
    for mt in mats:
        pm = R.vmp_pmat_alloc(dnum, rank, cols, key_size)
        R.vmp_prepare(pm, mt)
        pms.append(pm)
    res = VecZnx(n, cols, size).fill_uniform(base2k, rng)
    cur = res.copy()
    R.glwe_trace_assign(res, base2k, gals, pms)
    for p, mt in zip(gals, mats):
        for c in range(cols):
            R.vec_znx_rsh_assign(base2k, 1, cur, c)
        big = exact.vmp_exact(np.ascontiguousarray(cur.data[:, 1:, :]), mt.data, 0, key_size)
        big[:size, 0, :] += cur.data[:, 0, :].astype(object)
        nxt = np.zeros_like(cur.data)
        for c in range(cols):
            v = exact.automorphism_exact(big[:, c, :], p)
            v[:size] += cur.data[:, c, :].astype(object)
            nxt[:, c, :] = exact.normalize_exact(v, base2k, size)
        cur.data[...] = nxt
    assert np.array_equal(res.data, cur.data)


def test_P9_vec_znx_limbwise_family_direct():
    """reference/vec_znx/{add,sub,negate,copy}.rs restated in the oracle vs the direct numpy statement (zero-extended operands,
    wrapping i64), for every ordering of the sizes."""
    n = 16
    R = RefModule(n)
    rng = seeded(99)

    def ext(v, col, size):   # limbs of column col, zero-extended / truncated to `size`
        out = np.zeros((size, n), dtype=np.int64)
        k = min(size, v.size)
        out[:k] = v.data[:k, col]
        return out

    with np.errstate(over="ignore"):
        for rs, asz, bsz in [(3, 2, 4), (3, 4, 2), (2, 3, 3), (4, 1, 1), (1, 3, 2), (2, 2, 2)]:
            a = VecZnx(n, 2, asz).fill_uniform(63, rng)
            b = VecZnx(n, 2, bsz).fill_uniform(63, rng)
            r = VecZnx(n, 2, rs).fill_uniform(63, rng)
            R.vec_znx_add_into(r, 1, a, 0, b, 1)
            assert np.array_equal(r.data[:, 1], ext(a, 0, rs) + ext(b, 1, rs))
            R.vec_znx_sub(r, 0, a, 1, b, 0)
            assert np.array_equal(r.data[:, 0], ext(a, 1, rs) - ext(b, 0, rs))
            r0 = r.copy()
            R.vec_znx_add_assign(r, 0, a, 0)
            assert np.array_equal(r.data[:, 0], r0.data[:, 0] + ext(a, 0, rs)) and np.array_equal(r.data[:, 1], r0.data[:, 1])
            r0 = r.copy()
            R.vec_znx_sub_assign(r, 1, a, 1)
            assert np.array_equal(r.data[:, 1], r0.data[:, 1] - ext(a, 1, rs))
            r0 = r.copy()
            R.vec_znx_sub_negate_assign(r, 0, b, 0)
            assert np.array_equal(r.data[:, 0], ext(b, 0, rs) - r0.data[:, 0])
            R.vec_znx_negate(r, 1, b, 1)
            assert np.array_equal(r.data[:, 1], -ext(b, 1, rs))
            r0 = r.copy()
            R.vec_znx_negate_assign(r, 0)
            assert np.array_equal(r.data[:, 0], -r0.data[:, 0])
            R.vec_znx_copy(r, 0, a, 1)
            assert np.array_equal(r.data[:, 0], ext(a, 1, rs))


def test_P10_shifts_are_offset_normalizations():
    """reference/vec_znx/shift.rs restated literally (lsh, lsh_assign, rsh out of place) vs vec_znx_normalize at equal bases with
    res_offset = +k / -k, and vs the exact value: lsh/rsh by k bits multiplies the torus value by 2^(+-k) (mod 1, rounding of
    the dropped bits as the normalization does); every (res_size, a_size) ordering, k from 0 past the total precision."""
    n, base2k = 16, 11
    R = RefModule(n)
    rng = seeded(1010)
    for rs, asz in [(3, 3), (2, 4), (4, 2), (1, 3), (3, 1)]:
        for k in list(range(0, base2k * (max(rs, asz) + 1) + 3)):
            a = VecZnx(n, 2, asz).fill_uniform(40, rng)
            r1 = VecZnx(n, 2, rs).fill_uniform(20, rng)
            r2 = r1.copy()
            R.vec_znx_lsh(base2k, k, r1, 1, a, 0)
            R.vec_znx_big_normalize(r2, base2k, k, 1, a, base2k, 0)
            assert np.array_equal(r1.data, r2.data), ("lsh", rs, asz, k)
            r1 = VecZnx(n, 2, rs).fill_uniform(20, rng)
            r2 = r1.copy()
            R.vec_znx_rsh(base2k, k, r1, 0, a, 1)
            R.vec_znx_big_normalize(r2, base2k, -k, 0, a, base2k, 1)
            assert np.array_equal(r1.data, r2.data), ("rsh", rs, asz, k)
        for k in range(0, base2k * rs + 3):
            x = VecZnx(n, 2, rs).fill_uniform(40, rng)
            y = x.copy()
            z = VecZnx(n, 2, rs)
            R.vec_znx_lsh_assign(base2k, k, x, 1)
            R.vec_znx_lsh(base2k, k, z, 1, y, 1)
            assert np.array_equal(x.data[:, 1], z.data[:, 1]) and np.array_equal(x.data[:, 0], y.data[:, 0]), ("lsh_assign", rs, k)
    # exact value: rsh keeps the torus value / 2^k up to the precision of res (balanced rounding of what falls off)
    a = VecZnx(n, 1, 3).fill_uniform(base2k, rng)
    for k in (1, 5, 11, 13):
        r = VecZnx(n, 1, 4)
        R.vec_znx_rsh(base2k, k, r, 0, a, 0)
        assert exact.torus_equal(a.data[:, 0], base2k, r.data[:, 0], base2k, res_offset=-k)


def test_P11_glwe_pack_tree_walk():
    """The oracle's C restatement of glwe_pack (glwe_packing.rs:122-176) vs an independent Python walk of the same tree with
    a dict, as the reference's HashMap code reads, built from the oracle's primitive operations (rotate, add / sub, rsh,
    normalize_assign, the pinned glwe_automorphism family, trace): sparse and dense occupancy, every branch of pack_internal."""
    n, rank, size, base2k, dnum = 32, 1, 3, 12, 3
    cols = rank + 1
    log_n = 5
    R = RefModule(n)
    rng = seeded(1111)
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    keys = []
    for _ in gals:
        mat = MatZnx(n, dnum, rank, cols, size).fill_uniform(base2k, rng)
        pm = R.vmp_pmat_alloc(dnum, rank, cols, size)
        R.vmp_prepare(pm, mat)
        keys.append(pm)

    def glwe(fn, *cts_and_args):
        for c in range(cols):
            fn(c)

    def py_pack(cts, log_gap_out):
        a = dict(cts)
        for i in range(log_n - log_gap_out):
            t = 1 << (log_n - 1 - i)
            for j in range(t):
                lo, hi = a.pop(j, None), a.pop(j + t, None)
                if lo is not None and hi is not None:
                    tmp = lo.copy()
                    for c in range(cols):
                        R.vec_znx_rotate(-t, lo, c, tmp, c)
                    tmp_b = VecZnx(n, cols, size)
                    for c in range(cols):
                        R.vec_znx_sub(tmp_b, c, lo, c, hi, c)
                        R.vec_znx_rsh_assign(base2k, 1, tmp_b, c)
                        R.vec_znx_add_assign(lo, c, hi, c)
                        R.vec_znx_rsh_assign(base2k, 1, lo, c)
                        R.vec_znx_normalize_assign(base2k, tmp_b, c)
                    src = tmp_b.copy()
                    R.glwe_automorphism(tmp_b, base2k, src, base2k, keys[i], 1, base2k, gals[i], "automorphism")
                    for c in range(cols):
                        R.vec_znx_sub_assign(lo, c, tmp_b, c)
                        R.vec_znx_normalize_assign(base2k, lo, c)
                    tmp = lo.copy()
                    for c in range(cols):
                        R.vec_znx_rotate(t, lo, c, tmp, c)
                elif lo is not None:
                    for c in range(cols):
                        R.vec_znx_rsh_assign(base2k, 1, lo, c)
                    src = lo.copy()
                    R.glwe_automorphism(lo, base2k, src, base2k, keys[i], 1, base2k, gals[i], "add")
                elif hi is not None:
                    tmp_b = VecZnx(n, cols, size)
                    for c in range(cols):
                        R.vec_znx_rotate(t, tmp_b, c, hi, c)
                        R.vec_znx_rsh_assign(base2k, 1, tmp_b, c)
                    R.glwe_automorphism(hi, base2k, tmp_b, base2k, keys[i], 1, base2k, gals[i], "sub_negate")
                if lo is not None:
                    a[j] = lo
                elif hi is not None:
                    a[j] = hi
        res = a[0].copy()
        skip = log_n - log_gap_out
        R.glwe_trace_assign(res, base2k, gals[skip:], keys[skip:])
        return res

    for indices, log_gap_out in (([0, 4, 8, 12, 16, 20, 24, 28], 2), ([0, 3, 9, 16, 17, 31], 0), ([4], 2), ([0, 16], 4)):
        cts = {j: VecZnx(n, cols, size).fill_uniform(base2k, rng) for j in indices}
        want = py_pack({j: v.copy() for j, v in cts.items()}, log_gap_out)
        res = VecZnx(n, cols, size)
        R.glwe_pack(res, base2k, cts, log_gap_out, gals, keys)
        assert np.array_equal(res.data, want.data), (indices, log_gap_out)


def test_P12_extended_blind_rotation_degenerates_to_block_binary():
    """execute_block_binary_extended restated literally (algorithm.rs:121-273): with extension_factor 1 it must reproduce the
    block-binary rotation (pinned by P6) bit for bit; with extension_factor > 1 and every a_i a multiple of the factor
    (ai_lo = 0: no movement between the accumulators) accumulator 0 is the block-binary rotation of lut[0] by a_i / ext."""
    n, rank, n_lwe, blk, dnum, bsz, rsz, k = 64, 1, 6, 3, 2, 2, 2, 12
    R = RefModule(n)
    rng = seeded(1212)
    cols = rank + 1
    brk = np.empty((n_lwe, n * dnum * cols * cols * bsz))
    for i in range(n_lwe):
        mat = MatZnx(n, dnum, cols, cols, bsz).fill_uniform(k, rng)
        pm = R.vmp_pmat_alloc(dnum, cols, cols, bsz)
        R.vmp_prepare(pm, mat)
        brk[i] = pm.data.reshape(-1)
    xpa = R.blind_rotation_x_pow_a()
    lut = VecZnx(n, 1, rsz).fill_uniform(k, rng)
    lwe = rng.integers(-n, n, n_lwe + 1, dtype=np.int64)
    r1, r2 = VecZnx(n, cols, rsz), VecZnx(n, cols, rsz)
    R.blind_rotation_execute(r1, k, lwe, lut, brk, dnum, bsz, blk, xpa)
    R.blind_rotation_execute_extended(r2, k, lwe, np.ascontiguousarray(lut.data.reshape(1, rsz, 1, n)), brk, dnum, bsz, blk, xpa)
    assert np.array_equal(r1.data, r2.data)
    ext = 4
    luts = rng.integers(-(1 << (k - 1)), 1 << (k - 1), (ext, rsz, 1, n), dtype=np.int64)
    luts[0] = lut.data.reshape(rsz, 1, n)
    r3 = VecZnx(n, cols, rsz)
    R.blind_rotation_execute_extended(r3, k, lwe * ext, luts, brk, dnum, bsz, blk, xpa)
    assert np.array_equal(r3.data, r1.data)

