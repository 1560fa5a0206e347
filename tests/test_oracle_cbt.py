"""CPU: the oracle's circuit bootstrapping (poulpy-bin-fhe/src/circuit_bootstrapping/circuit.rs:219-421) is a composition of pieces pinned
elsewhere (P6 blind rotation, P3 normalize, glwe_trace_assign, rotate, P8 ggsw_expand_row).  The reference's tests run it with ONE BASE2K
PER OBJECT (tests/circuit_bootstrapping.rs:49-53: result 15, blind-rotation key 13, tensor keys 12, automorphism keys 11): here the C
restatement `pzr_circuit_bootstrap_bases` is compared with the same composition written out in Python on the pinned primitives, for equal
and for different bases."""
import numpy as np
import pytest

from oracle.ref import RefModule
from poulpy_amd.layouts import MatZnx, VecZnx
from tests.helpers import seeded


def cdiv(a, b):
    return -(-a // b)


def cbt_shape(res_base2k, brk_base2k, tsk_base2k, atk_base2k, res_limbs):
    """the sizes the reference derives (tests/circuit_bootstrapping.rs:62-71, circuit.rs:303-331, glwe_trace.rs:107-112)"""
    k_res = res_limbs * res_base2k
    k_brk = k_res + brk_base2k
    k_atk = k_res + tsk_base2k      # (the reference's test names these two the other way round)
    k_tsk = k_res + atk_base2k
    return dict(res_size=res_limbs, glwe_size=cdiv(k_brk, brk_base2k), atk_glwe_size=cdiv(k_brk, atk_base2k),
                trace_size=cdiv(max(k_brk, k_res), atk_base2k), atk_size=cdiv(k_atk, atk_base2k), tsk_size=cdiv(k_tsk, tsk_base2k),
                res_conv_size=cdiv(k_res, tsk_base2k))


def make_cbt_inputs(ref, n, rank, n_lwe, brk_dnum, bases, sh, atk_dnum, tsk_dnum, rng, skip=0, prepare_also=None):
    """random key material in each key's own base; prepare_also(rows, cols_in, size, mat) lets the GPU test prepare the same matrices"""
    k_brk, k_atk, k_tsk, _ = bases
    cols = rank + 1
    log_n = n.bit_length() - 1

    def prepared(rows, cols_in, size, k):
        mat = MatZnx(n, rows, cols_in, cols, size).fill_uniform(k, rng)
        pr = ref.vmp_pmat_alloc(rows, cols_in, cols, size)
        ref.vmp_prepare(pr, mat)
        return (pr, prepare_also(rows, cols_in, size, mat)) if prepare_also else (pr, None)

    lut = VecZnx(n, 1, sh["glwe_size"]).fill_uniform(k_brk, rng)
    brk = [prepared(brk_dnum, cols, sh["glwe_size"], k_brk) for _ in range(n_lwe)]
    gals = ([-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)])[skip:]
    atk = [prepared(atk_dnum, rank, sh["atk_size"], k_atk) for _ in gals]
    tsk = [prepared(tsk_dnum, rank, sh["tsk_size"], k_tsk) for _ in range(rank)]
    return lut, brk, gals, atk, tsk


def compose_constant(ref, n, rank, bases, sh, lwe_2n, lut, brk_r, brk_dnum, block_size, xpa, gals, atk, tsk, res_dnum, gap):
    k_brk, k_atk, k_tsk, k_res = bases
    cols = rank + 1
    acc_brk = VecZnx(n, cols, sh["glwe_size"])
    ref.blind_rotation_execute(acc_brk, k_brk, lwe_2n, lut, brk_r, brk_dnum, sh["glwe_size"], block_size, xpa)
    acc = VecZnx(n, cols, sh["atk_glwe_size"])
    if k_atk == k_brk:
        acc.data[...] = acc_brk.data
    else:
        for c in range(cols):
            ref.vec_znx_normalize(acc, k_atk, 0, c, acc_brk, k_brk, c)
    g = MatZnx(n, res_dnum, cols, cols, sh["res_size"])
    for i in range(res_dnum):
        tmp = VecZnx(n, cols, sh["trace_size"])
        tmp.data[:acc.size] = acc.data
        ref.glwe_trace_assign(tmp, k_atk, gals, atk)
        row = VecZnx(n, cols, sh["res_size"])
        if k_res == k_atk:
            row.data[...] = tmp.data[:sh["res_size"]]
        else:
            for c in range(cols):
                ref.vec_znx_normalize(row, k_res, 0, c, tmp, k_atk, c)
        g.data[i, 0] = row.data
        if i + 1 < res_dnum:
            rot = VecZnx(n, cols, acc.size)
            for c in range(cols):
                ref.vec_znx_rotate(-gap, rot, c, acc, c)
            acc = rot
    ref.ggsw_expand_row(g, k_res, tsk, 1, k_tsk)
    return g.data


@pytest.mark.parametrize("bases,res_limbs,rank,block_size", [
    ((13, 13, 13, 13), 2, 1, 2),
    ((13, 11, 12, 15), 2, 1, 2),      # the reference's test bases (brk, atk, tsk, res)
    ((13, 11, 12, 15), 2, 2, 1),      # rank 2, standard rotation
    ((12, 17, 12, 9), 3, 1, 3),       # automorphism keys in a larger base than everything else
])
def test_cbt_bases_is_the_composition_of_the_pinned_pieces(bases, res_limbs, rank, block_size):
    n, n_lwe, brk_dnum, res_dnum = 64, 6, 2, 2
    k_brk, k_atk, k_tsk, k_res = bases
    ref = RefModule(n)
    rng = seeded(sum(bases) + rank)
    sh = cbt_shape(k_res, k_brk, k_tsk, k_atk, res_limbs)
    lut, brk, gals, atk, tsk = make_cbt_inputs(ref, n, rank, n_lwe, brk_dnum, bases, sh, sh["trace_size"], sh["res_conv_size"], rng)
    brk_r = np.stack([b[0].data.reshape(-1) for b in brk])
    lwe = rng.integers(-n, n, n_lwe + 1, dtype=np.int64)
    gap = 2 * int(rng.integers(1, n // 8))
    xpa = ref.blind_rotation_x_pow_a() if block_size > 1 else np.zeros((1, 1))
    g = MatZnx(n, res_dnum, rank + 1, rank + 1, sh["res_size"])
    g.data[...] = 0x5A
    ref.circuit_bootstrap_bases(g, bases, False, lwe, lut, brk_r, brk_dnum, sh["glwe_size"], sh["glwe_size"], sh["atk_glwe_size"],
                                sh["trace_size"], block_size, xpa, gals, [a[0] for a in atk], [t[0] for t in tsk], gap)
    want = compose_constant(ref, n, rank, bases, sh, lwe, lut, brk_r, brk_dnum, block_size, xpa, gals, [a[0] for a in atk],
                            [t[0] for t in tsk], res_dnum, gap)
    assert np.array_equal(g.data, want)
    if len(set(bases)) == 1:   # the one-base entry point is the same function
        g2 = MatZnx(n, res_dnum, rank + 1, rank + 1, sh["res_size"])
        ref.circuit_bootstrap_to_constant(g2, k_brk, lwe, lut, brk_r, brk_dnum, sh["glwe_size"], sh["glwe_size"], block_size, xpa, gals,
                                          [a[0] for a in atk], [t[0] for t in tsk], gap)
        assert np.array_equal(g2.data, want)


def test_cbt_bases_rows_carry_the_same_torus_values_in_any_result_base():
    """column 0 of every GGSW row is the trace of the rotated accumulator re-expressed in the result base: with the same keys and two
    result bases the rows agree as torus elements up to the precision of the coarser one (P3's statement, through the whole chain)"""
    from oracle import exact
    n, rank, n_lwe, brk_dnum, res_dnum, block_size = 64, 1, 4, 2, 2, 2
    ref = RefModule(n)
    rows = {}
    for k_res, res_limbs in ((15, 2), (10, 3)):
        bases = (13, 11, 12, k_res)
        rng = seeded(99)
        sh = cbt_shape(15, 13, 12, 11, 2)      # same key material for both
        lut, brk, gals, atk, tsk = make_cbt_inputs(ref, n, rank, n_lwe, brk_dnum, (13, 11, 12, 15), sh, sh["trace_size"], 3, rng)
        brk_r = np.stack([b[0].data.reshape(-1) for b in brk])
        lwe = rng.integers(-n, n, n_lwe + 1, dtype=np.int64)
        g = MatZnx(n, res_dnum, rank + 1, rank + 1, res_limbs)
        ref.circuit_bootstrap_bases(g, bases, False, lwe, lut, brk_r, brk_dnum, sh["glwe_size"], sh["glwe_size"], sh["atk_glwe_size"],
                                    sh["trace_size"], block_size, ref.blind_rotation_x_pow_a(), gals, [a[0] for a in atk],
                                    [t[0] for t in tsk], 6)
        rows[k_res] = g.data[:, 0].copy()     # (row, limb, col, n)
    for i in range(res_dnum):
        for c in range(rank + 1):
            assert exact.torus_equal(rows[15][i, :, c, :], 15, rows[10][i, :, c, :], 10, 0, 2)


@pytest.mark.parametrize("res_k,key_k,kbits,rank", [(14, 13, 57, 1), (12, 15, 40, 2)])
def test_trace_assign_bases_is_normalize_trace_normalize(res_k, key_k, kbits, rank):
    """pzr_glwe_trace_assign_bases == glwe_trace.rs:153-163 written out on the pinned normalize (P3) and the equal-base trace"""
    n = 64
    ref = RefModule(n)
    rng = seeded(res_k * 100 + key_k)
    cols = rank + 1
    res_size, conv_size, key_size = cdiv(kbits, res_k), cdiv(kbits, key_k), cdiv(kbits + key_k, key_k)
    log_n = n.bit_length() - 1
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    keys = []
    for _ in gals:
        mat = MatZnx(n, conv_size, rank, cols, key_size).fill_uniform(key_k, rng)
        pr = ref.vmp_pmat_alloc(conv_size, rank, cols, key_size)
        ref.vmp_prepare(pr, mat)
        keys.append(pr)
    ct = VecZnx(n, cols, res_size).fill_uniform(res_k, rng)
    got = ct.copy()
    ref.glwe_trace_assign_bases(got, res_k, conv_size, key_k, gals, keys)
    conv = VecZnx(n, cols, conv_size)
    for c in range(cols):
        ref.vec_znx_normalize(conv, key_k, 0, c, ct, res_k, c)
    ref.glwe_trace_assign(conv, key_k, gals, keys)
    want = VecZnx(n, cols, res_size)
    for c in range(cols):
        ref.vec_znx_normalize(want, res_k, 0, c, conv, key_k, c)
    assert np.array_equal(got.data, want.data)


def test_pack_bases_with_equal_bases_is_pack_and_a_lone_ciphertext_is_its_trace():
    """pzr_glwe_pack_bases at equal bases is pzr_glwe_pack (P11); with one ciphertext at index 0 and log_gap_out = log_n nothing is packed
    and the closing glwe_trace (skip = 0) is everything: normalize into the keys' base, full trace, normalize back (glwe_trace.rs:107-126)"""
    n, rank, size, k = 64, 1, 3, 13
    ref = RefModule(n)
    rng = seeded(42)
    cols = rank + 1
    log_n = n.bit_length() - 1
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    keys = []
    for _ in gals:
        mat = MatZnx(n, size, rank, cols, size + 1).fill_uniform(k, rng)
        pr = ref.vmp_pmat_alloc(size, rank, cols, size + 1)
        ref.vmp_prepare(pr, mat)
        keys.append(pr)
    data = {j: VecZnx(n, cols, size).fill_uniform(k, rng) for j in (0, 5, 17, 32)}
    r1, r2 = VecZnx(n, cols, size), VecZnx(n, cols, size)
    ref.glwe_pack(r1, k, {j: v.copy() for j, v in data.items()}, 0, gals, keys)
    ref.glwe_pack_bases(r2, k, k, size, {j: v.copy() for j, v in data.items()}, 0, gals, keys)
    assert np.array_equal(r1.data, r2.data)
    lone = data[0]
    r3 = VecZnx(n, cols, size)
    conv, back = VecZnx(n, cols, 4), VecZnx(n, cols, size)
    keys11 = []
    for _ in gals:
        mat = MatZnx(n, 4, rank, cols, 5).fill_uniform(11, rng)
        pr = ref.vmp_pmat_alloc(4, rank, cols, 5)
        ref.vmp_prepare(pr, mat)
        keys11.append(pr)
    ref.glwe_pack_bases(r3, k, 11, 4, {0: lone.copy()}, log_n, gals, keys11)
    for c in range(cols):
        ref.vec_znx_normalize(conv, 11, 0, c, lone, k, c)
    ref.glwe_trace_assign(conv, 11, gals, keys11)
    for c in range(cols):
        ref.vec_znx_normalize(back, k, 0, c, conv, 11, c)
    assert np.array_equal(r3.data, back.data)
