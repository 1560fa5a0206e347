"""GPU parity: libpoulpy_hip.so (through the C ABI) vs the CPU oracle on the same inputs.

Mirrors the reference's cross-backend HAL suite (poulpy-hal/src/test_suite/{vec_znx_dft,svp,vmp,
vec_znx_big}.rs): DFT-domain values are never compared; results are pushed through
idft + normalize and the final i64 limbs must be IDENTICAL (bit-exact).
"""
import os

import numpy as np
import pytest

from poulpy_amd.layouts import MatZnx, ScalarZnx, SvpPPol, VecZnx, VecZnxBig, VecZnxDft, VmpPMat
from tests.helpers import garbage_dft, normalize_all, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from oracle.ref import RefModule
    from poulpy_amd.hal import Module
    cache = {}

    def get(n):
        if n not in cache:
            cache[n] = (RefModule(n), Module(n))
        return cache[n]
    return get


@pytest.mark.parametrize("n", [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072])
def test_dft_idft_roundtrip_and_oracle(mods, n):
    """vec_znx_dft_apply -> vec_znx_idft_apply returns the input exactly, for every plan size."""
    ref, hip = mods(n)
    rng = seeded(n)
    a = VecZnx(n, 2, 3).fill_uniform(40, rng)
    a_before = a.data.copy()
    d = hip.vec_znx_dft_alloc(2, 3)
    for c in range(2):
        hip.vec_znx_dft_apply(1, 0, d, c, a, c)
    assert np.array_equal(a.data, a_before)
    big = hip.vec_znx_big_alloc(2, 3)
    for c in range(2):
        hip.vec_znx_idft_apply(big, c, d, c)
    assert np.array_equal(big.data, a.data)
    # forward spectrum against the oracle's, slot k of the reference <-> natural index bitrev(k)
    m = n // 2
    lg = m.bit_length() - 1
    dr = ref.vec_znx_dft_alloc(2, 3)
    ref.vec_znx_dft_apply(1, 0, dr, 0, a, 0)
    rr = dr.at(0, 1)
    spec_ref = rr[:m] + 1j * rr[m:]
    hh = d.at(0, 1).view(np.complex128)
    idx = np.arange(m)
    rev = np.zeros(m, dtype=np.int64)
    for bit in range(lg):
        rev |= ((idx >> bit) & 1) << (lg - 1 - bit)
    scale = np.abs(spec_ref).max()
    assert np.abs(hh[rev] - spec_ref).max() <= 1e-10 * scale


# n = 8, 16 (round 4): the reference's smallest rings (vmp.rs:67 asserts n >= 8; poulpy-cpu-ref/src/tests.rs:11-23 runs its convolution
# suite at Module::new(8)) - m = 2 x 2 / 2 x 4 on the radix-2 instantiations of the four passes
@pytest.mark.parametrize("n", [8, 16, 256])
@pytest.mark.parametrize("step,offset", [(1, 0), (1, 1), (1, 2), (2, 2), (2, 1), (3, 0)])
def test_vec_znx_dft_apply_step_offset(mods, step, offset, n):
    """poulpy-hal/src/test_suite/vec_znx_dft.rs:360-456: (step, offset) limb selection, untouched limbs."""
    base2k = 12
    ref, hip = mods(n)
    rng = seeded(1000 + step * 10 + offset)
    for a_size in range(1, 6):
        for res_size in range(1, 6):
            a = VecZnx(n, 2, a_size).fill_uniform(base2k, rng)
            pre = rng.standard_normal((res_size, 2, n))
            dr, dh = VecZnxDft(n, 2, res_size, pre.copy()), VecZnxDft(n, 2, res_size, pre.copy())
            for c in range(2):
                ref.vec_znx_dft_apply(step, offset, dr, c, a, 1 - c)
                hip.vec_znx_dft_apply(step, offset, dh, c, a, 1 - c)
            # limbs the reference leaves untouched must be untouched (bitwise) on the GPU too
            steps = -(-a_size // step)
            min_steps = min(res_size, steps)
            for j in range(min_steps):
                if offset + j * step >= a_size:
                    assert np.array_equal(dh.data[j], pre[j])
                    dh.data[j] = 0.0
                    dr.data[j] = 0.0
            br, bh = ref.vec_znx_idft_apply_consume(dr), hip.vec_znx_idft_apply_consume(dh)
            assert np.array_equal(br.data, bh.data)


@pytest.mark.parametrize("n", [8, 16, 256])
def test_svp(mods, n):
    """poulpy-hal/src/test_suite/svp.rs: prepare + apply_dft / dft_to_dft / assign."""
    base2k = 12
    ref, hip = mods(n)
    rng = seeded(7)
    for cols in (1, 2):
        for a_size in range(1, 5):
            for res_size in range(1, 5):
                s = ScalarZnx(n, cols).fill_uniform(base2k, rng)
                pr, ph = SvpPPol(n, cols), SvpPPol(n, cols)
                b = VecZnx(n, cols, a_size).fill_uniform(base2k, rng)
                for c in range(cols):
                    ref.svp_prepare(pr, c, s, c)
                    hip.svp_prepare(ph, c, s, c)
                dr, dh = garbage_dft(n, cols, res_size, rng), garbage_dft(n, cols, res_size, rng)
                for c in range(cols):
                    ref.svp_apply_dft(dr, c, pr, c, b, c)
                    hip.svp_apply_dft(dh, c, ph, c, b, c)
                # dft_to_dft on top
                er, eh = garbage_dft(n, cols, res_size, rng), garbage_dft(n, cols, res_size, rng)
                bdr, bdh = ref.vec_znx_dft_alloc(cols, a_size), hip.vec_znx_dft_alloc(cols, a_size)
                for c in range(cols):
                    ref.vec_znx_dft_apply(1, 0, bdr, c, b, c)
                    hip.vec_znx_dft_apply(1, 0, bdh, c, b, c)
                    ref.svp_apply_dft_to_dft(er, c, pr, c, bdr, c)
                    hip.svp_apply_dft_to_dft(eh, c, ph, c, bdh, c)
                    ref.svp_apply_dft_to_dft_assign(bdr, c, pr, c)
                    hip.svp_apply_dft_to_dft_assign(bdh, c, ph, c)
                for (xr, xh) in ((dr, dh), (er, eh), (bdr, bdh)):
                    nr = normalize_all(ref, ref.vec_znx_idft_apply_consume(xr), base2k)
                    nh = normalize_all(hip, hip.vec_znx_idft_apply_consume(xh), base2k)
                    assert np.array_equal(nr.data, nh.data)


@pytest.mark.parametrize("n", [8, 16, 64, 256])
def test_vmp_apply_dft_to_dft(mods, n):
    """poulpy-hal/src/test_suite/vmp.rs:150-310 incl. limb_offset 1..size_out."""
    base2k = 12
    ref, hip = mods(n)
    rng = seeded(n + 3)
    for cols_in in (1, 2):
        for cols_out in (1, 2):
            for size_in in range(1, 5):
                for size_out in range(1, 5):
                    rows = size_in
                    a = VecZnx(n, cols_in, size_in).fill_uniform(base2k, rng)
                    a0 = a.data.copy()
                    adr, adh = ref.vec_znx_dft_alloc(cols_in, size_in), hip.vec_znx_dft_alloc(cols_in, size_in)
                    for j in range(cols_in):
                        ref.vec_znx_dft_apply(1, 0, adr, j, a, j)
                        hip.vec_znx_dft_apply(1, 0, adh, j, a, j)
                    mat = MatZnx(n, rows, cols_in, cols_out, size_out).fill_uniform(base2k, rng)
                    m0 = mat.data.copy()
                    pr, ph = ref.vmp_pmat_alloc(rows, cols_in, cols_out, size_out), hip.vmp_pmat_alloc(rows, cols_in, cols_out, size_out)
                    ref.vmp_prepare(pr, mat)
                    hip.vmp_prepare(ph, mat)
                    assert np.array_equal(mat.data, m0) and np.array_equal(a.data, a0)
                    rr, rh = garbage_dft(n, cols_out, size_out, rng), garbage_dft(n, cols_out, size_out, rng)
                    ref.vmp_apply_dft_to_dft(rr, adr, pr, 0)
                    hip.vmp_apply_dft_to_dft(rh, adh, ph, 0)
                    nr = normalize_all(ref, ref.vec_znx_idft_apply_consume(rr), base2k)
                    nh = normalize_all(hip, hip.vec_znx_idft_apply_consume(rh), base2k)
                    assert np.array_equal(nr.data, nh.data), (cols_in, cols_out, size_in, size_out)
                    for limb_offset in range(1, size_out):
                        rr, rh = ref.vec_znx_dft_alloc(cols_out, size_out), hip.vec_znx_dft_alloc(cols_out, size_out)
                        ref.vmp_apply_dft_to_dft(rr, adr, pr, limb_offset)
                        hip.vmp_apply_dft_to_dft(rh, adh, ph, limb_offset)
                        nr = normalize_all(ref, ref.vec_znx_idft_apply_consume(rr), base2k)
                        nh = normalize_all(hip, hip.vec_znx_idft_apply_consume(rh), base2k)
                        assert np.array_equal(nr.data, nh.data), (cols_in, cols_out, size_in, size_out, limb_offset)
                    # vmp_apply_dft (family_common.rs:17-54)
                    rr, rh = garbage_dft(n, cols_out, size_out, rng), garbage_dft(n, cols_out, size_out, rng)
                    ref.vmp_apply_dft(rr, a, pr)
                    hip.vmp_apply_dft(rh, a, ph)
                    nr = normalize_all(ref, ref.vec_znx_idft_apply_consume(rr), base2k)
                    nh = normalize_all(hip, hip.vec_znx_idft_apply_consume(rh), base2k)
                    assert np.array_equal(nr.data, nh.data)


@pytest.mark.parametrize("n", [8, 64])
def test_vec_znx_big_normalize_offsets(mods, n):
    """poulpy-hal/src/test_suite/vec_znx_big.rs:785-872: 63-bit inputs, res_offset in [-base2k, base2k],
    same base and cross base."""
    ref, hip = mods(n)
    rng = seeded(11)
    for a_base2k, res_base2k in ((12, 12), (17, 17), (12, 17), (19, 12), (50, 13), (7, 31)):
        for a_size in (1, 2, 3, 5):
            for res_size in (1, 2, 4):
                a = VecZnxBig(n, 2, a_size).fill_uniform(63, rng)
                for off in range(-a_base2k, a_base2k + 1, 3):
                    rr, rh = VecZnx(n, 2, res_size), VecZnx(n, 2, res_size)
                    rr.data[...] = 77
                    rh.data[...] = 77
                    for c in range(2):
                        ref.vec_znx_big_normalize(rr, res_base2k, off, c, a, a_base2k, 1 - c)
                        hip.vec_znx_big_normalize(rh, res_base2k, off, c, a, a_base2k, 1 - c)
                    assert np.array_equal(rr.data, rh.data), (a_base2k, res_base2k, a_size, res_size, off)


@pytest.mark.parametrize("n", [8, 16, 128])
def test_dft_elementwise_ops(mods, n):
    """poulpy-hal/src/test_suite/vec_znx_dft.rs add/sub/copy/zero family, compared after idft+normalize."""
    base2k = 12
    ref, hip = mods(n)
    rng = seeded(5)
    for a_size in (1, 2, 4):
        for b_size in (1, 3):
            for res_size in (1, 2, 5):
                a = VecZnx(n, 2, a_size).fill_uniform(base2k, rng)
                b = VecZnx(n, 2, b_size).fill_uniform(base2k, rng)
                r0 = VecZnx(n, 2, res_size).fill_uniform(base2k, rng)

                def spectra(mod):
                    da, db, dr = mod.vec_znx_dft_alloc(2, a_size), mod.vec_znx_dft_alloc(2, b_size), mod.vec_znx_dft_alloc(2, res_size)
                    for c in range(2):
                        mod.vec_znx_dft_apply(1, 0, da, c, a, c)
                        mod.vec_znx_dft_apply(1, 0, db, c, b, c)
                        mod.vec_znx_dft_apply(1, 0, dr, c, r0, c)
                    return da, db, dr

                ops = [
                    lambda mod, da, db, dr: [mod.vec_znx_dft_add_into(dr, c, da, c, db, 1 - c) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_sub(dr, c, da, c, db, 1 - c) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_add_assign(dr, c, da, 1 - c) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_sub_assign(dr, c, da, 1 - c) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_sub_negate_assign(dr, c, da, 1 - c) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_add_scaled_assign(dr, c, da, c, 1) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_add_scaled_assign(dr, c, da, c, -2) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_copy(2, 1, dr, c, da, c) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_copy(1, 0, dr, c, db, c) for c in range(2)],
                    lambda mod, da, db, dr: [mod.vec_znx_dft_zero(dr, 1)],
                ]
                for k, op in enumerate(ops):
                    outs = []
                    for mod in (ref, hip):
                        da, db, dr = spectra(mod)
                        op(mod, da, db, dr)
                        outs.append(normalize_all(mod, mod.vec_znx_idft_apply_consume(dr), base2k).data)
                    assert np.array_equal(outs[0], outs[1]), (k, a_size, b_size, res_size)


@pytest.mark.parametrize("n", [8, 16, 64])
def test_big_add_small_assign(mods, n):
    ref, hip = mods(n)
    rng = seeded(9)
    for a_size in (1, 3):
        for res_size in (1, 2, 4):
            a = VecZnx(n, 2, a_size).fill_uniform(30, rng)
            r = VecZnxBig(n, 2, res_size).fill_uniform(62, rng)
            rr, rh = r.copy(), r.copy()
            ref.vec_znx_big_add_small_assign(rr, 1, a, 0)
            hip.vec_znx_big_add_small_assign(rh, 1, a, 0)
            assert np.array_equal(rr.data, rh.data)


# ------------------------------------------------------------------------------------------
# batched, device-resident GLWE operations (CoreImpl-level boundary)
# ------------------------------------------------------------------------------------------
def _run_glwe_op(hip, ref, ks, n, rank, rank_out, a_size, a_base2k, key_size, key_base2k, dnum, dsize, res_size, res_base2k, batch,
                 seed, chunk=0, fuse=(True, True), auto=None, in_place=False, pin=False, wide_in=None):
    """auto = (galois element, mode) runs the glwe_automorphism family on top of the key switch.  wide_in = (ciphertext, bits): that ciphertext's
    digits are drawn from +-2^(bits-1) instead of the balanced a_base2k range (an un-normalized input)."""
    from poulpy_amd.hal import GlweOpParams
    rng = seeded(seed)
    cols_a = rank + 1
    cols_in = rank if ks else rank + 1
    cols_out = (rank_out if ks else rank) + 1
    mat = MatZnx(n, dnum, cols_in, cols_out, key_size).fill_uniform(key_base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, cols_in, cols_out, key_size), hip.vmp_pmat_alloc(dnum, cols_in, cols_out, key_size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    a_all = np.empty((batch, a_size, cols_a, n), dtype=np.int64)
    want = np.empty((batch, res_size, cols_out, n), dtype=np.int64)
    for b in range(batch):
        a = VecZnx(n, cols_a, a_size).fill_uniform(a_base2k, rng)
        if wide_in is not None and b == wide_in[0]:
            a.data[...] = rng.integers(-(1 << (wide_in[1] - 1)), 1 << (wide_in[1] - 1), a.data.shape, dtype=np.int64)
        a_all[b] = a.data
        res = VecZnx(n, cols_out, res_size)
        if auto is not None:
            ref.glwe_automorphism(res, res_base2k, a, a_base2k, pr, dsize, key_base2k, auto[0], auto[1])
        elif ks:
            ref.glwe_keyswitch(res, res_base2k, a, a_base2k, pr, dsize, key_base2k)
        else:
            ref.glwe_external_product(res, res_base2k, a, a_base2k, pr, dsize, key_base2k)
        want[b] = res.data
    d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
    d_key = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    if in_place:
        assert a_all.shape == want.shape
        d_res = d_a
    else:
        d_res = hip.device_alloc(want.nbytes)
        hip.lib.pz_memset_d(hip.handle, d_res.ptr, 0x5A, want.nbytes)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=dsize, key_size=key_size, key_base2k=key_base2k, a_size=a_size, a_base2k=a_base2k,
                     res_size=res_size, res_base2k=res_base2k, rank_out=rank_out)
    hip.set_chunk(chunk)
    hip.set_fusion(*fuse)
    if pin:
        hip.pin_key(d_key.ptr, dnum, cols_in, cols_out, key_size)
        hip.glwe_external_product_batched(d_res.ptr, d_a.ptr, d_key.ptr, p, batch) if not ks else None  # first call after pinning
    if auto is not None:
        hip.glwe_automorphism_batched(d_res.ptr, d_a.ptr, d_key.ptr, p, auto[0], auto[1], batch)
    elif ks:
        hip.glwe_keyswitch_batched(d_res.ptr, d_a.ptr, d_key.ptr, p, batch)
    else:
        hip.glwe_external_product_batched(d_res.ptr, d_a.ptr, d_key.ptr, p, batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    if pin:
        hip.unpin_key(d_key.ptr)
    hip.set_chunk(0)
    hip.set_fusion(True, True)
    for buf in ((d_a, d_key) if in_place else (d_a, d_key, d_res)):
        buf.free()
    return got, want


@pytest.mark.parametrize("fuse", [(True, True), (False, False)], ids=["fused", "unfused"])
@pytest.mark.parametrize("rank", [1, 2])
@pytest.mark.parametrize("dsize", [1, 2, 3, 4])
def test_glwe_external_product_batched(mods, rank, dsize, fuse):
    """poulpy-core/src/test_suite/external_product/glwe_ct.rs sweeps rank in {1,2}, dsize 1..max and
    different base2k for input / key / output; here against the oracle's restatement, bit-exact."""
    n = 256
    ref, hip = mods(n)
    for (a_b, k_b, r_b) in ((12, 12, 12), (17, 14, 15)):
        dnum = 5 if dsize == 1 else 2
        got, want = _run_glwe_op(hip, ref, False, n, rank, rank, 5, a_b, 6, k_b, dnum, dsize, 4, r_b, batch=7,
                                 seed=100 * rank + dsize + a_b, chunk=3, fuse=fuse)
        assert np.array_equal(got, want), (rank, dsize, a_b, k_b, r_b)


@pytest.mark.parametrize("fuse", [(True, True), (False, False)], ids=["fused", "unfused"])
@pytest.mark.parametrize("dsize", [1, 2])
def test_glwe_keyswitch_batched(mods, dsize, fuse):
    n = 256
    ref, hip = mods(n)
    for (rank_in, rank_out) in ((1, 1), (2, 1), (1, 2)):
        for (a_b, k_b, r_b) in ((12, 12, 12), (16, 13, 15)):
            got, want = _run_glwe_op(hip, ref, True, n, rank_in, rank_out, 4, a_b, 5, k_b, 3 if dsize == 1 else 2, dsize, 4, r_b,
                                     batch=5, seed=7 + rank_in * 10 + rank_out + dsize, chunk=2, fuse=fuse)
            assert np.array_equal(got, want), (rank_in, rank_out, dsize, a_b, k_b, r_b)


@pytest.mark.parametrize("n", [8, 16])
def test_tiny_rings_batched_glwe_ops(mods, n):
    """N = 8 and 16 (VERDICT r03 missing 3) through the batched entry points: external product (dsize 1-2, mixed bases), key switch
    (rank 1 -> 2), the four automorphism forms, the trace and the tensor relinearization - the five-kernel path on the radix-2 plans
    (no fused / small-ring pipeline at these sizes), against the oracle, bit-exact."""
    from tests.test_gpu_cnv import _run_relinearize
    ref, hip = mods(n)
    for (a_b, k_b, r_b, dsize) in ((12, 12, 12, 1), (16, 13, 15, 1), (13, 13, 13, 2)):
        got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 4, a_b, 5, k_b, 3 if dsize == 1 else 2, dsize, 4, r_b, batch=7, seed=n + a_b + dsize, chunk=3)
        assert np.array_equal(got, want), ("external product", a_b, k_b, r_b, dsize)
        got, want = _run_glwe_op(hip, ref, True, n, 1, 2, 4, a_b, 5, k_b, 3 if dsize == 1 else 2, dsize, 4, r_b, batch=7, seed=n + a_b + dsize + 1)
        assert np.array_equal(got, want), ("key switch", a_b, k_b, r_b, dsize)
    for mode in ("automorphism", "add", "sub", "sub_negate"):
        for gal in (-1, 5, 2 * n - 3, 3):
            got, want = _run_glwe_op(hip, ref, True, n, 1, 1, 3, 12, 4, 12, 3, 1, 3, 12, batch=5, seed=n + gal % 97, auto=(gal, mode),
                                     in_place=(mode == "add"))
            assert np.array_equal(got, want), (mode, gal)
    got, want = _run_relinearize(hip, ref, n, 1, 4, 12, 5, 12, 4, 1, 4, 12, batch=5, seed=n + 77)
    assert np.array_equal(got, want)
    got, want = _run_relinearize(hip, ref, n, 2, 4, 13, 5, 13, 2, 2, 4, 13, batch=3, seed=n + 78)
    assert np.array_equal(got, want)


def test_config2_external_product_n4096(mods):
    """BASELINE configs[1]: GGSW external product, N=2^12, 4 limbs, base2k=17 (batch reduced to what the
    CPU oracle checks in seconds; the batch dimension is embarrassingly parallel)."""
    n = 4096
    ref, hip = mods(n)
    got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 4, 17, 4, 17, 4, 1, 4, 17, batch=16, seed=4096)
    assert np.array_equal(got, want)


def test_metric_config_external_product_n65536(mods):
    """The metric configuration (N=2^16, 8 limbs, rank 1, base2k=12, dnum=8) on a few ciphertexts, with
    the exactness margin of the inverse transform reported (SURVEY.md §7 'exactness margin')."""
    n = 65536
    ref, hip = mods(n)
    got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 8, 12, 8, 12, 8, 1, 8, 12, batch=3, seed=65536)
    assert np.array_equal(got, want)
    # the same call on the probing instantiations of the same kernels: same bits, and the margin as a first-class output
    hip.set_margin_probe(True)
    try:
        got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 8, 12, 8, 12, 8, 1, 8, 12, batch=3, seed=65536)
        margin = hip.get_margin()
    finally:
        hip.set_margin_probe(False)
    assert np.array_equal(got, want)
    assert 0.0 < margin < 1e-4, f"rounding margin at the metric shape: max |x-round(x)| = {margin} (4.8e-6 in round 1)"


def test_pinned_key_n65536(mods):
    """pz_module_pin_key: the cached row-sliced key gives the same bits (external product and key switch), the pin is refused
    twice, and unpinning an unknown key is an error."""
    from poulpy_amd.hal import PoulpyHipError
    n = 65536
    ref, hip = mods(n)
    got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 8, 12, 8, 12, 8, 1, 8, 12, batch=3, seed=11, pin=True)
    assert np.array_equal(got, want)
    got, want = _run_glwe_op(hip, ref, True, n, 1, 1, 8, 12, 8, 12, 8, 1, 8, 12, batch=2, seed=12, pin=True)
    assert np.array_equal(got, want)
    buf = hip.device_alloc(n * 8 * 4)
    hip.pin_key(buf.ptr, 1, 1, 2, 2)
    with pytest.raises(PoulpyHipError):
        hip.pin_key(buf.ptr, 1, 1, 2, 2)
    hip.unpin_key(buf.ptr)
    with pytest.raises(PoulpyHipError):
        hip.unpin_key(buf.ptr)
    buf.free()


@pytest.mark.parametrize("n", [1024, 2048])
def test_pinned_key_on_the_small_ring_pipeline(mods, n):
    """N = 1024 / 2048: a pinned key carries its P'[q1][poly][q2] copy (round 3: these rings permuted the key on every batched call);
    same bits as unpinned for the external product, the key switch and an automorphism form, and the permutation kernel no longer
    runs inside the call."""
    ref, hip = mods(n)
    for (ks, auto) in ((False, None), (True, None), (True, (5, "add"))):
        want_u, ref_u = _run_glwe_op(hip, ref, ks, n, 1, 1, 4, 14, 4, 14, 4, 1, 4, 14, batch=9, seed=n + int(ks), auto=auto)
        assert np.array_equal(want_u, ref_u)
        hip.set_kernel_timing(True)
        try:
            got, want = _run_glwe_op(hip, ref, ks, n, 1, 1, 4, 14, 4, 14, 4, 1, 4, 14, batch=9, seed=n + int(ks), auto=auto, pin=True)
            stats = hip.kernel_stats()
        finally:
            hip.set_kernel_timing(False)
        assert np.array_equal(got, want) and np.array_equal(got, want_u), (n, ks, auto)
        # one k_small_permute launch: pz_module_pin_key's own - none inside the one (key switch, automorphism) or two (external product) calls
        assert stats.get("elementwise", (0, 0.0))[0] == 1, stats


def test_small_shapes_on_the_fused_pipeline(mods):
    """<= 8 polynomials in and out select the 8-slot tile of the middle kernel (8 ciphertexts per tile): 4 limbs at N = 2^16 and
    2^13, external product and key switch, batches that leave the last tile partly empty."""
    for (n, size, batch) in ((65536, 4, 9), (8192, 4, 11), (4096, 3, 17)):
        ref, hip = mods(n)
        got, want = _run_glwe_op(hip, ref, False, n, 1, 1, size, 13, size, 13, size, 1, size, 13, batch=batch, seed=n + size)
        assert np.array_equal(got, want), (n, size, "external product")
        got, want = _run_glwe_op(hip, ref, True, n, 1, 1, size, 13, size, 13, size, 1, size, 13, batch=batch - 2, seed=n + size + 1)
        assert np.array_equal(got, want), (n, size, "key switch")


def test_large_shapes_on_the_fused_pipeline(mods):
    """17..32 polynomials in or out select the 32-slot tile of the middle kernel (two ciphertexts per tile): rank 2 with 8 limbs
    (24 polynomials), rank 1 with 16 limbs (BASELINE configs[4] shape: 32 output polynomials), rank 3 at N = 2^13."""
    for (n, rank, size, dnum, batch) in ((65536, 2, 8, 8, 3), (65536, 1, 16, 16, 3), (8192, 3, 6, 4, 7), (16384, 1, 12, 5, 5)):
        ref, hip = mods(n)
        got, want = _run_glwe_op(hip, ref, False, n, rank, rank, size, 12, size, 12, dnum, 1, size, 12, batch=batch, seed=n + rank + size)
        assert np.array_equal(got, want), (n, rank, size, "external product")
        got, want = _run_glwe_op(hip, ref, True, n, rank, rank, size, 12, size, 12, dnum, 1, size - 1, 12, batch=batch, seed=n + rank + size + 1)
        assert np.array_equal(got, want), (n, rank, size, "key switch")


def test_config3_keyswitch_n65536(mods):
    """BASELINE configs[2]: GLWE key-switch via VmpPMat, N=2^16, 8 limbs (GGLWE rows=8, cols_in=1, cols_out=2)."""
    n = 65536
    ref, hip = mods(n)
    got, want = _run_glwe_op(hip, ref, True, n, 1, 1, 8, 12, 8, 12, 8, 1, 8, 12, batch=2, seed=3)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [32, 64, 512, 2048, 16384])
def test_fused_tail_shapes(mods, n):
    """Fused inverse-pass-1 + normalize against the oracle for every plan family and for output sizes smaller /
    larger than the key size (carry-only limbs, zero-filled limbs)."""
    ref, hip = mods(n)
    for res_size in (2, 4, 6):
        got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 3, 14, 4, 14, 3, 1, res_size, 14, batch=9, seed=n + res_size)
        assert np.array_equal(got, want), (n, res_size)
        got, want = _run_glwe_op(hip, ref, True, n, 2, 1, 3, 14, 4, 14, 3, 1, res_size, 14, batch=3, seed=n + res_size + 1)
        assert np.array_equal(got, want), (n, res_size)


def test_tall_plan_external_product_n65536(monkeypatch):
    """The alternative four-step split at N = 2^16 (m1 = 128, m2 = 256: k_mid<2>, two ciphertexts per tile; the default is
    256 x 128 with k_mid128<4>) gives the same bits."""
    from oracle.ref import RefModule
    from poulpy_amd.hal import Module
    monkeypatch.setenv("POULPY_DBG_SPLIT", "t")
    n = 65536
    ref, hip = RefModule(n), Module(n)
    monkeypatch.delenv("POULPY_DBG_SPLIT")
    got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 8, 12, 8, 12, 8, 1, 8, 12, batch=5, seed=99)
    assert np.array_equal(got, want)
    got, want = _run_glwe_op(hip, ref, True, n, 1, 1, 8, 12, 8, 12, 8, 1, 6, 12, batch=3, seed=98)
    assert np.array_equal(got, want)
    hip.close()


# ------------------------------------------------------------------------------------------
# automorphism family (HAL ops + CoreImpl glwe_automorphism*) and ggsw_external_product
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [32, 256, 1024, 65536])
def test_vec_znx_automorphism(mods, n):
    """poulpy-hal/src/test_suite/vec_znx.rs automorphism tests: p in {-5, 5} there; more Galois elements and ragged
    sizes here (res limbs beyond a.size zeroed, garbage-prefilled outputs)."""
    ref, hip = mods(n)
    rng = seeded(n + 1)
    for p in (-5, 5, 3, -1, 2 * n - 1, 2 * n + 7, 5 ** 7):
        for (a_size, res_size) in ((3, 3), (2, 4), (4, 1)):
            a = VecZnx(n, 2, a_size).fill_uniform(50, rng)
            rr = VecZnx(n, 3, res_size).fill_uniform(60, rng)
            rh = rr.copy()
            ref.vec_znx_automorphism(p, rr, 2, a, 1)
            hip.vec_znx_automorphism(p, rh, 2, a, 1)
            assert np.array_equal(rr.data, rh.data), (p, a_size, res_size)
            br, bh = VecZnxBig(n, 2, a_size), VecZnxBig(n, 2, a_size)
            br.data[:] = a.data
            bh.data[:] = a.data
            ref.vec_znx_big_automorphism_assign(p, br, 0)
            hip.vec_znx_big_automorphism_assign(p, bh, 0)
            assert np.array_equal(br.data, bh.data), (p, a_size)
            ar, ah = a.copy(), a.copy()
            ref.vec_znx_automorphism_assign(p, ar, 1)
            hip.vec_znx_automorphism_assign(p, ah, 1)
            assert np.array_equal(ar.data, ah.data)


def test_vec_znx_automorphism_rejects_even_and_alias(mods):
    from poulpy_amd.hal import PoulpyHipError
    n = 64
    _, hip = mods(n)
    a = VecZnx(n, 1, 2)
    r = VecZnx(n, 1, 2)
    with pytest.raises(PoulpyHipError):
        hip.vec_znx_automorphism(4, r, 0, a, 0)
    with pytest.raises(PoulpyHipError):
        hip.vec_znx_automorphism(3, a, 0, a, 0)


@pytest.mark.parametrize("fuse", [(True, True), (False, False)], ids=["fused", "unfused"])
@pytest.mark.parametrize("mode", ["automorphism", "add", "sub", "sub_negate"])
def test_glwe_automorphism_batched(mods, mode, fuse):
    """poulpy-core/src/test_suite/automorphism/glwe_ct.rs sweeps rank and dsize for p = -5; here every variant of
    automorphism/glwe_ct.rs:51-275 against the oracle's restatement, bit-exact, incl. cross-base2k and dsize 2."""
    n = 256
    ref, hip = mods(n)
    for rank in (1, 2):
        for (a_b, k_b, r_b, dsize) in ((12, 12, 12, 1), (16, 13, 15, 1), (13, 13, 13, 2)):
            for gal in (-5, 3 ** 5):
                got, want = _run_glwe_op(hip, ref, True, n, rank, rank, 4, a_b, 5, k_b, 3 if dsize == 1 else 2, dsize, 4, r_b,
                                         batch=5, seed=17 + rank + dsize, chunk=2, fuse=fuse, auto=(gal, mode))
                assert np.array_equal(got, want), (mode, rank, a_b, k_b, r_b, dsize, gal)


@pytest.mark.parametrize("mode", ["automorphism", "add"])
def test_glwe_automorphism_assign_form(mods, mode):
    """glwe_automorphism_assign / _add_assign (glwe_ct.rs:74-94, :142-183): res == a."""
    n = 512
    ref, hip = mods(n)
    got, want = _run_glwe_op(hip, ref, True, n, 1, 1, 4, 14, 4, 14, 4, 1, 4, 14, batch=6, seed=99, chunk=4, auto=(-5, mode),
                             in_place=True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("mode", ["automorphism", "sub_negate"])
def test_config5_shape_automorphism_n65536(mods, mode):
    """CKKS-rotate shape of BASELINE configs[4] at the FFT64-representable base: N=2^16, 8 limbs, rank 1, p = 5."""
    n = 65536
    ref, hip = mods(n)
    got, want = _run_glwe_op(hip, ref, True, n, 1, 1, 8, 12, 8, 12, 8, 1, 8, 12, batch=2, seed=5, auto=(5, mode))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("in_place", [False, True], ids=["out-of-place", "in-place"])
@pytest.mark.parametrize("n", [8192, 65536])
def test_plain_automorphism_spectral_form(mods, n, in_place):
    """glwe_automorphism (res = phi(normalize(big)), glwe_ct.rs:51-72) for Galois elements = 1 mod 4 on the 128-point-row plans: the
    permutation is folded into the middle kernel's spectrum position; the tail undoes phi's signs in front of the carry chain and puts
    them back on the digits, so no permutation pass runs over the result.  5, 5^k, 2N - 3 (= 1 mod 4), ragged limbs, rank 2; 3 and -1
    (= 3 mod 4) keep the key switch + signed permutation pass."""
    ref, hip = mods(n)
    k = 12
    cases = [(1, 5, 4, 4, 4, 4, 3), (1, pow(5, 9, 2 * n), 3, 4, 3, 3, 2), (2, 2 * n - 3, 3, 3, 3, 3, 2), (1, 3, 4, 4, 4, 4, 2), (1, -1, 4, 3, 3, 4, 2),
             (1, 5, 4, 3, 3, 4, 2)]
    for (rank, gal, a_size, key_size, dnum, res_size, batch) in cases:
        if in_place and a_size != res_size:
            continue
        got, want = _run_glwe_op(hip, ref, True, n, rank, rank, a_size, k, key_size, k, dnum, 1, res_size, k, batch, seed=700 + rank + a_size + batch,
                                 auto=(gal, "automorphism"), in_place=in_place)
        assert np.array_equal(got, want), (rank, gal, a_size, key_size, res_size)


def test_ggsw_external_product(mods):
    """external_product/ggsw.rs:54-58: a loop of GLWE external products over the (row, column) entries of a GGSW."""
    from poulpy_amd.hal import GlweOpParams
    n, rank, dnum_a, size, base2k, dnum = 256, 1, 3, 4, 13, 4
    ref, hip = mods(n)
    rng = seeded(77)
    cols = rank + 1
    mat = MatZnx(n, dnum, cols, cols, size).fill_uniform(base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, size), hip.vmp_pmat_alloc(dnum, cols, cols, size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    a = MatZnx(n, dnum_a, cols, cols, size).fill_uniform(base2k, rng)   # the GGSW being multiplied
    want = np.empty_like(a.data)
    flat_a = a.data.reshape(dnum_a * cols, size, cols, n)
    flat_w = want.reshape(dnum_a * cols, size, cols, n)
    for e in range(dnum_a * cols):
        ct = VecZnx(n, cols, size)
        ct.data[:] = flat_a[e]
        res = VecZnx(n, cols, size)
        ref.glwe_external_product(res, base2k, ct, base2k, pr, 1, base2k)
        flat_w[e] = res.data
    d_a = hip.device_alloc(a.data.nbytes).upload(a.data)
    d_key = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    d_res = hip.device_alloc(want.nbytes)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=rank)
    hip.ggsw_external_product(d_res.ptr, d_a.ptr, dnum_a, d_key.ptr, p)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_a, d_key, d_res):
        buf.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,dnum,size,key_dnum,key_size,base2k,dsize,count", [
    (256, 1, 3, 4, 4, 4, 13, 1, 1),     # five-kernel path (N below the fused plans)
    (256, 2, 2, 3, 3, 5, 13, 1, 2),     # rank 2: two tensor keys, body in column 1 / 2; two GGSWs
    (4096, 1, 2, 4, 4, 4, 14, 1, 3),    # fused pipeline, body column 1
    (4096, 2, 2, 3, 3, 4, 13, 1, 1),    # fused pipeline, rank 2: body columns 1 and 2 (split column launches)
    (8192, 3, 1, 2, 2, 3, 12, 1, 1),    # rank 3
    (1024, 1, 2, 4, 2, 5, 12, 2, 2),    # dsize 2
    (65536, 1, 2, 8, 8, 8, 12, 1, 1),   # metric-shape ring
])
def test_ggsw_expand_row_batched(mods, n, rank, dnum, size, key_dnum, key_size, base2k, dsize, count):
    """conversion/gglwe_to_ggsw.rs:116-268 in place on `count` contiguous device GGSWs vs the oracle's restatement; column 0
    entries must come back untouched."""
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(900 + n + rank)
    cols = rank + 1
    keys_r, keys_d = [], []
    for c in range(rank):
        mat = MatZnx(n, key_dnum, rank, cols, key_size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(key_dnum, rank, cols, key_size), hip.vmp_pmat_alloc(key_dnum, rank, cols, key_size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        keys_r.append(pr)
        keys_d.append(hip.device_alloc(ph.data.nbytes).upload(ph.data))
    ggsws = [MatZnx(n, dnum, cols, cols, size).fill_uniform(base2k, rng) for _ in range(count)]
    flat = np.stack([g.data for g in ggsws])
    d = hip.device_alloc(flat.nbytes).upload(flat)
    for g in ggsws:
        ref.ggsw_expand_row(g, base2k, keys_r, dsize, base2k)
    want = np.stack([g.data for g in ggsws])
    p = GlweOpParams(rank=rank, dnum=key_dnum, dsize=dsize, key_size=key_size, key_base2k=base2k, a_size=size, a_base2k=base2k,
                     res_size=size, res_base2k=base2k, rank_out=rank)
    hip.ggsw_expand_row_batched(d.ptr, dnum, [k.ptr for k in keys_d], p, count)
    hip.sync()
    got = d.download(np.int64, want.size).reshape(want.shape)
    for buf in keys_d + [d]:
        buf.free()
    assert np.array_equal(got[:, :, 0], flat[:, :, 0])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,cols_in", [(256, 1, 1), (4096, 2, 2), (4096, 1, 3)])
def test_ggsw_from_gglwe_batched(mods, n, rank, cols_in):
    """conversion/gglwe_to_ggsw.rs:32-61: strided copy of the a.at(row, 0) entries + ggsw_expand_row, two GGLWEs per call."""
    from poulpy_amd.hal import GlweOpParams
    dnum, size, key_dnum, key_size, base2k, count = 2, 3, 3, 4, 13, 2
    ref, hip = mods(n)
    rng = seeded(950 + n + rank)
    cols = rank + 1
    keys_r, keys_d = [], []
    for c in range(rank):
        mat = MatZnx(n, key_dnum, rank, cols, key_size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(key_dnum, rank, cols, key_size), hip.vmp_pmat_alloc(key_dnum, rank, cols, key_size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        keys_r.append(pr)
        keys_d.append(hip.device_alloc(ph.data.nbytes).upload(ph.data))
    gglwes = [MatZnx(n, dnum, cols_in, cols, size).fill_uniform(base2k, rng) for _ in range(count)]
    want = []
    for a in gglwes:
        g = MatZnx(n, dnum, cols, cols, size).fill_uniform(base2k, rng)   # stale contents must be overwritten
        ref.ggsw_from_gglwe(g, base2k, a, keys_r, 1, base2k)
        want.append(g.data)
    want = np.stack(want)
    flat_a = np.stack([a.data for a in gglwes])
    d_a = hip.device_alloc(flat_a.nbytes).upload(flat_a)
    d_g = hip.device_alloc(want.nbytes).upload(rng.integers(-9, 9, want.shape, dtype=np.int64))
    p = GlweOpParams(rank=rank, dnum=key_dnum, dsize=1, key_size=key_size, key_base2k=base2k, a_size=size, a_base2k=base2k,
                     res_size=size, res_base2k=base2k, rank_out=rank)
    hip.ggsw_from_gglwe_batched(d_g.ptr, d_a.ptr, cols_in, dnum, [k.ptr for k in keys_d], p, count)
    hip.sync()
    got = d_g.download(np.int64, want.size).reshape(want.shape)
    for buf in keys_d + [d_a, d_g]:
        buf.free()
    assert np.array_equal(got, want)


# ------------------------------------------------------------------------------------------
# SURVEY.md 8f rank 2 / BASELINE configs[3]: CGGI blind rotation on a batch of LWE ciphertexts
# ------------------------------------------------------------------------------------------
def _run_blind_rotation(hip, ref, n, rank, n_lwe, block_size, dnum, brk_size, res_size, base2k, batch, seed, fuse=(True, True), lut_size=None):
    from poulpy_amd.hal import BlindRotationParams
    rng = seeded(seed)
    cols = rank + 1
    lut_size = res_size if lut_size is None else lut_size
    lut = VecZnx(n, 1, lut_size).fill_uniform(base2k, rng)
    brk_r = np.empty((n_lwe, n * dnum * cols * cols * brk_size), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        mat = MatZnx(n, dnum, cols, cols, brk_size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, brk_size), hip.vmp_pmat_alloc(dnum, cols, cols, brk_size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        brk_r[i] = pr.data.reshape(-1)
        brk_h[i] = ph.data.reshape(-1)
    lwe = rng.integers(-n, n, (batch, n_lwe + 1), dtype=np.int64)   # mod_switch_2n output range
    lwe[0, 1] = 0        # X^0 - 1 = 0: a coefficient that contributes nothing
    lwe[-1, 0] = n - 1
    xpa = ref.blind_rotation_x_pow_a() if block_size > 1 else np.zeros((1, 1))
    want = np.empty((batch, res_size, cols, n), dtype=np.int64)
    for b in range(batch):
        res = VecZnx(n, cols, res_size)
        ref.blind_rotation_execute(res, base2k, np.ascontiguousarray(lwe[b]), lut, brk_r, dnum, brk_size, block_size, xpa)
        want[b] = res.data
    d_lwe = hip.device_alloc(lwe.nbytes).upload(lwe)
    d_lut = hip.device_alloc(lut.data.nbytes).upload(lut.data)
    d_brk = hip.device_alloc(brk_h.nbytes).upload(brk_h)
    d_res = hip.device_alloc(want.nbytes)
    hip.lib.pz_memset_d(hip.handle, d_res.ptr, 0x33, want.nbytes)
    p = BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block_size, dnum=dnum, brk_size=brk_size, base2k=base2k,
                            res_size=res_size, lut_size=lut_size)
    hip.set_fusion(*fuse)
    hip.blind_rotation_execute_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, p, batch)
    hip.sync()
    hip.set_fusion(True, True)
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_lwe, d_lut, d_brk, d_res):
        buf.free()
    return got, want


@pytest.mark.parametrize("fuse", [(True, True), (False, False)], ids=["one-kernel", "composed"])
@pytest.mark.parametrize("block_size", [1, 3])
@pytest.mark.parametrize("rank", [1, 2])
def test_blind_rotation_batched(mods, rank, block_size, fuse):
    """poulpy-bin-fhe/src/blind_rotation/tests/test_suite: block sizes 1 and 7 at n_glwe 256; here both variants of
    algorithm.rs against the oracle's restatement on random key material, bit-exact (7 LWE coefficients: a trailing
    partial block, which chunks_exact drops)."""
    n = 256
    ref, hip = mods(n)
    got, want = _run_blind_rotation(hip, ref, n, rank, 7, block_size, 2, 3, 2, 14, batch=5, seed=31 + rank, fuse=fuse)
    assert np.array_equal(got, want)


def test_blind_rotation_shapes(mods):
    """Shapes that exercise every branch of the one-kernel path: res limbs beyond the key precision (zeroed), more key limbs
    than result limbs (carry-only first step), odd batch (half-empty last tile), lut shorter than res, rank 3 (the reference
    bench shape, 4 x 8 product), n = 256 / 512 / 1024."""
    for (n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch) in ((256, 1, 6, 2, 1, 2, 3, 12, 3), (512, 3, 6, 3, 1, 2, 1, 18, 5),
                                                            (1024, 1, 8, 4, 3, 3, 3, 13, 3), (256, 2, 4, 2, 2, 2, 2, 15, 1),
                                                            (512, 1, 5, 5, 1, 3, 2, 16, 2)):
        ref, hip = mods(n)
        got, want = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=batch, seed=n + rank + blk)
        assert np.array_equal(got, want), (n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch)


@pytest.mark.parametrize("block_size", [1, 7])
def test_reference_blind_rotation_test_shape(mods, block_size):
    """the shape of poulpy-bin-fhe's own blind-rotation tests (blind_rotation/tests/fft64_ref.rs:24-35 `standard` / `block_binary`,
    test_suite/generic_blind_rotation.rs:37-44): N = 512, n_lwe = 224, rank 1, base2k 19, key of 3 limbs x 2 rows, result 2 limbs, table
    1 limb, block size 1 / 7 — all 224 coefficients, on random key material."""
    n = 512
    ref, hip = mods(n)
    got, want = _run_blind_rotation(hip, ref, n, 1, 224, block_size, 2, 3, 2, 19, batch=3, seed=224 + block_size, lut_size=1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,block_size", [(1024, 7), (1024, 1), (2048, 4)])
def test_config4_shape_blind_rotation(mods, n, block_size):
    """BASELINE configs[3] shape (gate-bootstrap blind rotation, N = 2^10 class parameters: rank 1, 2 limbs of 17 bits,
    dnum 2, block size 7) on a short LWE (28 coefficients) so that the CPU oracle finishes in seconds."""
    ref, hip = mods(n)
    got, want = _run_blind_rotation(hip, ref, n, 1, 28, block_size, 2, 2, 2, 17, batch=9, seed=n + block_size)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,n_lwe,blk,dnum,bsz,rsz,k,batch", [
    (4096, 1, 9, 3, 2, 2, 2, 14, 11),      # 4 polynomials in / out: eight ciphertexts per tile, ragged last tile
    (4096, 1, 15, 7, 3, 3, 3, 13, 9),      # circuit-bootstrapping key layout, two blocks + a trailing partial block
    (8192, 2, 4, 2, 2, 3, 2, 13, 5),       # 6 in, 9 out: four ciphertexts x 16 slots; key limbs > result limbs
    (4096, 3, 6, 3, 3, 3, 3, 12, 3),       # rank 3: 12 x 12
    (4096, 3, 4, 2, 3, 5, 3, 12, 2),       # 20 output polynomials: two ciphertexts x 32 slots
    (32768, 1, 4, 4, 2, 2, 3, 12, 2),      # 128 x 128 plan, result limbs beyond the key precision
    (4096, 1, 6, 3, 3, 2, 2, 14, 4),       # dnum > res_size: not covered by the block step on the pipeline (composed path)
])
def test_blind_rotation_block_step_on_the_glwe_pipeline(mods, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch):
    """execute_block_binary (algorithm.rs:265-368) for ring degrees whose plan has 128-point rows: every LWE block is one pass of the
    three-kernel pipeline (k_mid128<.., BR>); against the oracle, and against the composed path bit for bit."""
    ref, hip = mods(n)
    got, want = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=batch, seed=n + rank + blk + dnum)
    assert np.array_equal(got, want), (n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch)
    got2, _ = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=batch, seed=n + rank + blk + dnum, fuse=(False, False))
    assert np.array_equal(got2, want)


@pytest.mark.parametrize("a_b,k_b,r_b", [(16, 13, 15), (13, 13, 15), (15, 13, 13), (12, 17, 12)],
                         ids=["in-key-out all differ", "result in another base", "input in another base", "key in the larger base"])
@pytest.mark.parametrize("n", [1024, 2048])
def test_small_ring_mixed_bases(mods, n, a_b, k_b, r_b):
    """N = 1024 / 2048 with the base2k triples poulpy-core's tests use (input, key, result all different: test_suite/external_product/
    glwe_ct.rs:33-36, keyswitch/glwe_ct.rs:33-36, automorphism/glwe_ct.rs:35-38): the input is re-expressed in the key's base in front of
    the two-kernel pipeline, a result in another base goes through balanced key-base digits and one cross-base pass; against the oracle
    and the five-kernel path bit for bit.  The automorphism family rides along when only the input differs."""
    ref, hip = mods(n)
    cases = [(False, 1, 1, 4, 4, 3, 4, 7, None), (True, 1, 1, 3, 4, 3, 3, 9, None), (True, 2, 1, 3, 3, 3, 4, 4, None), (False, 2, 2, 2, 3, 2, 2, 3, None)]
    if k_b == r_b:
        cases += [(True, 1, 1, 3, 4, 3, 3, 5, (5, "add")), (True, 1, 1, 4, 3, 3, 4, 3, (-1, "automorphism")), (True, 2, 2, 3, 3, 3, 3, 2, (3, "sub_negate"))]
    for (ks, rank, rank_out, a_size, key_size, dnum, res_size, batch, auto) in cases:
        args = (ks, n, rank, rank_out, a_size, a_b, key_size, k_b, dnum, 1, res_size, r_b, batch)
        hip.set_small_path(True)
        got, want = _run_glwe_op(hip, ref, *args, seed=500 + rank + a_size + batch, auto=auto)
        assert np.array_equal(got, want), (ks, rank, rank_out, a_size, key_size, res_size, auto)
        hip.set_small_path(False)
        try:
            got2, _ = _run_glwe_op(hip, ref, *args, seed=500 + rank + a_size + batch, auto=auto)
        finally:
            hip.set_small_path(True)
        assert np.array_equal(got2, want)


@pytest.mark.parametrize("in_place", [False, True], ids=["out-of-place", "in-place"])
@pytest.mark.parametrize("mode", ["automorphism", "add", "sub", "sub_negate"])
@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_small_ring_automorphism_family(mods, n, mode, in_place):
    """N = 1024 / 2048 / 4096: the glwe_automorphism family on the two-kernel pipeline (phi as an index / sign map in k_small_inv's carry-chain
    stage: on the normalized digits for the plain form, on the big value before +- a for the others) against the oracle and the
    five-kernel path bit for bit; Galois elements -1, 5, 5^(2^k) (the trace's), 3; ragged limb counts; in place (res == a)."""
    k = 14
    ref, hip = mods(n)
    cases = [(1, -1, 4, 4, 4, 4, 9), (1, 5, 3, 4, 3, 3, 5), (2, pow(5, 16, 2 * n), 3, 3, 3, 3, 4), (1, 3, 4, 3, 4, 4, 3), (1, 2 * n - 5, 2, 4, 2, 3, 2)]
    for (rank, gal, a_size, key_size, dnum, res_size, batch) in cases:
        if in_place and a_size != res_size:
            continue
        hip.set_small_path(True)
        got, want = _run_glwe_op(hip, ref, True, n, rank, rank, a_size, k, key_size, k, dnum, 1, res_size, k, batch, seed=300 + rank + a_size + batch,
                                 auto=(gal, mode), in_place=in_place)
        assert np.array_equal(got, want), (rank, gal, a_size, key_size, res_size)
        hip.set_small_path(False)
        try:
            got2, _ = _run_glwe_op(hip, ref, True, n, rank, rank, a_size, k, key_size, k, dnum, 1, res_size, k, batch,
                                   seed=300 + rank + a_size + batch, auto=(gal, mode), in_place=in_place)
        finally:
            hip.set_small_path(True)
        assert np.array_equal(got2, want)


@pytest.mark.parametrize("n,rank,n_lwe,blk,dnum,bsz,rsz,k,batch", [
    (2048, 1, 15, 7, 3, 3, 3, 13, 9),      # the gate-bootstrap key layout at N = 2048: two blocks + a trailing partial block, ragged batch
    (2048, 1, 8, 4, 2, 2, 3, 14, 3),       # result limbs beyond the key precision (zeroed)
    (2048, 1, 6, 3, 2, 4, 2, 12, 17),      # key limbs > result limbs (carry-only first steps), three XCD rounds
    (1024, 2, 6, 3, 3, 3, 3, 13, 5),       # rank 2 at N = 1024 (the circuit-bootstrapping shape: 9 inputs, 9 outputs)
    (1024, 2, 4, 2, 4, 3, 4, 12, 2),       # 12 inputs
    (2048, 2, 4, 2, 2, 2, 2, 15, 4),       # rank 2 at N = 2048
    (2048, 3, 6, 3, 2, 1, 2, 16, 3),       # rank 3, one key limb
    (2048, 1, 6, 3, 3, 2, 3, 13, 4),       # more accumulator limbs transformed than the key has: the inverse kernel cannot chain the next forward
    (4096, 1, 6, 3, 3, 2, 2, 14, 4),       # N = 4096 with dnum > res_size: neither pipeline form applies (composed path, per-op kernels)
])
def test_blind_rotation_small_ring_transforms(mods, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch):
    """execute_block_binary (algorithm.rs:265-368) at N = 1024 / 2048 off the one-kernel path: k_small_fwd | block step on S-ordered spectra
    with the block's keys re-ordered | k_small_inv (no product: inverse transform + accumulator + carry chain) — against the oracle, and
    against the per-op composition bit for bit."""
    ref, hip = mods(n)
    hip.set_small_path(True)
    got, want = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=batch, seed=n + rank + blk + dnum)
    assert np.array_equal(got, want), (n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch)
    hip.set_small_path(False)
    try:
        got2, _ = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=batch, seed=n + rank + blk + dnum)
    finally:
        hip.set_small_path(True)
    assert np.array_equal(got2, want)


@pytest.mark.parametrize("ks,rank,rank_out,a_size,key_size,dnum,res_size,batch,in_place", [
    (False, 1, 1, 4, 4, 4, 4, 19, False),     # BASELINE configs[1] shape, ragged batch (19 = 2 x 8 + 3: partly filled XCD slots)
    (False, 1, 1, 4, 4, 4, 4, 8, True),       # in place
    (True, 1, 1, 4, 4, 4, 4, 5, False),       # key switch: body added before the carry chain
    (True, 1, 1, 3, 3, 3, 3, 9, True),        # odd number of limbs (a pair with one polynomial), in place
    (False, 1, 1, 2, 4, 2, 2, 3, False),      # key limbs below the result: carry-only first steps
    (False, 1, 1, 3, 2, 3, 4, 4, False),      # result limbs beyond the key precision: zero limbs
    (False, 2, 2, 2, 3, 2, 3, 5, False),      # rank 2: three output columns per ciphertext
    (True, 2, 1, 4, 4, 4, 2, 6, False),       # rank 2 -> 1 key switch
    (True, 1, 3, 1, 1, 1, 1, 10, False),      # single limb everywhere, four output columns
])
@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_small_ring_two_kernel_pipeline(mods, n, ks, rank, rank_out, a_size, key_size, dnum, res_size, batch, in_place):
    """N = 1024 / 2048 / 4096: k_small_fwd + k_small_inv (whole polynomials in LDS, the spectra cross HBM once) against the oracle, and
    the other pipeline (three kernels at N = 4096, five below) on the same inputs bit for bit."""
    k = 17
    ref, hip = mods(n)
    hip.set_small_path(True)
    got, want = _run_glwe_op(hip, ref, ks, n, rank, rank_out, a_size, k, key_size, k, dnum, 1, res_size, k, batch, seed=900 + a_size + key_size + batch,
                             in_place=in_place)
    assert np.array_equal(got, want)
    hip.set_small_path(False)
    try:
        got3, _ = _run_glwe_op(hip, ref, ks, n, rank, rank_out, a_size, k, key_size, k, dnum, 1, res_size, k, batch, seed=900 + a_size + key_size + batch,
                               in_place=in_place)
    finally:
        hip.set_small_path(True)
    assert np.array_equal(got3, want)


@pytest.mark.parametrize("ks", [False, True], ids=["external_product", "keyswitch"])
@pytest.mark.parametrize("fuse", [(True, True), (False, False)], ids=["pipeline", "five-kernel"])
def test_largest_ring_degree(mods, ks, fuse):
    """N = 131072, the largest ring degree the module accepts (m = 256 x 256: the pipeline runs `k_mid<2>`, two ciphertexts per tile):
    external product and key switch against the oracle, ragged batch."""
    n = 131072
    ref, hip = mods(n)
    got, want = _run_glwe_op(hip, ref, ks, n, 1, 1, 3, 12, 3, 12, 3, 1, 3, 12, batch=3, seed=131 + ks, fuse=fuse)
    assert np.array_equal(got, want)


# ------------------------------------------------------------------------------------------
# seeded shape sweep over the batched entry points (every plan family, fused and five-kernel paths)
# ------------------------------------------------------------------------------------------
def test_seeded_shape_sweep(mods):
    """40 random shapes (fixed seed): N in 2^9..2^16, rank 1-2, 1-8 limbs for input / key / output, dnum, dsize 1-2, same or
    different base2k, ragged batches, external product / key switch / automorphism family, against the oracle, bit-exact."""
    import os
    rng = np.random.default_rng(int(os.environ.get("POULPY_SWEEP_SEED", "20260101")))   # other seeds / longer sweeps by hand
    for case in range(int(os.environ.get("POULPY_SWEEP_CASES", "40"))):
        n = int(2 ** rng.integers(9, 17))
        rank = int(rng.integers(1, 3))
        a_size, key_size, res_size = (int(x) for x in rng.integers(1, 9 if n < 65536 else 7, 3))
        dsize = int(rng.integers(1, 3))
        dnum = int(rng.integers(1, max(2, (a_size + dsize - 1) // dsize + 1)))
        same = bool(rng.integers(0, 3))                      # two thirds same base (the fused pipeline), one third mixed
        key_b = int(rng.integers(10, 15))
        a_b, res_b = (key_b, key_b) if same else (int(rng.integers(9, 18)), int(rng.integers(9, 18)))
        batch = int(rng.integers(1, 12)) if n >= 16384 else int(rng.integers(1, 40))
        kind = int(rng.integers(0, 4))
        chunk = int(rng.integers(0, batch + 1)) if rng.integers(0, 2) else 0     # waves of `chunk` ciphertexts, or one wave
        fuse = (True, True) if rng.integers(0, 4) else (False, False)             # a quarter on the five-kernel path
        ref, hip = mods(n)
        seed = 7000 + case
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(dict(case=case, n=n, rank=rank, a_size=a_size, key_size=key_size, res_size=res_size, dsize=dsize, dnum=dnum, a_b=a_b,
                       key_b=key_b, res_b=res_b, batch=batch, kind=kind), flush=True)
        if kind == 0:
            got, want = _run_glwe_op(hip, ref, False, n, rank, rank, a_size, a_b, key_size, key_b, dnum, dsize, res_size, res_b, batch, seed,
                                     chunk=chunk, fuse=fuse, pin=bool(rng.integers(0, 2)))
        elif kind == 1:
            rank_out = int(rng.integers(1, 3))
            got, want = _run_glwe_op(hip, ref, True, n, rank, rank_out, a_size, a_b, key_size, key_b, dnum, dsize, res_size, res_b, batch, seed,
                                     chunk=chunk, fuse=fuse)
        else:
            mode = ["automorphism", "add", "sub", "sub_negate"][int(rng.integers(0, 4))]
            gal = int(rng.choice([-5, 5, 25, 2 * n - 1, 3]))
            got, want = _run_glwe_op(hip, ref, True, n, rank, rank, a_size, a_b, key_size, key_b, dnum, dsize, res_size, res_b, batch, seed,
                                     chunk=chunk, fuse=fuse, auto=(gal, mode))
        assert np.array_equal(got, want), dict(case=case, n=n, rank=rank, a_size=a_size, key_size=key_size, res_size=res_size, dsize=dsize,
                                               dnum=dnum, a_b=a_b, key_b=key_b, res_b=res_b, batch=batch, kind=kind, chunk=chunk, fuse=fuse)


def test_seeded_shape_sweep_wide_limbs(mods):
    """The same sweep with 9 - 16 limbs on at least one of input / key / output, N <= 2^14 (the sweep above caps limbs at
    8): 17 - 32 polynomials per ciphertext run the 32-slot tile of the middle kernel (k_mid128 / k_mid128r<2,32,..>, one register slot or a
    ring of three for the key rows), more than 32 (rank 2 with > 10 limbs) the five-kernel path; N < 2^12 the small-ring and per-op paths."""
    import os
    rng = np.random.default_rng(int(os.environ.get("POULPY_SWEEP_SEED", "20260404")))
    for case in range(int(os.environ.get("POULPY_SWEEP_CASES", "24"))):
        n = int(2 ** rng.integers(9, 15))
        rank = int(rng.integers(1, 3))
        sizes = [int(x) for x in rng.integers(1, 17, 3)]
        sizes[int(rng.integers(0, 3))] = int(rng.integers(9, 17))     # at least one container beyond 8 limbs
        a_size, key_size, res_size = sizes
        dsize = int(rng.integers(1, 3))
        dnum = int(rng.integers(1, max(2, (a_size + dsize - 1) // dsize + 1)))
        same = bool(rng.integers(0, 3))
        key_b = int(rng.integers(10, 14))
        a_b, res_b = (key_b, key_b) if same else (int(rng.integers(9, 16)), int(rng.integers(9, 16)))
        batch = int(rng.integers(1, 10)) if n >= 8192 else int(rng.integers(1, 24))
        kind = int(rng.integers(0, 4))
        chunk = int(rng.integers(0, batch + 1)) if rng.integers(0, 2) else 0
        fuse = (True, True) if rng.integers(0, 4) else (False, False)
        ref, hip = mods(n)
        seed = 7500 + case
        desc = dict(case=case, n=n, rank=rank, a_size=a_size, key_size=key_size, res_size=res_size, dsize=dsize, dnum=dnum, a_b=a_b, key_b=key_b,
                    res_b=res_b, batch=batch, kind=kind, chunk=chunk, fuse=fuse)
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(desc, flush=True)
        if kind == 0:
            got, want = _run_glwe_op(hip, ref, False, n, rank, rank, a_size, a_b, key_size, key_b, dnum, dsize, res_size, res_b, batch, seed,
                                     chunk=chunk, fuse=fuse, pin=bool(rng.integers(0, 2)))
        elif kind == 1:
            rank_out = int(rng.integers(1, 3))
            got, want = _run_glwe_op(hip, ref, True, n, rank, rank_out, a_size, a_b, key_size, key_b, dnum, dsize, res_size, res_b, batch, seed,
                                     chunk=chunk, fuse=fuse)
        else:
            mode = ["automorphism", "add", "sub", "sub_negate"][int(rng.integers(0, 4))]
            gal = int(rng.choice([-5, 5, 25, 2 * n - 1, 3]))
            got, want = _run_glwe_op(hip, ref, True, n, rank, rank, a_size, a_b, key_size, key_b, dnum, dsize, res_size, res_b, batch, seed,
                                     chunk=chunk, fuse=fuse, auto=(gal, mode))
        assert np.array_equal(got, want), desc


def test_seeded_blind_rotation_sweep(mods):
    """25 random blind-rotation shapes (fixed seed): N 2^8..2^11, rank 1-3, block size 1-5, 1-3 limbs for key / accumulator / LUT,
    dnum 1-3, ragged batches: the one-kernel path where it applies (N <= 1024 and the shape fits LDS), the composed path elsewhere."""
    import os
    rng = np.random.default_rng(int(os.environ.get("POULPY_SWEEP_SEED", "77")))
    for case in range(int(os.environ.get("POULPY_SWEEP_CASES", "25"))):
        n = int(2 ** rng.integers(8, 12))
        rank = int(rng.integers(1, 4))
        blk = int(rng.integers(1, 6))
        n_lwe = int(rng.integers(1, 4)) * blk + int(rng.integers(0, blk))   # usually a trailing partial block
        dnum, bsz, rsz = (int(x) for x in rng.integers(1, 4, 3))
        k = int(rng.integers(10, 16))
        batch = int(rng.integers(1, 8))
        ref, hip = mods(n)
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(dict(case=case, n=n, rank=rank, blk=blk, n_lwe=n_lwe, dnum=dnum, bsz=bsz, rsz=rsz, k=k, batch=batch), flush=True)
        got, want = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=batch, seed=9000 + case)
        assert np.array_equal(got, want), dict(case=case, n=n, rank=rank, blk=blk, n_lwe=n_lwe, dnum=dnum, bsz=bsz, rsz=rsz, k=k, batch=batch)


def test_seeded_per_op_sweep(mods):
    """60 random shapes (fixed seed) through the per-op ABI: vec_znx_dft_apply with any (step, offset) incl. out of range,
    vmp_apply_dft_to_dft with more / fewer rows than input limbs and any limb_offset incl. >= size, idft into smaller / larger
    outputs, cross-base normalize with offsets; N 2^5..2^13 (every plan family)."""
    import os
    _per_op_sweep(mods, int(os.environ.get("POULPY_SWEEP_SEED", "4242")), int(os.environ.get("POULPY_SWEEP_CASES", "60")), 5,
                  int(os.environ.get("POULPY_SWEEP_MAXLOGN", "13")))


def test_seeded_per_op_sweep_tiny_rings(mods):
    """the same chains at N = 8 and 16 (24 shapes, fixed seed)."""
    import os
    _per_op_sweep(mods, int(os.environ.get("POULPY_SWEEP_SEED", "816")), int(os.environ.get("POULPY_SWEEP_CASES", "24")), 3, 4)


def _per_op_sweep(mods, seed, cases, min_logn, max_logn):
    import os
    rng = np.random.default_rng(seed)
    for case in range(cases):
        n = int(2 ** rng.integers(min_logn, max_logn + 1))
        ref, hip = mods(n)
        cols_in, cols_out = (int(x) for x in rng.integers(1, 4, 2))
        a_size, rows, size, res_size = (int(x) for x in rng.integers(1, 7, 4))
        step, offset = int(rng.integers(1, 4)), int(rng.integers(0, 8))
        limb_offset = int(rng.integers(0, size + 2))
        k = int(rng.integers(9, 16))
        desc = dict(case=case, n=n, cols_in=cols_in, cols_out=cols_out, a_size=a_size, rows=rows, size=size, res_size=res_size, step=step,
                    offset=offset, limb_offset=limb_offset, k=k)
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(desc, flush=True)
        a = VecZnx(n, cols_in, a_size).fill_uniform(k, rng)
        mat = MatZnx(n, rows, cols_in, cols_out, size).fill_uniform(k, rng)
        d_size = int(rng.integers(1, 7))
        outs = []
        for mod in (ref, hip):
            ad = mod.vec_znx_dft_alloc(cols_in, d_size)
            ad.data[...] = 0.0                                   # limbs the (step, offset) selection leaves untouched must agree
            for j in range(cols_in):
                mod.vec_znx_dft_apply(step, offset, ad, j, a, j)
            pm = mod.vmp_pmat_alloc(rows, cols_in, cols_out, size)
            mod.vmp_prepare(pm, mat)
            rd = mod.vec_znx_dft_alloc(cols_out, res_size)
            rd.data[...] = 0.0                                   # zero-tail semantics are only claimed on a zeroed result
            mod.vmp_apply_dft_to_dft(rd, ad, pm, limb_offset)
            big = VecZnxBig(n, cols_out, int(rng.integers(1, 7)) if mod is ref else outs[0][1])
            big_size = big.size
            big.data[...] = 13
            for c in range(cols_out):
                mod.vec_znx_idft_apply(big, c, rd, c)
            res_k = int(rng.integers(9, 20)) if mod is ref else outs[0][2]
            off = int(rng.integers(-k, k + 1)) if mod is ref else outs[0][3]
            res = VecZnx(n, cols_out, int(rng.integers(1, 7)) if mod is ref else outs[0][4])
            res.data[...] = -9
            for c in range(cols_out):
                mod.vec_znx_big_normalize(res, res_k, off, c, big, k, c)
            outs.append((res.data.copy(), big_size, res_k, off, res.size, big.data.copy()))
        assert np.array_equal(outs[0][5], outs[1][5]), ("big", desc)
        assert np.array_equal(outs[0][0], outs[1][0]), ("normalized", desc)


def test_seeded_svp_and_elementwise_sweep(mods):
    """50 random shapes (fixed seed) through svp_prepare / svp_apply_dft / _dft_to_dft / _assign, the limb-wise DFT-domain family
    (add_into, sub, add_assign, sub_assign, sub_negate_assign, add_scaled_assign, copy with (step, offset), zero), idft_tmpa,
    big_add_small_assign and the automorphism ops, with ragged sizes; compared after idft + normalize."""
    import os
    rng = np.random.default_rng(int(os.environ.get("POULPY_SWEEP_SEED", "909")))
    for case in range(int(os.environ.get("POULPY_SWEEP_CASES", "50"))):
        n = int(2 ** rng.integers(5, 13))
        ref, hip = mods(n)
        cols = int(rng.integers(1, 4))
        sa, sb, sr = (int(x) for x in rng.integers(1, 6, 3))
        k = int(rng.integers(9, 15))
        step, offset, scale = int(rng.integers(1, 4)), int(rng.integers(0, 6)), int(rng.integers(-3, 4))
        gal = int(rng.choice([-5, 5, 3, 2 * n - 1, 25]))
        op = int(rng.integers(0, 8))
        ca, cb, cr = (int(x) for x in rng.integers(0, cols, 3))
        desc = dict(case=case, n=n, cols=cols, sa=sa, sb=sb, sr=sr, k=k, step=step, offset=offset, scale=scale, gal=gal, op=op, ca=ca, cb=cb, cr=cr)
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(desc, flush=True)
        a = VecZnx(n, cols, sa).fill_uniform(k, rng)
        b = VecZnx(n, cols, sb).fill_uniform(k, rng)
        sc = ScalarZnx(n, cols).fill_uniform(k, rng)
        r0 = rng.standard_normal((sr, cols, n)) * 1e3
        outs = []
        for mod in (ref, hip):
            da, db, dr = mod.vec_znx_dft_alloc(cols, sa), mod.vec_znx_dft_alloc(cols, sb), mod.vec_znx_dft_alloc(cols, sr)
            for c in range(cols):
                mod.vec_znx_dft_apply(1, 0, da, c, a, c)
                mod.vec_znx_dft_apply(1, 0, db, c, b, c)
                mod.vec_znx_dft_apply(1, 0, dr, c, VecZnx(n, cols, sr, np.ascontiguousarray(np.rint(r0).astype(np.int64))), c)
            pp = SvpPPol(n, cols)
            for c in range(cols):
                mod.svp_prepare(pp, c, sc, c)
            if op == 0:
                mod.vec_znx_dft_add_into(dr, cr, da, ca, db, cb)
            elif op == 1:
                mod.vec_znx_dft_sub(dr, cr, da, ca, db, cb)
            elif op == 2:
                mod.vec_znx_dft_add_assign(dr, cr, da, ca)
                mod.vec_znx_dft_sub_assign(dr, cr, db, cb)
            elif op == 3:
                mod.vec_znx_dft_sub_negate_assign(dr, cr, da, ca)
                mod.vec_znx_dft_add_scaled_assign(dr, cr, db, cb, scale)
            elif op == 4:
                mod.vec_znx_dft_copy(step, offset, dr, cr, da, ca)
            elif op == 5:
                mod.svp_apply_dft(dr, cr, pp, ca, b, cb)
            elif op == 6:
                mod.svp_apply_dft_to_dft(dr, cr, pp, ca, db, cb)
                mod.vec_znx_dft_zero(dr, (cr + 1) % cols)
            else:   # one product only per case: two would leave the 53-bit range where different FFTs round alike
                mod.svp_apply_dft_to_dft_assign(dr, cr, pp, cb)
            big = VecZnxBig(n, cols, sr)
            big.data[...] = 3
            for c in range(cols):
                mod.vec_znx_idft_apply_tmpa(big, c, dr, c)
            mod.vec_znx_big_add_small_assign(big, cr, a, ca)
            mod.vec_znx_big_automorphism_assign(gal, big, cr)
            res = VecZnx(n, cols, sr + 1)
            res.data[...] = -1
            for c in range(cols):
                mod.vec_znx_big_normalize(res, k, 0, c, big, k, c)
            rot = VecZnx(n, cols, sb)
            mod.vec_znx_automorphism(gal, rot, cb, res, cr)
            outs.append((res.data.copy(), rot.data.copy()))
        assert np.array_equal(outs[0][0], outs[1][0]), desc
        assert np.array_equal(outs[0][1], outs[1][1]), desc


def test_concurrent_callers_on_one_module(mods):
    """poulpy-bin-fhe shares one &Module between scoped worker threads (bdd_arithmetic/eval.rs:210-221): concurrent calls on one
    module with disjoint buffers must be safe (serialised internally) and every caller must get its own result."""
    from concurrent.futures import ThreadPoolExecutor
    n, cols, size, k = 2048, 2, 3, 14
    ref, hip = mods(n)

    def work(seed):
        rng = seeded(seed)
        a = VecZnx(n, cols, size).fill_uniform(k, rng)
        s = ScalarZnx(n, cols).fill_uniform(k, rng)
        outs = []
        for mod in (ref, hip):
            pp = SvpPPol(n, cols)
            d = mod.vec_znx_dft_alloc(cols, size)
            for c in range(cols):
                mod.svp_prepare(pp, c, s, c)
            for c in range(cols):
                mod.svp_apply_dft(d, c, pp, c, a, c)
            big = mod.vec_znx_idft_apply_consume(d)
            outs.append(normalize_all(mod, big, k).data.copy())
        return np.array_equal(outs[0], outs[1])

    with ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(work, range(300, 340)))
    assert all(results)


def test_host_key_change_through_one_module_reaches_its_siblings(mods):
    """a sibling module that mirrors a host-resident prepared key must notice when ANOTHER module re-prepares that buffer,
    also when the change would slip through the sampled fingerprint: keys 1 and 2 differ in ONE coefficient of one matrix
    polynomial, prepared into the same host buffer through module A; module B (a clone) computed with key 1 before and must
    compute with key 2 after.  And freeing a pz_alloc_bytes block invalidates the mirrors of keys inside it without touching the modules."""
    import ctypes as C
    from poulpy_amd.hal import GlweOpParams, Module
    n, cols, size, k = 4096, 2, 3, 14
    ref, _ = mods(n)
    A = Module(n)
    B = A.clone()
    rng = seeded(77)
    mat1 = MatZnx(n, size, cols, cols, size).fill_uniform(k, rng)
    mat2 = MatZnx(n, size, cols, cols, size, mat1.data.copy())
    mat2.data.reshape(-1)[12345] ^= 1          # one coefficient, one bit
    a = VecZnx(n, cols, size).fill_uniform(k, rng)
    p = GlweOpParams(rank=1, dnum=size, dsize=1, key_size=size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k, rank_out=1)
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)
    wants = []
    for mat in (mat1, mat2):
        pr = ref.vmp_pmat_alloc(size, cols, cols, size)
        ref.vmp_prepare(pr, mat)
        w = VecZnx(n, cols, size)
        ref.glwe_external_product(w, k, a, k, pr, 1, k)
        wants.append(w.data.copy())
    assert not np.array_equal(wants[0], wants[1])
    ph = A.vmp_pmat_alloc(size, cols, cols, size)           # ONE host buffer for both keys
    A.vmp_prepare(ph, mat1)
    got = np.zeros_like(wants[0])
    B.glwe_external_product_batched(hp(got), hp(a.data), hp(ph.data), p, 1)     # B mirrors key 1
    assert np.array_equal(got, wants[0])
    A.vmp_prepare(ph, mat2)                                                      # re-prepared through A, same address
    got[...] = 0
    B.glwe_external_product_batched(hp(got), hp(a.data), hp(ph.data), p, 1)
    assert np.array_equal(got, wants[1]), "the sibling kept computing with the stale mirror"
    B.close()
    A.close()


def test_freed_host_key_gives_its_device_mirrors_back_at_once(mods):
    """pz_free_bytes used to only publish the freed range - the mirror and its row-sliced copy (2 x the key bytes)
    stayed on the device until the owning module next looked a key up, which a parked sibling never does.  Now idle modules are swept when
    the range is published: device memory comes back without another call on either module.  Then the module-local sweep on entry: a
    re-prepared host key (published through module A) is dropped by B at B's next call of ANY kind."""
    import ctypes as C
    import torch
    from poulpy_amd.hal import GlweOpParams, Module
    n, cols, size, k = 16384, 2, 8, 12        # 32 MiB key: mirror + sliced copy = 64 MiB per module
    ref, _ = mods(n)
    A = Module(n)
    B = A.clone()
    lib = A.lib
    rng = seeded(78)
    mat = MatZnx(n, size, cols, cols, size).fill_uniform(k, rng)
    key_bytes = n * size * cols * cols * size * 8
    blk = lib.pz_alloc_bytes(key_bytes)
    assert blk
    host_key = np.ctypeslib.as_array((C.c_double * (key_bytes // 8)).from_address(blk))
    ph = A.vmp_pmat_alloc(size, cols, cols, size)
    A.vmp_prepare(ph, mat)
    host_key[:] = ph.data.reshape(-1)
    a = VecZnx(n, cols, size).fill_uniform(k, rng)
    p = GlweOpParams(rank=1, dnum=size, dsize=1, key_size=size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k, rank_out=1)
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)
    got = np.zeros((size, cols, n), dtype=np.int64)
    for M in (A, B):     # both modules mirror the key (and keep its row-sliced copy)
        M.glwe_external_product_batched(hp(got), hp(a.data), C.c_void_p(blk), p, 1)
    pr = ref.vmp_pmat_alloc(size, cols, cols, size)
    ref.vmp_prepare(pr, mat)
    want = VecZnx(n, cols, size)
    ref.glwe_external_product(want, k, a, k, pr, 1, k)
    assert np.array_equal(got, want.data)
    A.sync(); B.sync()
    free0, _ = torch.cuda.mem_get_info()
    lib.pz_free_bytes(C.c_void_p(blk))            # no call on A or B follows
    free1, _ = torch.cuda.mem_get_info()
    assert free1 - free0 >= 3 * key_bytes, (free0, free1, key_bytes)    # 4 x key bytes expected; allocator granularity may hold some back
    # entry sweep: B mirrors a key that is then zeroed through A; B's next call - one that never looks a key up - releases the mirror
    ph2 = A.vmp_pmat_alloc(size, cols, cols, size)
    A.vmp_prepare(ph2, mat)
    B.glwe_external_product_batched(hp(got), hp(a.data), hp(ph2.data), p, 1)
    B.sync()
    free2, _ = torch.cuda.mem_get_info()
    lib.pz_vmp_zero.argtypes = [C.c_void_p, C.c_void_p] + [C.c_size_t] * 4
    assert lib.pz_vmp_zero(A.handle, hp(ph2.data), size, cols, cols, size) == 0     # publishes; B is idle: swept right here
    free3, _ = torch.cuda.mem_get_info()
    assert free3 - free2 >= key_bytes, (free2, free3)
    B.close()
    A.close()


def test_sibling_modules_run_concurrently(mods):
    """pz_module_clone: every worker thread on its own sibling (shared device tables; own stream, workspaces, lock) — per-op host calls
    and fused GLWE calls on host containers from 6 threads at once, each result against the oracle; siblings outlive their parent."""
    import ctypes as C
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from poulpy_amd.hal import GlweOpParams, Module
    n, cols, size, k = 4096, 2, 3, 14
    ref, _ = mods(n)
    parent = Module(n)
    tls = threading.local()
    sibs, sibs_lock = [], threading.Lock()

    def my_module():
        if not hasattr(tls, "mod"):
            tls.mod = parent.clone()
            with sibs_lock:
                sibs.append(tls.mod)
        return tls.mod

    mat = MatZnx(n, size, cols, cols, size).fill_uniform(k, seeded(1))
    pr = ref.vmp_pmat_alloc(size, cols, cols, size)
    ref.vmp_prepare(pr, mat)
    p = GlweOpParams(rank=1, dnum=size, dsize=1, key_size=size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k, rank_out=1)
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)

    def work(seed):
        hip = my_module()
        rng = seeded(seed)
        ok = True
        # per-op sequence on host containers
        a = VecZnx(n, cols, size).fill_uniform(k, rng)
        sc = ScalarZnx(n, cols).fill_uniform(k, rng)
        outs = []
        for mod in (ref, hip):
            pp = SvpPPol(n, cols)
            d = mod.vec_znx_dft_alloc(cols, size)
            for c in range(cols):
                mod.svp_prepare(pp, c, sc, c)
            for c in range(cols):
                mod.svp_apply_dft(d, c, pp, c, a, c)
            outs.append(normalize_all(mod, mod.vec_znx_idft_apply_consume(d), k).data.copy())
        ok = ok and np.array_equal(outs[0], outs[1])
        # fused external product on host containers with this sibling's own key mirror
        ph = hip.vmp_pmat_alloc(size, cols, cols, size)
        hip.vmp_prepare(ph, mat)
        want = VecZnx(n, cols, size)
        ref.glwe_external_product(want, k, a, k, pr, 1, k)
        got = np.zeros_like(want.data)
        hip.glwe_external_product_batched(hp(got), hp(a.data), hp(ph.data), p, 1)
        return ok and np.array_equal(got, want.data)

    with ThreadPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(work, range(700, 736)))
    assert all(results)
    assert 1 <= len(sibs) <= 6
    parent.close()                       # the tables are reference-counted: siblings keep working after their parent is gone
    a = VecZnx(n, cols, size).fill_uniform(k, seeded(9))
    d = sibs[0].vec_znx_dft_alloc(cols, size)
    sibs[0].vec_znx_dft_apply(1, 0, d, 0, a, 0)
    big = sibs[0].vec_znx_big_alloc(cols, size)
    sibs[0].vec_znx_idft_apply(big, 0, d, 0)
    assert np.array_equal(big.data[:, 0], a.data[:, 0])
    for sib in sibs:
        sib.close()


@pytest.mark.parametrize("n", [512, 4096, 65536])
def test_batched_primitives_compose_an_external_product(mods, n):
    """The four batched primitives of the ABI (pz_vec_znx_dft_apply_batched, pz_vmp_apply_dft_to_dft_batched,
    pz_vec_znx_idft_apply_consume_batched, pz_vec_znx_big_normalize_batched) on device-resident ciphertexts, composed exactly like
    external_product/glwe.rs:197-271 (incl. a (step, offset) limb selection and a limb_offset), against the oracle per ciphertext."""
    ref, hip = mods(n)
    rng = seeded(n + 77)
    cols, a_size, size, dnum, k, batch = 2, 5, 4, 3, 13, 6
    for (step, offset, limb_offset) in ((1, 0, 0), (2, 1, 1)):
        sel = len(range(offset, a_size, step))
        d_size = min(sel, dnum)
        mat = MatZnx(n, dnum, cols, cols, size).fill_uniform(k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, size), hip.vmp_pmat_alloc(dnum, cols, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        a_all = np.empty((batch, a_size, cols, n), dtype=np.int64)
        want = np.empty((batch, size, cols, n), dtype=np.int64)
        for b in range(batch):
            a = VecZnx(n, cols, a_size).fill_uniform(k, rng)
            a_all[b] = a.data
            ad = ref.vec_znx_dft_alloc(cols, d_size)
            for c in range(cols):
                ref.vec_znx_dft_apply(step, offset, ad, c, a, c)
            rd = ref.vec_znx_dft_alloc(cols, size)
            ref.vmp_apply_dft_to_dft(rd, ad, pr, limb_offset)
            big = ref.vec_znx_idft_apply_consume(rd)
            res = VecZnx(n, cols, size)
            for c in range(cols):
                ref.vec_znx_big_normalize(res, k, 0, c, big, k, c)
            want[b] = res.data
        d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
        d_key = hip.device_alloc(ph.data.nbytes).upload(ph.data)
        d_ad = hip.device_alloc(batch * d_size * cols * n * 8)
        d_rd = hip.device_alloc(batch * size * cols * n * 8)
        d_res = hip.device_alloc(want.nbytes)
        hip.lib.pz_memset_d(hip.handle, d_rd.ptr, 0, batch * size * cols * n * 8)
        for c in range(cols):
            hip.vec_znx_dft_apply_batched(batch, step, offset, d_ad.ptr, cols, d_size, c, d_a.ptr, cols, a_size, c)
        hip.vmp_apply_dft_to_dft_batched(batch, d_rd.ptr, cols, size, d_ad.ptr, cols, d_size, d_key.ptr, dnum, cols, cols, size, limb_offset)
        hip.vec_znx_idft_apply_consume_batched(batch, d_rd.ptr, cols, size)
        for c in range(cols):
            hip.vec_znx_big_normalize_batched(batch, d_res.ptr, cols, size, k, 0, c, d_rd.ptr, cols, size, k, c)
        hip.sync()
        got = d_res.download(np.int64, want.size).reshape(want.shape)
        for buf in (d_a, d_key, d_ad, d_rd, d_res):
            buf.free()
        assert np.array_equal(got, want), (n, step, offset, limb_offset)


def test_tmp_bytes_workspace_and_small_utilities(mods):
    """*_tmp_bytes must return the reference's numbers (poulpy-core sizes its scratch arenas from them); vmp_zero,
    workspace queries and the event helpers behave."""
    from poulpy_amd.hal import BlindRotationParams, GlweOpParams
    n = 1024
    ref, hip = mods(n)
    assert hip.vec_znx_big_normalize_tmp_bytes() == ref.vec_znx_big_normalize_tmp_bytes() == 3 * n * 8        # normalize.rs:13-15
    assert hip.vec_znx_idft_apply_tmp_bytes() == 0                                                          # hal_defaults/vec_znx_dft.rs:68-73
    for (rs, asz, rows, ci, co, sz) in ((4, 3, 5, 2, 2, 4), (1, 1, 1, 1, 1, 1), (6, 7, 3, 1, 2, 5)):
        assert hip.vmp_apply_dft_to_dft_tmp_bytes(rs, asz, rows, ci, co, sz) == ref.vmp_apply_dft_to_dft_tmp_bytes(rs, asz, rows, ci, co, sz)
        assert hip.vmp_prepare_tmp_bytes(rows, ci, co, sz) == ref.vmp_prepare_tmp_bytes(rows, ci, co, sz) == n * 8   # vmp.rs:13-15
        # family_common.rs:3-15: the DFT of a (cols_in x min(a.size, rows)) plus the dft_to_dft scratch
        assert hip.vmp_apply_dft_tmp_bytes(rs, asz, rows, ci, co, sz) == n * 8 * ci * min(asz, rows) + ref.vmp_apply_dft_to_dft_tmp_bytes(
            rs, min(asz, rows), rows, ci, co, sz)
    pm = hip.vmp_pmat_alloc(2, 1, 2, 2)
    pm.data[...] = 3.5
    hip.vmp_zero(pm)
    assert not pm.data.any()
    p = GlweOpParams(rank=1, dnum=2, dsize=1, key_size=2, key_base2k=12, a_size=2, a_base2k=12, res_size=2, res_base2k=12, rank_out=1)
    w1, w8 = hip.glwe_op_workspace_bytes(p, 1, 0), hip.glwe_op_workspace_bytes(p, 8, 0)
    assert 0 < w1 <= w8 and hip.glwe_op_workspace_bytes(p, 8, 2) >= hip.glwe_op_workspace_bytes(p, 8, 1) > 0
    bp = BlindRotationParams(rank=1, n_lwe=10, block_size=5, dnum=2, brk_size=2, base2k=12, res_size=2, lut_size=2)
    assert hip.blind_rotation_workspace_bytes(bp, 4) > 0
    e0, e1 = hip.event_create(), hip.event_create()
    hip.event_record(e0)
    hip.event_record(e1)
    assert hip.event_elapsed_ms(e0, e1) >= 0.0


def test_batched_entry_points_reject_bad_arguments(mods):
    """Shape / pointer violations are reported (PZ_ERR_INVALID -> PoulpyHipError), never undefined behaviour: host pointers where
    device pointers are required (the GLWE-level calls accept host containers: test_glwe_ops_on_host_containers), empty shapes, an even Galois element, an automorphism key that changes the rank, an unknown mode,
    base2k out of range; a zero batch is a no-op."""
    import ctypes as C
    from poulpy_amd.hal import BlindRotationParams, GlweOpParams, PoulpyHipError
    n = 256
    _, hip = mods(n)
    good = GlweOpParams(rank=1, dnum=2, dsize=1, key_size=2, key_base2k=12, a_size=2, a_base2k=12, res_size=2, res_base2k=12, rank_out=1)
    d = hip.device_alloc(n * 8 * 64)
    host = np.zeros(n * 64, dtype=np.int64)
    hp = host.ctypes.data_as(C.c_void_p)
    hip.glwe_external_product_batched(d.ptr, d.ptr, d.ptr, good, 0)                      # batch 0: nothing to do
    with pytest.raises(PoulpyHipError):
        hip.glwe_external_product_batched(None, d.ptr, d.ptr, good, 1)                   # null pointer
    with pytest.raises(PoulpyHipError):
        hip.glwe_keyswitch_batched(d.ptr, d.ptr, None, good, 1)
    with pytest.raises(PoulpyHipError):
        hip.glwe_pack_batched(hp, [0], [d.ptr], 0, [5] * 8, [d.ptr] * 8, good, d.ptr, 1 << 20, 1)   # host pointer where a device one is required
    bad = GlweOpParams(rank=1, dnum=0, dsize=1, key_size=2, key_base2k=12, a_size=2, a_base2k=12, res_size=2, res_base2k=12, rank_out=1)
    with pytest.raises(PoulpyHipError):
        hip.glwe_external_product_batched(d.ptr, d.ptr, d.ptr, bad, 1)                   # empty shape
    with pytest.raises(PoulpyHipError):
        hip.glwe_automorphism_batched(d.ptr, d.ptr, d.ptr, good, 4, "automorphism", 1)   # even Galois element
    with pytest.raises(PoulpyHipError):
        hip.glwe_automorphism_batched(d.ptr, d.ptr, d.ptr, good, 5, 9, 1)                # unknown mode
    rk = GlweOpParams(rank=1, dnum=2, dsize=1, key_size=2, key_base2k=12, a_size=2, a_base2k=12, res_size=2, res_base2k=12, rank_out=2)
    with pytest.raises(PoulpyHipError):
        hip.glwe_automorphism_batched(d.ptr, d.ptr, d.ptr, rk, 5, "add", 1)              # key changes the rank
    br = BlindRotationParams(rank=1, n_lwe=4, block_size=2, dnum=1, brk_size=1, base2k=70, res_size=1, lut_size=1)
    with pytest.raises(PoulpyHipError):
        hip.blind_rotation_execute_batched(d.ptr, d.ptr, d.ptr, d.ptr, br, 1)            # base2k out of range
    br0 = BlindRotationParams(rank=1, n_lwe=0, block_size=2, dnum=1, brk_size=1, base2k=12, res_size=1, lut_size=1)
    with pytest.raises(PoulpyHipError):
        hip.blind_rotation_execute_batched(d.ptr, d.ptr, d.ptr, d.ptr, br0, 1)           # empty LWE
    with pytest.raises(PoulpyHipError):
        hip.vec_znx_big_normalize_batched(1, hp, 1, 1, 12, 0, 0, d.ptr, 1, 1, 12, 0)     # host pointer to a batched primitive
    d.free()


@pytest.mark.parametrize("n", [64, 2048, 65536])
def test_vec_znx_rsh_assign(mods, n):
    """hal_impl.rs:217: every shift 0..base2k*size incl. several limbs falling off (where the reference's last loop aliases limbs:
    replayed literally), ragged sizes, the other column untouched."""
    ref, hip = mods(n)
    rng = seeded(n + 9)
    for base2k in (7, 13):
        for size in (1, 2, 4):
            for k in sorted(set([0, 1, 2, base2k - 1, base2k, base2k + 1, 2 * base2k, base2k * size])):
                if k > base2k * size:
                    continue
                a = VecZnx(n, 2, size).fill_uniform(40, rng)
                r, h = a.copy(), a.copy()
                ref.vec_znx_rsh_assign(base2k, k, r, 1)
                hip.vec_znx_rsh_assign(base2k, k, h, 1)
                assert np.array_equal(r.data, h.data), (base2k, size, k)


@pytest.mark.parametrize("fuse", [(True, True), (False, False)], ids=["fused", "unfused"])
def test_glwe_trace_batched(mods, fuse):
    """poulpy-core/src/glwe_trace.rs:129-176 on device-resident ciphertexts against the oracle: the Galois elements of a full trace
    (-1, 5, 5^2, ...) with one prepared key per step, N = 512 (five-kernel path) and N = 8192 (fused pipeline)."""
    from poulpy_amd.hal import GlweOpParams
    for (n, rank, size, dnum, key_size, k, batch, nsteps) in ((512, 1, 3, 3, 4, 13, 5, 4), (8192, 1, 4, 4, 4, 12, 3, 3), (8192, 2, 3, 2, 3, 14, 2, 2)):
        ref, hip = mods(n)
        rng = seeded(n + rank)
        cols = rank + 1
        gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(nsteps - 1)]
        prs, d_keys = [], []
        for _ in gals:
            mat = MatZnx(n, dnum, rank, cols, key_size).fill_uniform(k, rng)
            pr, ph = ref.vmp_pmat_alloc(dnum, rank, cols, key_size), hip.vmp_pmat_alloc(dnum, rank, cols, key_size)
            ref.vmp_prepare(pr, mat)
            hip.vmp_prepare(ph, mat)
            prs.append(pr)
            d_keys.append(hip.device_alloc(ph.data.nbytes).upload(ph.data))
        cts = np.empty((batch, size, cols, n), dtype=np.int64)
        want = np.empty_like(cts)
        for b in range(batch):
            ct = VecZnx(n, cols, size).fill_uniform(k, rng)
            cts[b] = ct.data
            ref.glwe_trace_assign(ct, k, gals, prs)
            want[b] = ct.data
        d_res = hip.device_alloc(cts.nbytes).upload(cts)
        p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=key_size, key_base2k=k, a_size=size, a_base2k=k, res_size=size,
                         res_base2k=k, rank_out=rank)
        hip.set_fusion(*fuse)
        hip.glwe_trace_batched(d_res.ptr, gals, [d.ptr for d in d_keys], p, batch)
        hip.sync()
        hip.set_fusion(True, True)
        got = d_res.download(np.int64, want.size).reshape(want.shape)
        for d in d_keys + [d_res]:
            d.free()
        assert np.array_equal(got, want), (n, rank, size)


@pytest.mark.parametrize("size,key_size,res_gals", [(8, 8, [-1, 5, 25, 625]), (3, 4, [5, 25]), (4, 3, [5, 25, 625])])
@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384, 32768, 65536])
def test_glwe_trace_shifted_stores(mods, n, size, key_size, res_gals):
    """glwe_trace where the one-bit vec_znx_rsh in front of a step rides on the previous step's last kernel: N = 2^16 (256 x 128 plan) on
    the tail of the spectral automorphism form (Galois element = 1 mod 4; k_inv_tail<.., RSH>), N <= 4096 on k_small_inv<.., AU>; against
    the oracle's literal sequence (rsh, glwe_automorphism_add_assign per step).  -1 first (the full trace's order); key limbs above /
    below the ciphertext's (zero-extended limb, carry-only first step).  Round 6: on the pipeline plans the body column of every step takes
    the 16-bit operand with the shifted store behind the f64 chain (k_inv_tail<.., RSH, 7, SGN>; base2k <= 14)."""
    _trace_shifted_stores(mods, n, size, key_size, res_gals, 12)


@pytest.mark.parametrize("k", [14, 15])
def test_glwe_trace_shifted_stores_wider_bases(mods, k):
    """base2k 14: the widest base at which the trace's steps take the 16-bit body operand (phi(body) + a0 of two normalized digits always fits);
    base2k 15: the i64 pre-pass and the operand variant with the shifted store, as before."""
    _trace_shifted_stores(mods, 8192, 3, 4, [-1, 5, 25, 625], k)
    _trace_shifted_stores(mods, 16384, 4, 3, [5, 25], k)


def _trace_shifted_stores(mods, n, size, key_size, res_gals, k):
    from poulpy_amd.hal import GlweOpParams
    rank, dnum, batch = 1, 2, 2
    if key_size > 4 and n <= 4096:
        key_size, size = 4, 4     # the small-ring pipeline takes up to four key limbs
    ref, hip = mods(n)
    rng = seeded(n + size)
    cols = rank + 1
    prs, d_keys = [], []
    for _ in res_gals:
        mat = MatZnx(n, dnum, rank, cols, key_size).fill_uniform(k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, rank, cols, key_size), hip.vmp_pmat_alloc(dnum, rank, cols, key_size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        prs.append(pr)
        d_keys.append(hip.device_alloc(ph.data.nbytes).upload(ph.data))
    gals = [g % (2 * n) if g > 0 else g for g in res_gals]
    cts = np.empty((batch, size, cols, n), dtype=np.int64)
    want = np.empty_like(cts)
    for b in range(batch):
        ct = VecZnx(n, cols, size).fill_uniform(k, rng)
        cts[b] = ct.data
        ref.glwe_trace_assign(ct, k, gals, prs)
        want[b] = ct.data
    d_res = hip.device_alloc(cts.nbytes).upload(cts)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=key_size, key_base2k=k, a_size=size, a_base2k=k, res_size=size, res_base2k=k,
                     rank_out=rank)
    hip.glwe_trace_batched(d_res.ptr, gals, [d.ptr for d in d_keys], p, batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for d in d_keys + [d_res]:
        d.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,res_k,key_k,kbits,batch,nsteps", [
    (256, 1, 14, 13, 4 * 14 + 1, 3, 8),     # poulpy-core test_suite/trace.rs:36-39: result base2k, keys base2k - 1, k = 4 base2k + 1, full trace
    (256, 2, 12, 15, 40, 2, 5),             # keys in the larger base, partial trace
    (4096, 1, 13, 12, 4 * 13 + 1, 5, 3),    # fused automorphism pipeline under the conversions
])
def test_glwe_trace_batched_result_in_another_base_than_the_keys(mods, n, rank, res_k, key_k, kbits, batch, nsteps):
    """glwe_trace_assign with res.base2k != key.base2k (glwe_trace.rs:153-163): normalize into the keys' base, trace, normalize back."""
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(n + rank + res_k)
    cols = rank + 1
    res_size, conv_size = -(-kbits // res_k), -(-kbits // key_k)
    key_size = -(-(kbits + key_k) // key_k)
    dnum = conv_size
    log_n = n.bit_length() - 1
    gals = ([-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)])[log_n - nsteps:]
    prs, d_keys = [], []
    for _ in gals:
        mat = MatZnx(n, dnum, rank, cols, key_size).fill_uniform(key_k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, rank, cols, key_size), hip.vmp_pmat_alloc(dnum, rank, cols, key_size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        prs.append(pr)
        d_keys.append(hip.device_alloc(ph.data.nbytes).upload(ph.data))
    cts = np.empty((batch, res_size, cols, n), dtype=np.int64)
    want = np.empty_like(cts)
    for b in range(batch):
        ct = VecZnx(n, cols, res_size).fill_uniform(res_k, rng)
        cts[b] = ct.data
        ref.glwe_trace_assign_bases(ct, res_k, conv_size, key_k, gals, prs)
        want[b] = ct.data
    d_res = hip.device_alloc(cts.nbytes).upload(cts)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=key_size, key_base2k=key_k, a_size=conv_size, a_base2k=key_k, res_size=res_size,
                     res_base2k=res_k, rank_out=rank)
    for _ in range(3):    # second and third call: captured / replayed as a HIP graph, same contents restored each time
        d_res.upload(cts)
        hip.glwe_trace_batched(d_res.ptr, gals, [d.ptr for d in d_keys], p, batch)
        hip.sync()
        got = d_res.download(np.int64, want.size).reshape(want.shape)
        assert np.array_equal(got, want)
    for d in d_keys + [d_res]:
        d.free()


@pytest.mark.parametrize("n,rank,size,ct_k,key_k,log_gap_out,indices,batch", [
    (64, 3, 3, 12, 13, 0, [0, 5, 17, 32, 33, 63], 2),      # poulpy-core test_suite/glwe_packing.rs:40-45: ciphertexts base2k - 1, keys base2k, rank 3
    (256, 1, 3, 15, 11, 3, [0, 8, 16, 128, 136], 3),      # keys in the smaller base
    (4096, 1, 3, 12, 13, 9, [0, 512, 1024, 2048, 3584], 2),
])
def test_glwe_pack_batched_keys_in_another_base(mods, n, rank, size, ct_k, key_k, log_gap_out, indices, batch):
    """glwe_pack with ciphertexts / result and automorphism keys in different bases: pack_internal's arithmetic in the ciphertexts' base,
    converting automorphisms, closing trace on a temporary in the keys' base."""
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(7100 + n + rank + log_gap_out)
    cols = rank + 1
    log_n = n.bit_length() - 1
    kbits = size * ct_k - 3
    trace_size = -(-kbits // key_k)
    key_size = -(-(kbits + key_k) // key_k)
    dnum = trace_size
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    keys_r, keys_d = [], []
    for _ in gals:
        mat = MatZnx(n, dnum, rank, cols, key_size).fill_uniform(key_k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, rank, cols, key_size), hip.vmp_pmat_alloc(dnum, rank, cols, key_size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        keys_r.append(pr)
        keys_d.append(hip.device_alloc(ph.data.nbytes).upload(ph.data))
    data = {j: rng.integers(-(1 << (ct_k - 1)), 1 << (ct_k - 1), (batch, size, cols, n), dtype=np.int64) for j in indices}
    want = np.empty((batch, size, cols, n), dtype=np.int64)
    for b in range(batch):
        cts = {j: VecZnx(n, cols, size, data[j][b].copy()) for j in indices}
        res = VecZnx(n, cols, size)
        ref.glwe_pack_bases(res, ct_k, key_k, trace_size, cts, log_gap_out, gals, keys_r)
        want[b] = res.data
    d_cts = [hip.device_alloc(data[j].nbytes).upload(data[j]) for j in indices]
    d_res = hip.device_alloc(want.nbytes)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=key_size, key_base2k=key_k, a_size=size, a_base2k=ct_k, res_size=size,
                     res_base2k=ct_k, rank_out=rank)
    nbytes = hip.glwe_pack_bases_tmp_bytes(p, trace_size, batch)
    d_tmp = hip.device_alloc(nbytes)
    from poulpy_amd.hal import PoulpyHipError
    with pytest.raises(PoulpyHipError):     # the one-base entry point refuses keys in another base
        hip.glwe_pack_batched(d_res.ptr, indices, [d.ptr for d in d_cts], log_gap_out, gals, [k.ptr for k in keys_d], p, d_tmp.ptr, nbytes, batch)
    hip.glwe_pack_bases_batched(d_res.ptr, indices, [d.ptr for d in d_cts], log_gap_out, gals, [k.ptr for k in keys_d], p, trace_size,
                                d_tmp.ptr, nbytes, batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in keys_d + d_cts + [d_res, d_tmp]:
        buf.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [32, 1024, 65536])
def test_vec_znx_rotate(mods, n):
    """hal_impl.rs:225-232: res = X^k * a for any k (negative, > 2n), ragged sizes, the assign form, other columns untouched."""
    ref, hip = mods(n)
    rng = seeded(n + 5)
    for k in (0, 1, -1, 7, n - 1, n, n + 1, 2 * n - 1, 2 * n, -n - 3, 5 * n + 2):
        for (a_size, res_size) in ((3, 3), (2, 4), (4, 1)):
            a = VecZnx(n, 2, a_size).fill_uniform(50, rng)
            rr = VecZnx(n, 3, res_size).fill_uniform(60, rng)
            rh = rr.copy()
            ref.vec_znx_rotate(k, rr, 2, a, 1)
            hip.vec_znx_rotate(k, rh, 2, a, 1)
            assert np.array_equal(rr.data, rh.data), (k, a_size, res_size)
            want = a.copy()
            tmp = VecZnx(n, 2, a_size)
            ref.vec_znx_rotate(k, tmp, 0, a, 1)
            want.data[:, 1] = tmp.data[:, 0]
            got = a.copy()
            hip.vec_znx_rotate_assign(k, got, 1)
            assert np.array_equal(got.data, want.data), (k, a_size)


# ------------------------------------------------------------------------------------------
# circuit bootstrapping (constant mode): blind rotation -> dnum traces of rotated copies -> ggsw_expand_row
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,rank,n_lwe,block_size,brk_dnum,glwe_size,res_dnum,res_size,batch,skip", [
    (256, 1, 6, 3, 2, 3, 2, 2, 2, 0),     # GGSW narrower than the GLWE of the rotation (the reference bench's proportions)
    (256, 2, 4, 2, 3, 4, 2, 2, 3, 0),     # rank 2 as in poulpy-bench's circuit_bootstrapping shape
    (512, 1, 4, 1, 2, 2, 3, 3, 2, 0),     # standard (block_size 1) rotation; GGSW wider than the GLWE: zero-extended trace input
    (1024, 2, 7, 7, 3, 4, 2, 2, 2, 0),    # N = 1024, rank 2, block 7, size 4, dnum 3 (bench shape, short LWE)
    (256, 1, 6, 3, 2, 3, 2, 2, 2, 3),     # partial trace (steps 3..log_n): the exponent mode's post_process with equal gaps (circuit.rs:418-420)
])
def test_circuit_bootstrapping_to_constant(mods, n, rank, n_lwe, block_size, brk_dnum, glwe_size, res_dnum, res_size, batch, skip):
    """circuit_bootstrapping/circuit.rs:219-370 (constant mode, one base2k) through the C ABI vs the oracle's composition of the
    pinned pieces (blind rotation, glwe_trace, rotate, ggsw_expand_row); random lookup table and gap."""
    from poulpy_amd.hal import BlindRotationParams, CircuitBootstrappingParams
    base2k, atk_dnum, tsk_dnum = 13, 3, 2
    ref, hip = mods(n)
    rng = seeded(1300 + n + rank)
    cols = rank + 1
    log_n = n.bit_length() - 1
    gap = 2 * int(rng.integers(1, n // 8))

    def prepared(rows, cols_in, size):
        mat = MatZnx(n, rows, cols_in, cols, size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(rows, cols_in, cols, size), hip.vmp_pmat_alloc(rows, cols_in, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        return pr, ph

    lut = VecZnx(n, 1, glwe_size).fill_uniform(base2k, rng)
    brk_r = np.empty((n_lwe, n * brk_dnum * cols * cols * glwe_size), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        pr, ph = prepared(brk_dnum, cols, glwe_size)
        brk_r[i], brk_h[i] = pr.data.reshape(-1), ph.data.reshape(-1)
    tmp_size = max(glwe_size, res_size)
    gals = ([-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)])[skip:]   # glwe_trace.rs:166-168, steps skip..log_n
    atk = [prepared(atk_dnum, rank, tmp_size) for _ in gals]
    tsk = [prepared(tsk_dnum, rank, res_size + 1) for _ in range(rank)]
    lwe = rng.integers(-n, n, (batch, n_lwe + 1), dtype=np.int64)
    xpa = ref.blind_rotation_x_pow_a() if block_size > 1 else np.zeros((1, 1))
    want = np.empty((batch, res_dnum, cols, res_size, cols, n), dtype=np.int64)
    for b in range(batch):
        g = MatZnx(n, res_dnum, cols, cols, res_size)
        ref.circuit_bootstrap_to_constant(g, base2k, np.ascontiguousarray(lwe[b]), lut, brk_r, brk_dnum, glwe_size, glwe_size, block_size,
                                          xpa, gals, [a[0] for a in atk], [t[0] for t in tsk], gap)
        want[b] = g.data

    bufs = []

    def up(arr):
        d = hip.device_alloc(arr.nbytes).upload(arr)
        bufs.append(d)
        return d

    d_lwe, d_lut, d_brk = up(lwe), up(lut.data), up(brk_h)
    d_atk = [up(a[1].data) for a in atk]
    d_tsk = [up(t[1].data) for t in tsk]
    d_res = up(rng.integers(-5, 5, want.shape, dtype=np.int64))   # stale contents must be overwritten
    p = CircuitBootstrappingParams(
        br=BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block_size, dnum=brk_dnum, brk_size=glwe_size, base2k=base2k,
                               res_size=glwe_size, lut_size=glwe_size),
        atk_dnum=atk_dnum, atk_size=tmp_size, tsk_dnum=tsk_dnum, tsk_size=res_size + 1, res_dnum=res_dnum, res_size=res_size, gap=gap)
    nbytes = hip.circuit_bootstrapping_tmp_bytes(p, batch)
    assert nbytes > 0
    d_tmp = hip.device_alloc(nbytes)
    bufs.append(d_tmp)
    with pytest.raises(Exception):   # undersized scratch is refused, as the reference's scratch.available() assert
        hip.circuit_bootstrapping_execute_to_constant_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, gals, [k.ptr for k in d_atk],
                                                              [k.ptr for k in d_tsk], p, d_tmp.ptr, nbytes - 1, batch)
    hip.circuit_bootstrapping_execute_to_constant_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, gals, [k.ptr for k in d_atk],
                                                          [k.ptr for k in d_tsk], p, d_tmp.ptr, nbytes, batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in bufs:
        buf.free()
    assert np.array_equal(got[:, :, 0], want[:, :, 0]), "trace rows differ"
    assert np.array_equal(got, want)


def test_composite_calls_replay_as_hip_graphs(mods):
    """The launch-bound composite calls are captured into a HIP graph the second time they come with the same arguments and
    replayed afterwards: same addresses and shapes, NEW contents every call, results must follow the contents (composed
    blind-rotation path: hundreds of launches per call)."""
    from poulpy_amd.hal import BlindRotationParams
    n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch = 2048, 1, 6, 3, 2, 2, 2, 13, 3
    ref, hip = mods(n)
    rng = seeded(4100)
    cols = rank + 1
    lut = VecZnx(n, 1, rsz).fill_uniform(k, rng)
    brk_r = np.empty((n_lwe, n * dnum * cols * cols * bsz), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        mat = MatZnx(n, dnum, cols, cols, bsz).fill_uniform(k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, bsz), hip.vmp_pmat_alloc(dnum, cols, cols, bsz)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        brk_r[i], brk_h[i] = pr.data.reshape(-1), ph.data.reshape(-1)
    xpa = ref.blind_rotation_x_pow_a()
    d_lwe = hip.device_alloc(batch * (n_lwe + 1) * 8)
    d_lut = hip.device_alloc(lut.data.nbytes).upload(lut.data)
    d_brk = hip.device_alloc(brk_h.nbytes).upload(brk_h)
    d_res = hip.device_alloc(batch * rsz * cols * n * 8)
    p = BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=blk, dnum=dnum, brk_size=bsz, base2k=k, res_size=rsz, lut_size=rsz)
    hip.set_graphs(True)
    before = hip.graph_launches()
    for it in range(5):
        lwe = rng.integers(-n, n, (batch, n_lwe + 1), dtype=np.int64)
        d_lwe.upload(lwe)
        hip.blind_rotation_execute_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, p, batch)
        hip.sync()
        got = d_res.download(np.int64, batch * rsz * cols * n).reshape(batch, rsz, cols, n)
        for b in range(batch):
            res = VecZnx(n, cols, rsz)
            ref.blind_rotation_execute(res, k, np.ascontiguousarray(lwe[b]), lut, brk_r, dnum, bsz, blk, xpa)
            assert np.array_equal(got[b], res.data), (it, b)
    # (the workspace-guard mode runs every call plainly: a replay would not re-arm the guards; POULPY_DBG_GRAPHS=0 is the A/B knob for plain launches)
    if os.environ.get("POULPY_DBG_CANARY") != "1" and os.environ.get("POULPY_DBG_GRAPHS") != "0":
        assert hip.graph_launches() - before >= 2, "the repeated call was never served by a graph"
    # switched off: plain launches again, same results
    hip.set_graphs(False)
    mid = hip.graph_launches()
    hip.blind_rotation_execute_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, p, batch)
    hip.sync()
    assert hip.graph_launches() == mid
    assert np.array_equal(d_res.download(np.int64, batch * rsz * cols * n).reshape(batch, rsz, cols, n), got)
    hip.set_graphs(True)
    for buf in (d_lwe, d_lut, d_brk, d_res):
        buf.free()


@pytest.mark.parametrize("n", [32, 1024, 65536])
def test_vec_znx_limbwise_family(mods, n):
    """i64 VecZnx add / sub / negate / copy / zero / normalize[_assign] (hal_impl.rs:34-131, :289) vs the oracle's restatement of
    reference/vec_znx/{add,sub,negate,copy,normalize}.rs: every ordering of the three sizes, other columns untouched, 63-bit
    values (wrapping arithmetic), in-place forms."""
    ref, hip = mods(n)
    rng = seeded(5200 + n)
    sizes = [(3, 2, 4), (3, 4, 2), (2, 3, 3), (4, 1, 1), (1, 3, 2)] if n <= 1024 else [(3, 2, 4), (2, 3, 1)]
    for rs, asz, bsz in sizes:
        a = VecZnx(n, 2, asz).fill_uniform(63, rng)
        b = VecZnx(n, 3, bsz).fill_uniform(63, rng)
        for name in ("vec_znx_add_into", "vec_znx_sub"):
            r1, r2 = VecZnx(n, 2, rs).fill_uniform(20, rng), None
            r2 = r1.copy()
            getattr(ref, name)(r1, 1, a, 0, b, 2)
            getattr(hip, name)(r2, 1, a, 0, b, 2)
            assert np.array_equal(r1.data, r2.data), (name, rs, asz, bsz)
        for name in ("vec_znx_add_assign", "vec_znx_sub_assign", "vec_znx_sub_negate_assign", "vec_znx_negate", "vec_znx_copy"):
            r1 = VecZnx(n, 2, rs).fill_uniform(63, rng)
            r2 = r1.copy()
            getattr(ref, name)(r1, 0, a, 1)
            getattr(hip, name)(r2, 0, a, 1)
            assert np.array_equal(r1.data, r2.data), (name, rs, asz)
        r1 = VecZnx(n, 2, rs).fill_uniform(63, rng)
        r2 = r1.copy()
        ref.vec_znx_negate_assign(r1, 1)
        hip.vec_znx_negate_assign(r2, 1)
        assert np.array_equal(r1.data, r2.data)
        hip.vec_znx_zero(r2, 0)
        assert not r2.data[:, 0].any() and np.array_equal(r2.data[:, 1], r1.data[:, 1])
        # normalize: out of place (cross base) == vec_znx_big_normalize's oracle; in place, same base
        for (rk, ak, off) in ((12, 12, 0), (15, 11, 3), (9, 17, -4)):
            src = VecZnx(n, 2, asz).fill_uniform(50, rng)
            r1 = VecZnx(n, 2, rs).fill_uniform(20, rng)
            r2 = r1.copy()
            ref.vec_znx_big_normalize(r1, rk, off, 1, src, ak, 0)
            hip.vec_znx_normalize(r2, rk, off, 1, src, ak, 0)
            assert np.array_equal(r1.data, r2.data), (rk, ak, off)
        r1 = VecZnx(n, 2, rs).fill_uniform(55, rng)
        r2 = r1.copy()
        ref.vec_znx_normalize_assign(13, r1, 1)
        hip.vec_znx_normalize_assign(13, r2, 1)
        assert np.array_equal(r1.data, r2.data)


@pytest.mark.parametrize("n", [64, 4096])
def test_vec_znx_shifts(mods, n):
    """vec_znx_lsh / vec_znx_rsh / vec_znx_lsh_assign (hal_impl.rs:137-221) vs the oracle's literal restatement of
    reference/vec_znx/shift.rs: shifts from 0 past the total precision, every ordering of the sizes."""
    ref, hip = mods(n)
    rng = seeded(5300 + n)
    base2k = 13
    for rs, asz in [(3, 3), (2, 4), (4, 2)]:
        for k in (0, 1, 7, 13, 14, 26, 30, 39, 40, 60):
            a = VecZnx(n, 2, asz).fill_uniform(45, rng)
            for name in ("vec_znx_lsh", "vec_znx_rsh"):
                r1 = VecZnx(n, 2, rs).fill_uniform(20, rng)
                r2 = r1.copy()
                getattr(ref, name)(base2k, k, r1, 1, a, 0)
                getattr(hip, name)(base2k, k, r2, 1, a, 0)
                assert np.array_equal(r1.data, r2.data), (name, rs, asz, k)
            x1 = VecZnx(n, 2, rs).fill_uniform(45, rng)
            x2 = x1.copy()
            ref.vec_znx_lsh_assign(base2k, k, x1, 0)
            hip.vec_znx_lsh_assign(base2k, k, x2, 0)
            assert np.array_equal(x1.data, x2.data), ("lsh_assign", rs, k)


@pytest.mark.parametrize("n,rank,size,log_gap_out,indices,batch", [
    (64, 1, 3, 2, [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60], 2),   # dense at gap 4: every step merges pairs
    (64, 1, 3, 0, [0, 5, 17, 32, 33, 63], 3),                                        # sparse: all three branches of pack_internal
    (256, 2, 2, 3, [0, 8, 16, 128, 136], 2),                                         # rank 2
    (4096, 1, 3, 9, [0, 512, 1024, 2048, 3584], 2),                                  # fused automorphism pipeline
    (64, 1, 2, 6, [0], 2),                                                           # nothing to pack: the final trace is empty too
])
def test_glwe_pack_batched(mods, n, rank, size, log_gap_out, indices, batch):
    """glwe_packing.rs:122-176 through the C ABI vs the oracle's restatement (tree walk, pack_internal's three cases, final partial
    trace); every ciphertext of the batch is an independent problem with the same occupancy pattern."""
    from poulpy_amd.hal import GlweOpParams
    base2k, dnum = 13, 3
    ref, hip = mods(n)
    rng = seeded(6100 + n + rank + log_gap_out)
    cols = rank + 1
    log_n = n.bit_length() - 1
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    keys_r, keys_d = [], []
    for _ in gals:
        mat = MatZnx(n, dnum, rank, cols, size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, rank, cols, size), hip.vmp_pmat_alloc(dnum, rank, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        keys_r.append(pr)
        keys_d.append(hip.device_alloc(ph.data.nbytes).upload(ph.data))
    data = {j: rng.integers(-(1 << (base2k - 1)), 1 << (base2k - 1), (batch, size, cols, n), dtype=np.int64) for j in indices}
    want = np.empty((batch, size, cols, n), dtype=np.int64)
    for b in range(batch):
        cts = {j: VecZnx(n, cols, size, data[j][b].copy()) for j in indices}
        res = VecZnx(n, cols, size)
        ref.glwe_pack(res, base2k, cts, log_gap_out, gals, keys_r)
        want[b] = res.data
    d_cts = [hip.device_alloc(data[j].nbytes).upload(data[j]) for j in indices]
    d_res = hip.device_alloc(want.nbytes)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=rank)
    nbytes = hip.glwe_pack_tmp_bytes(p, batch)
    d_tmp = hip.device_alloc(nbytes)
    hip.glwe_pack_batched(d_res.ptr, indices, [d.ptr for d in d_cts], log_gap_out, gals, [k.ptr for k in keys_d], p, d_tmp.ptr, nbytes, batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in keys_d + d_cts + [d_res, d_tmp]:
        buf.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,log_gap_in,log_gap_out,log_domain", [
    (256, 1, 4, 2, 2),      # repacking branch of post_process: 4 shifted copies at gap 16 packed to gap 4
    (256, 2, 5, 0, 1),      # rank 2, packed to consecutive coefficients
    (256, 1, 4, 4, 2),      # equal gaps: the partial trace only
    (4096, 1, 9, 7, 1),     # fused automorphism pipeline inside the trace and the pack
])
def test_circuit_bootstrapping_to_exponent(mods, n, rank, log_gap_in, log_gap_out, log_domain):
    """circuit.rs:197-216 + post_process :373-421 (one base2k) through the C ABI vs the oracle's composition (blind rotation, partial
    trace, shifted copies, glwe_pack, ggsw_expand_row); random lookup table and gap."""
    from poulpy_amd.hal import BlindRotationParams, CircuitBootstrappingParams
    base2k, n_lwe, block_size, brk_dnum, glwe_size, res_dnum, res_size, batch, atk_dnum, tsk_dnum = 13, 6, 3, 2, 3, 2, 2, 2, 3, 2
    ref, hip = mods(n)
    rng = seeded(8800 + n + rank + log_gap_out)
    cols = rank + 1
    log_n = n.bit_length() - 1
    gap = 2 * int(rng.integers(1, n // 8))

    def prepared(rows, cols_in, size):
        mat = MatZnx(n, rows, cols_in, cols, size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(rows, cols_in, cols, size), hip.vmp_pmat_alloc(rows, cols_in, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        return pr, ph

    lut = VecZnx(n, 1, glwe_size).fill_uniform(base2k, rng)
    brk_r = np.empty((n_lwe, n * brk_dnum * cols * cols * glwe_size), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        pr, ph = prepared(brk_dnum, cols, glwe_size)
        brk_r[i], brk_h[i] = pr.data.reshape(-1), ph.data.reshape(-1)
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    atk = [prepared(atk_dnum, rank, glwe_size) for _ in gals]
    tsk = [prepared(tsk_dnum, rank, res_size + 1) for _ in range(rank)]
    lwe = rng.integers(-n, n, (batch, n_lwe + 1), dtype=np.int64)
    xpa = ref.blind_rotation_x_pow_a()
    want = np.empty((batch, res_dnum, cols, res_size, cols, n), dtype=np.int64)
    for b in range(batch):
        g = MatZnx(n, res_dnum, cols, cols, res_size)
        ref.circuit_bootstrap_to_exponent(g, base2k, np.ascontiguousarray(lwe[b]), lut, brk_r, brk_dnum, glwe_size, glwe_size, block_size,
                                          xpa, gals, [a[0] for a in atk], [t[0] for t in tsk], gap, log_gap_in, log_gap_out, log_domain)
        want[b] = g.data
    bufs = []

    def up(arr):
        d = hip.device_alloc(arr.nbytes).upload(arr)
        bufs.append(d)
        return d

    d_lwe, d_lut, d_brk = up(lwe), up(lut.data), up(brk_h)
    d_atk = [up(a[1].data) for a in atk]
    d_tsk = [up(t[1].data) for t in tsk]
    d_res = up(rng.integers(-5, 5, want.shape, dtype=np.int64))
    p = CircuitBootstrappingParams(
        br=BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block_size, dnum=brk_dnum, brk_size=glwe_size, base2k=base2k,
                               res_size=glwe_size, lut_size=glwe_size),
        atk_dnum=atk_dnum, atk_size=glwe_size, tsk_dnum=tsk_dnum, tsk_size=res_size + 1, res_dnum=res_dnum, res_size=res_size, gap=gap)
    nbytes = hip.circuit_bootstrapping_to_exponent_tmp_bytes(p, log_domain, batch)
    d_tmp = hip.device_alloc(nbytes)
    bufs.append(d_tmp)
    hip.circuit_bootstrapping_execute_to_exponent_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, gals, [k.ptr for k in d_atk],
                                                          [k.ptr for k in d_tsk], p, log_gap_in, log_gap_out, log_domain, d_tmp.ptr, nbytes,
                                                          batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in bufs:
        buf.free()
    assert np.array_equal(got[:, :, 0], want[:, :, 0]), "packed rows differ"
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,bases,res_limbs,block_size,mode", [
    (256, 1, (13, 11, 12, 15), 2, 3, "constant"),     # the reference's test bases (brk, atk, tsk, res), tests/circuit_bootstrapping.rs:49-53
    (256, 2, (13, 11, 12, 15), 2, 2, "constant"),     # rank 2
    (512, 1, (12, 17, 12, 9), 3, 1, "constant"),      # automorphism keys in the largest base, standard rotation
    (256, 1, (13, 11, 12, 15), 2, 3, "exponent"),     # post_process with equal gaps (partial trace)
    (256, 1, (13, 11, 12, 15), 2, 3, "repack"),       # post_process with repacking (tests/circuit_bootstrapping.rs:36-227)
    (2048, 1, (13, 11, 12, 15), 2, 3, "constant"),    # small-ring transforms in the rotation, two-kernel pipeline in the trace
    (256, 1, (13, 13, 13, 15), 2, 3, "constant"),     # only the result in another base
])
def test_circuit_bootstrapping_one_base2k_per_object(mods, n, rank, bases, res_limbs, block_size, mode):
    """circuit.rs:219-421 the way the reference's own tests run it: the blind-rotation key, the automorphism keys, the tensor keys and the
    result each in their own base2k (glwe_normalize between the layouts, glwe_trace's closing normalize, ggsw_expand_row's conversion);
    through the C ABI vs the oracle (tests/test_oracle_cbt.py pins that one on the composition of the primitives)."""
    from poulpy_amd.hal import BlindRotationParams, CircuitBootstrappingParams
    from tests.test_oracle_cbt import cbt_shape, make_cbt_inputs
    k_brk, k_atk, k_tsk, k_res = bases
    n_lwe, brk_dnum, res_dnum, batch = 6, 2, 2, 3
    ref, hip = mods(n)
    rng = seeded(4400 + n + rank + sum(bases))
    cols = rank + 1
    log_n = n.bit_length() - 1
    sh = cbt_shape(k_res, k_brk, k_tsk, k_atk, res_limbs)
    atk_dnum, tsk_dnum = sh["trace_size"], sh["res_conv_size"]

    def prep_hip(rows, cols_in, size, mat):
        ph = hip.vmp_pmat_alloc(rows, cols_in, cols, size)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        return ph

    lut, brk, gals, atk, tsk = make_cbt_inputs(ref, n, rank, n_lwe, brk_dnum, bases, sh, atk_dnum, tsk_dnum, rng, prepare_also=prep_hip)
    brk_r = np.stack([b[0].data.reshape(-1) for b in brk])
    brk_h = np.stack([b[1].data.reshape(-1) for b in brk])
    lwe = rng.integers(-n, n, (batch, n_lwe + 1), dtype=np.int64)
    gap = 2 * int(rng.integers(1, n // 8))
    log_gap_in, log_gap_out, log_domain = {"constant": (0, 0, 0), "exponent": (4, 4, 2), "repack": (4, 2, 2)}[mode]
    xpa = ref.blind_rotation_x_pow_a() if block_size > 1 else np.zeros((1, 1))
    want = np.empty((batch, res_dnum, cols, sh["res_size"], cols, n), dtype=np.int64)
    for b in range(batch):
        g = MatZnx(n, res_dnum, cols, cols, sh["res_size"])
        ref.circuit_bootstrap_bases(g, bases, mode != "constant", np.ascontiguousarray(lwe[b]), lut, brk_r, brk_dnum, sh["glwe_size"],
                                    sh["glwe_size"], sh["atk_glwe_size"], sh["trace_size"], block_size, xpa, gals, [a[0] for a in atk],
                                    [t[0] for t in tsk], gap, log_gap_in, log_gap_out, log_domain)
        want[b] = g.data
    bufs = []

    def up(arr):
        d = hip.device_alloc(arr.nbytes).upload(arr)
        bufs.append(d)
        return d

    d_lwe, d_lut, d_brk = up(lwe), up(lut.data), up(brk_h)
    d_atk = [up(a[1].data) for a in atk]
    d_tsk = [up(t[1].data) for t in tsk]
    d_res = up(rng.integers(-5, 5, want.shape, dtype=np.int64))
    p = CircuitBootstrappingParams(
        br=BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block_size, dnum=brk_dnum, brk_size=sh["glwe_size"], base2k=k_brk,
                               res_size=sh["glwe_size"], lut_size=sh["glwe_size"]),
        atk_dnum=atk_dnum, atk_size=sh["atk_size"], tsk_dnum=tsk_dnum, tsk_size=sh["tsk_size"], res_dnum=res_dnum, res_size=sh["res_size"],
        gap=gap, atk_base2k=k_atk, tsk_base2k=k_tsk, res_base2k=k_res, atk_glwe_size=sh["atk_glwe_size"], trace_size=sh["trace_size"])
    if mode == "constant":
        nbytes = hip.circuit_bootstrapping_tmp_bytes(p, batch)
        d_tmp = up(np.zeros(nbytes // 8 + 1, dtype=np.int64))
        hip.circuit_bootstrapping_execute_to_constant_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, gals, [k.ptr for k in d_atk],
                                                              [k.ptr for k in d_tsk], p, d_tmp.ptr, nbytes, batch)
    else:
        nbytes = hip.circuit_bootstrapping_to_exponent_tmp_bytes(p, log_domain, batch)
        d_tmp = up(np.zeros(nbytes // 8 + 1, dtype=np.int64))
        hip.circuit_bootstrapping_execute_to_exponent_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, gals, [k.ptr for k in d_atk],
                                                              [k.ptr for k in d_tsk], p, log_gap_in, log_gap_out, log_domain, d_tmp.ptr,
                                                              nbytes, batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in bufs:
        buf.free()
    assert np.array_equal(got[:, :, 0], want[:, :, 0]), "rows before ggsw_expand_row differ"
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,ext,n_lwe,blk,dnum,bsz,rsz,batch,fuse", [
    (256, 1, 2, 7, 3, 2, 2, 2, 5, True),
    (256, 1, 4, 6, 2, 2, 3, 2, 4, True),
    (512, 2, 8, 4, 2, 3, 2, 3, 3, True),
    (4096, 1, 2, 4, 2, 2, 2, 2, 2, True),
    (2048, 1, 4, 6, 3, 3, 3, 3, 3, True),      # transforms of the small-ring pipeline around the per-coefficient steps
    (1024, 2, 2, 4, 2, 2, 3, 2, 5, True),
    (2048, 1, 2, 4, 2, 2, 2, 4, 2, True),      # result limbs beyond the key precision
    (256, 1, 4, 6, 3, 2, 2, 3, 3, False),
    (512, 1, 2, 224, 7, 2, 3, 2, 2, True),     # blind_rotation/tests/fft64_ref.rs:36-40 `block_binary_extended` (base2k 19, all 224 coefficients)
])
def test_blind_rotation_extended(mods, n, rank, ext, n_lwe, blk, dnum, bsz, rsz, batch, fuse):
    """execute_block_binary_extended (algorithm.rs:121-273, extension_factor > 1) vs the oracle's literal restatement; the LWE
    values include every special case of the reference (a = 0, a multiple of ext, ai_hi = 0 with ai_lo != 0, ai_hi + 1 = 2n)."""
    from poulpy_amd.hal import BlindRotationParams
    k = 19 if n_lwe == 224 else 13
    ref, hip = mods(n)
    rng = seeded(9900 + n + ext)
    cols = rank + 1
    luts = rng.integers(-(1 << (k - 1)), 1 << (k - 1), (ext, rsz, 1, n), dtype=np.int64)
    brk_r = np.empty((n_lwe, n * dnum * cols * cols * bsz), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        mat = MatZnx(n, dnum, cols, cols, bsz).fill_uniform(k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, bsz), hip.vmp_pmat_alloc(dnum, cols, cols, bsz)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        brk_r[i], brk_h[i] = pr.data.reshape(-1), ph.data.reshape(-1)
    lwe = rng.integers(-n * ext, n * ext, (batch, n_lwe + 1), dtype=np.int64)   # mod_switch_2n(2 n ext) output range
    lwe[0, 1] = 0                       # contributes nothing
    lwe[0, 2] = ext * 5                 # ai_lo = 0
    lwe[1 % batch, 1] = ext - 1         # ai_hi = 0, ai_lo != 0: the second half is skipped
    lwe[1 % batch, 2] = 2 * n * ext - 1  # ai_hi + 1 = 2n: the first half is skipped
    lwe[-1, 0] = n * ext - 1
    xpa = ref.blind_rotation_x_pow_a()
    want = np.empty((batch, rsz, cols, n), dtype=np.int64)
    for b in range(batch):
        res = VecZnx(n, cols, rsz)
        ref.blind_rotation_execute_extended(res, k, np.ascontiguousarray(lwe[b]), luts, brk_r, dnum, bsz, blk, xpa)
        want[b] = res.data
    d_lwe = hip.device_alloc(lwe.nbytes).upload(lwe)
    d_lut = hip.device_alloc(luts.nbytes).upload(luts)
    d_brk = hip.device_alloc(brk_h.nbytes).upload(brk_h)
    d_res = hip.device_alloc(want.nbytes)
    p = BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=blk, dnum=dnum, brk_size=bsz, base2k=k, res_size=rsz, lut_size=rsz)
    nbytes = hip.blind_rotation_extended_tmp_bytes(p, ext, batch)
    d_tmp = hip.device_alloc(nbytes)
    hip.set_fusion(fuse, fuse)
    hip.blind_rotation_execute_extended_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, p, ext, d_tmp.ptr, nbytes, batch)
    hip.sync()
    hip.set_fusion(True, True)
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_lwe, d_lut, d_brk, d_res, d_tmp):
        buf.free()
    assert np.array_equal(got, want)


def test_circuit_bootstrapping_with_extension_factor(mods):
    """circuit.rs:219-370 with extension_factor 2 (the rotation dispatches to execute_block_binary_extended, algorithm.rs:76-118):
    expected GGSW composed in Python from the oracle's pieces (extended rotation, rotate, glwe_trace, ggsw_expand_row)."""
    from poulpy_amd.hal import BlindRotationParams, CircuitBootstrappingParams
    n, rank, ext, n_lwe, blk, brk_dnum, gsz, res_dnum, rsz, batch, base2k, atk_dnum, tsk_dnum = 256, 1, 2, 6, 3, 2, 3, 2, 2, 3, 13, 3, 2
    ref, hip = mods(n)
    rng = seeded(31337)
    cols = rank + 1
    log_n = n.bit_length() - 1
    gap = 2 * int(rng.integers(1, n // 8))

    def prepared(rows, cols_in, size):
        mat = MatZnx(n, rows, cols_in, cols, size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(rows, cols_in, cols, size), hip.vmp_pmat_alloc(rows, cols_in, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        hip.sync()
        return pr, ph

    luts = rng.integers(-(1 << (base2k - 1)), 1 << (base2k - 1), (ext, gsz, 1, n), dtype=np.int64)
    brk_r = np.empty((n_lwe, n * brk_dnum * cols * cols * gsz), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        pr, ph = prepared(brk_dnum, cols, gsz)
        brk_r[i], brk_h[i] = pr.data.reshape(-1), ph.data.reshape(-1)
    gals = [-1] + [pow(5, 1 << i, 2 * n) for i in range(log_n - 1)]
    atk = [prepared(atk_dnum, rank, gsz) for _ in gals]
    tsk = [prepared(tsk_dnum, rank, rsz + 1) for _ in range(rank)]
    lwe = rng.integers(-n * ext, n * ext, (batch, n_lwe + 1), dtype=np.int64)
    xpa = ref.blind_rotation_x_pow_a()
    want = np.empty((batch, res_dnum, cols, rsz, cols, n), dtype=np.int64)
    for b in range(batch):
        acc = VecZnx(n, cols, gsz)
        ref.blind_rotation_execute_extended(acc, base2k, np.ascontiguousarray(lwe[b]), luts, brk_r, brk_dnum, gsz, blk, xpa)
        g = MatZnx(n, res_dnum, cols, cols, rsz)
        for i in range(res_dnum):
            row = acc.copy()
            ref.glwe_trace_assign(row, base2k, gals, [a[0] for a in atk])
            g.data[i, 0] = row.data[:rsz]
            if i + 1 < res_dnum:
                nxt = acc.copy()
                for c in range(cols):
                    ref.vec_znx_rotate(-gap, nxt, c, acc, c)
                acc = nxt
        ref.ggsw_expand_row(g, base2k, [t[0] for t in tsk], 1, base2k)
        want[b] = g.data
    bufs = []

    def up(arr):
        d = hip.device_alloc(arr.nbytes).upload(arr)
        bufs.append(d)
        return d

    d_lwe, d_lut, d_brk = up(lwe), up(luts), up(brk_h)
    d_atk = [up(a[1].data) for a in atk]
    d_tsk = [up(t[1].data) for t in tsk]
    d_res = up(np.zeros(want.shape, dtype=np.int64))
    p = CircuitBootstrappingParams(
        br=BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=blk, dnum=brk_dnum, brk_size=gsz, base2k=base2k, res_size=gsz, lut_size=gsz),
        atk_dnum=atk_dnum, atk_size=gsz, tsk_dnum=tsk_dnum, tsk_size=rsz + 1, res_dnum=res_dnum, res_size=rsz, gap=gap, extension_factor=ext)
    nbytes = hip.circuit_bootstrapping_tmp_bytes(p, batch)
    d_tmp = hip.device_alloc(nbytes)
    bufs.append(d_tmp)
    hip.circuit_bootstrapping_execute_to_constant_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, gals, [k.ptr for k in d_atk],
                                                          [k.ptr for k in d_tsk], p, d_tmp.ptr, nbytes, batch)
    hip.sync()
    got = d_res.download(np.int64, want.size).reshape(want.shape)
    for buf in bufs:
        buf.free()
    assert np.array_equal(got, want)


def test_seeded_glwe_pack_sweep(mods):
    """20 random packing problems (fixed seed; POULPY_SWEEP_SEED / POULPY_SWEEP_CASES by hand): ring 2^5..2^7, rank 1-2, 2-3 limbs,
    random output gap and random occupancy on the grid of that gap (index 0 always present so that the tree ends there)."""
    import os
    rng = np.random.default_rng(int(os.environ.get("POULPY_SWEEP_SEED", "515")))
    for case in range(int(os.environ.get("POULPY_SWEEP_CASES", "20"))):
        n = int(2 ** rng.integers(5, 8))
        log_n = n.bit_length() - 1
        rank = int(rng.integers(1, 3))
        size = int(rng.integers(2, 4))
        log_gap_out = int(rng.integers(0, log_n + 1))
        grid = list(range(0, n, 1 << log_gap_out))
        extra = [int(x) for x in rng.choice(grid[1:], size=int(rng.integers(0, min(len(grid) - 1, 9) + 1)), replace=False)] if len(grid) > 1 else []
        indices = [0] + sorted(extra)
        batch = int(rng.integers(1, 4))
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(dict(case=case, n=n, rank=rank, size=size, log_gap_out=log_gap_out, indices=indices, batch=batch), flush=True)
        test_glwe_pack_batched(mods, n, rank, size, log_gap_out, indices, batch)


def test_seeded_extended_rotation_sweep(mods):
    """15 random extended blind rotations (fixed seed): ring 2^7..2^9, rank 1-2, extension factor 2-8, block size 2-4, 1-3 limbs,
    ragged batches, both the fused-tail and the op-by-op epilogue."""
    import os
    rng = np.random.default_rng(int(os.environ.get("POULPY_SWEEP_SEED", "616")))
    for case in range(int(os.environ.get("POULPY_SWEEP_CASES", "15"))):
        n = int(2 ** rng.integers(7, 10))
        rank = int(rng.integers(1, 3))
        ext = int(2 ** rng.integers(1, 4))
        blk = int(rng.integers(2, 5))
        n_lwe = int(rng.integers(1, 3)) * blk + int(rng.integers(0, blk))
        dnum, bsz, rsz = (int(x) for x in rng.integers(1, 4, 3))
        batch = int(rng.integers(1, 5))
        fuse = bool(rng.integers(0, 2))
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(dict(case=case, n=n, rank=rank, ext=ext, blk=blk, n_lwe=n_lwe, dnum=dnum, bsz=bsz, rsz=rsz, batch=batch, fuse=fuse), flush=True)
        test_blind_rotation_extended(mods, n, rank, ext, max(n_lwe, 2), blk, dnum, bsz, rsz, batch, fuse)


@pytest.mark.parametrize("in_place", [False, True], ids=["out-of-place", "in-place"])
@pytest.mark.parametrize("mode", ["add", "sub", "sub_negate"])
@pytest.mark.parametrize("n,rank,p", [(4096, 1, 5), (4096, 2, 13), (8192, 1, 5 ** 7), (4096, 1, 2 * 4096 - 3), (65536, 1, 5 ** 9)])
def test_glwe_automorphism_spectral_path(mods, n, rank, p, mode, in_place):
    """glwe_automorphism_add / _sub / _sub_negate with Galois elements = 1 mod 4 on the fused pipeline: the permutation is folded
    into the middle kernel's spectrum position (k_mid128<.., PERM>) and the tail writes the final result.  Every mode, ranks 1-2,
    small and large elements, the in-place (`_assign`) forms; and the generic scheme (POULPY-level: fusion off) must agree."""
    limbs = 3 if n < 65536 else 8
    got, want = _run_glwe_op(hip=mods(n)[1], ref=mods(n)[0], ks=True, n=n, rank=rank, rank_out=rank, a_size=limbs, a_base2k=12, key_size=limbs,
                             key_base2k=12, dnum=limbs, dsize=1, res_size=limbs, res_base2k=12, batch=3, seed=4200 + rank, auto=(p % (2 * n), mode),
                             in_place=in_place)
    assert np.array_equal(got, want)



@pytest.mark.parametrize("mode", ["automorphism", "add", "sub", "sub_negate"])
@pytest.mark.parametrize("n,p", [(8192, 5), (65536, 5 ** 9), (8192, 3)])
def test_glwe_automorphism_body_as_16_bit_copies_and_wide_inputs(mods, n, p, mode):
    """Round 6: on the spectral forms the pre-pass leaves the body-column operand (+-phi(body) [+ a0]) as 16-bit values in the tail's tile order
    (k_automorphism / _chunk with flags & 8, TailArgs::body16_wide); a value beyond 16 bits - an un-normalized input - raises the device flag and
    the wave runs the i64 scheme (a conditional second pre-pass + the operand variant of the tail).  Both branches against the oracle, out of place and in place: normalized digits at base2k 12 and 15 / 16 (the widest the copies
    take), one ciphertext of the batch with 20-bit digits (flag up for that wave: chunk = 2 keeps the other waves on the copies), and a
    Galois element = 3 mod 4 (p = 3: the conjugating spectral form for add / sub; the plain form keeps its permutation pass there)."""
    ref, hip = mods(n)
    limbs = 3 if n < 65536 else 4
    for in_place in (False, True):
        for (k, wide) in ((12, None), (15, None), (16 if mode == "automorphism" else 15, None), (12, (1, 20)), (12, (2, 17))):
            got, want = _run_glwe_op(hip, ref, True, n, 1, 1, limbs, k, limbs, k, limbs, 1, limbs, k, batch=4, seed=8100 + k + (wide[1] if wide else 0),
                                     auto=(p % (2 * n), mode), chunk=2, wide_in=wide, in_place=in_place)
            assert np.array_equal(got, want), (n, p, mode, k, wide, in_place)
    if n == 8192:   # rank 2: two body-less columns beside the body column
        for wide in (None, (0, 18)):
            got, want = _run_glwe_op(hip, ref, True, n, 2, 2, 3, 13, 3, 13, 3, 1, 3, 13, batch=3, seed=8300 + (wide[1] if wide else 0),
                                     auto=(p % (2 * n), mode), wide_in=wide)
            assert np.array_equal(got, want), (n, p, mode, "rank 2", wide)


@pytest.mark.parametrize("n,a_size,key_size,res_size,batch,chunk", [(4096, 8, 2, 1, 16, 0), (65536, 8, 2, 1, 5, 4), (8192, 6, 3, 1, 9, 0)])
def test_glwe_automorphism_spectral_path_long_input(mods, n, a_size, key_size, res_size, batch, chunk):
    """the spectral form with MORE input limbs than cols_out * res_size result limbs (the body operand of the tail has its own
    workspace segment, min(a_size, key_size) limbs per ciphertext: round-1 advisor finding), several waves, every mode"""
    for mode in ("add", "sub", "sub_negate"):
        got, want = _run_glwe_op(hip=mods(n)[1], ref=mods(n)[0], ks=True, n=n, rank=1, rank_out=1, a_size=a_size, a_base2k=12, key_size=key_size,
                                 key_base2k=12, dnum=key_size, dsize=1, res_size=res_size, res_base2k=12, batch=batch, seed=5100 + a_size,
                                 auto=(5, mode), chunk=chunk)
        assert np.array_equal(got, want), mode


# ------------------------------------------------------------------------------------------
# host containers at the GLWE-level entry points (what the Rust shim's CoreImpl overrides pass)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [256, 4096, 65536])
def test_glwe_ops_on_host_containers(mods, n):
    """pz_glwe_{external_product,keyswitch,automorphism}_batched / pz_glwe_tensor_relinearize_batched with HOST ciphertexts and a
    HOST-resident prepared key: staged ciphertexts, device mirror of the key (re-used across calls, refreshed when the host bytes
    change, dropped by vmp_prepare and pz_module_forget_host_key), in-place forms, mixed host / device arguments."""
    import ctypes as C
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(n + 77)
    rank, size, base2k, dnum = 1, (8 if n == 65536 else 3), 12, (8 if n == 65536 else 3)
    cols = rank + 1
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=rank)
    batch = 2
    for kind in ("external_product", "keyswitch", "automorphism_add", "relinearize"):
        ks = kind != "external_product"
        cols_in = rank if ks else cols
        a_cols = cols + 1 if kind == "relinearize" else cols
        mat = MatZnx(n, dnum, cols_in, cols, size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, cols_in, cols, size), hip.vmp_pmat_alloc(dnum, cols_in, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        a_all = np.empty((batch, size, a_cols, n), dtype=np.int64)
        want = np.empty((batch, size, cols, n), dtype=np.int64)

        def oracle(pm):
            for t in range(batch):
                a = VecZnx(n, a_cols, size, a_all[t].copy())
                r = VecZnx(n, cols, size)
                if kind == "external_product":
                    ref.glwe_external_product(r, base2k, a, base2k, pm, 1, base2k)
                elif kind == "keyswitch":
                    ref.glwe_keyswitch(r, base2k, a, base2k, pm, 1, base2k)
                elif kind == "automorphism_add":
                    ref.glwe_automorphism(r, base2k, a, base2k, pm, 1, base2k, 5, "add")
                else:
                    ref.glwe_tensor_relinearize(r, base2k, a, base2k, pm, 1, base2k)
                want[t] = r.data

        def run(res_arr, a_arr, key_ptr):
            rp = res_arr if isinstance(res_arr, C.c_void_p) else hp(res_arr)
            ap = a_arr if isinstance(a_arr, C.c_void_p) else hp(a_arr)
            if kind == "external_product":
                hip.glwe_external_product_batched(rp, ap, key_ptr, p, batch)
            elif kind == "keyswitch":
                hip.glwe_keyswitch_batched(rp, ap, key_ptr, p, batch)
            elif kind == "automorphism_add":
                hip.glwe_automorphism_batched(rp, ap, key_ptr, p, 5, "add", batch)
            else:
                hip.glwe_tensor_relinearize_batched(rp, ap, key_ptr, p, batch)

        for t in range(batch):
            a_all[t] = VecZnx(n, a_cols, size).fill_uniform(base2k, rng).data
        oracle(pr)
        got = np.full_like(want, 0x5A)
        a_before = a_all.copy()
        run(got, a_all, hp(ph.data))                 # everything on the host: first use uploads the key mirror
        assert np.array_equal(got, want) and np.array_equal(a_all, a_before), kind
        got[...] = 1
        run(got, a_all, hp(ph.data))                 # second use: mirror re-used
        assert np.array_equal(got, want), kind
        if a_cols == cols:                           # *_assign form: res is the input container
            inout = a_all.copy()
            run(inout, inout, hp(ph.data))
            assert np.array_equal(inout, want), kind
        # mixed: device ciphertexts, host key
        d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
        d_r = hip.device_alloc(want.nbytes)
        run(d_r.ptr, d_a.ptr, hp(ph.data))
        hip.sync()
        assert np.array_equal(d_r.download(np.int64, want.size).reshape(want.shape), want), kind
        # the host key changes: re-prepared in place (mirror dropped by vmp_prepare) ...
        mat2 = MatZnx(n, dnum, cols_in, cols, size).fill_uniform(base2k, rng)
        ref.vmp_prepare(pr, mat2)
        hip.vmp_prepare(ph, mat2)
        oracle(pr)
        run(got, a_all, hp(ph.data))
        assert np.array_equal(got, want), kind
        # ... or overwritten behind the backend's back (fingerprint mismatch -> refreshed), here with the first key again
        ph1 = hip.vmp_pmat_alloc(dnum, cols_in, cols, size)
        hip.vmp_prepare(ph1, mat)
        ph.data[...] = ph1.data
        ref.vmp_prepare(pr, mat)
        oracle(pr)
        run(got, a_all, hp(ph.data))
        assert np.array_equal(got, want), kind
        assert hip.lib.pz_module_forget_host_key(hip.handle, hp(ph.data)) == 0
        run(got, a_all, hp(ph.data))
        assert np.array_equal(got, want), kind
        assert hip.lib.pz_module_forget_host_key(hip.handle, hp(ph.data)) == 0
        # a key inside a pz_alloc_bytes block (Backend::OwnedBuf of the Rust shim): releasing the block drops its mirror
        lib = hip.lib
        lib.pz_module_host_key_mirrors.restype = C.c_size_t
        lib.pz_module_host_key_mirrors.argtypes = [C.c_void_p]
        before = lib.pz_module_host_key_mirrors(hip.handle)
        blk = lib.pz_alloc_bytes(ph.data.nbytes + 4096)
        inner = np.ctypeslib.as_array(C.cast(blk + 4096, C.POINTER(C.c_double)), shape=(ph.data.size,))
        inner[...] = ph.data.reshape(-1)
        run(got, a_all, C.c_void_p(blk + 4096))
        assert np.array_equal(got, want), kind
        assert lib.pz_module_host_key_mirrors(hip.handle) == before + 1
        lib.pz_free_bytes(C.c_void_p(blk))
        assert lib.pz_module_host_key_mirrors(hip.handle) == before
        for buf in (d_a, d_r):
            buf.free()


@pytest.mark.parametrize("n,size", [(4096, 3), (65536, 8)])
def test_glwe_ops_on_pinned_host_containers_duplex(mods, n, size):
    """The duplex host path (round 5): pinned host ciphertexts (pz_alloc_bytes: what the Rust shim's buffers are), several
    per call - the call runs as waves on three streams (upload | kernels | download).  Ragged batch (11 ciphertexts: waves of 2, the last of 1),
    out of place and in place (`*_assign`), a host-resident key; every ciphertext against the oracle, and the bytes beyond the batch untouched.
    Contract: logically synchronous (poulpy-hal/docs/backend_safety_contract.md:3-19) - the results are in host memory when the call returns."""
    import ctypes as C
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(n + 811)
    rank, base2k, dnum, batch = 1, 12, size, 11
    cols = rank + 1

    def pinned(shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = hip.lib.pz_alloc_bytes(C.c_size_t(nbytes))
        assert ptr
        return np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype).reshape(shape), ptr
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=rank)
    held = []
    try:
        for ks in (False, True):
            cols_in = rank if ks else cols
            mat = MatZnx(n, dnum, cols_in, cols, size).fill_uniform(base2k, rng)
            pr, ph = ref.vmp_pmat_alloc(dnum, cols_in, cols, size), hip.vmp_pmat_alloc(dnum, cols_in, cols, size)
            ref.vmp_prepare(pr, mat)
            hip.vmp_prepare(ph, mat)
            key_host, kp = pinned(ph.data.shape, np.float64)
            held.append(kp)
            key_host[...] = ph.data
            a_host, ap = pinned((batch + 1, size, cols, n), np.int64)      # one guard ciphertext behind the batch
            r_host, rp = pinned((batch + 1, size, cols, n), np.int64)
            held += [ap, rp]
            want = np.empty((batch, size, cols, n), dtype=np.int64)
            for t in range(batch):
                a = VecZnx(n, cols, size).fill_uniform(base2k, rng)
                a_host[t] = a.data
                r = VecZnx(n, cols, size)
                (ref.glwe_keyswitch if ks else ref.glwe_external_product)(r, base2k, a, base2k, pr, 1, base2k)
                want[t] = r.data
            a_host[batch] = 0x1111
            r_host[...] = 0x5A5A
            call = hip.glwe_keyswitch_batched if ks else hip.glwe_external_product_batched
            call(hp(r_host), hp(a_host), hp(key_host), p, batch)           # no sync call: the results must be there already
            assert np.array_equal(r_host[:batch], want), "duplex host path, out of place"
            assert (r_host[batch] == 0x5A5A).all() and (a_host[batch] == 0x1111).all()
            # same call again (graphs / key mirror re-used), then in place
            r_host[...] = 0
            call(hp(r_host), hp(a_host), hp(key_host), p, batch)
            assert np.array_equal(r_host[:batch], want)
            call(hp(a_host), hp(a_host), hp(key_host), p, batch)
            assert np.array_equal(a_host[:batch], want), "duplex host path, in place"
            assert (a_host[batch] == 0x1111).all()
            hip.forget_host_key(hp(key_host)) if hasattr(hip, "forget_host_key") else None
    finally:
        hip.sync()
        for ptr in held:
            hip.lib.pz_free_bytes(C.c_void_p(ptr))


def test_pinned_host_in_place_mismatch_and_partial_overlap(mods):
    """the duplex host path runs in front of glwe_args_in and used to skip its in-place check - res == a with a larger res
    layout let every wave write past the a-sized arena block.  Now: PZ_ERR_INVALID before anything is launched (as on the serial path), and host
    ranges that overlap without being equal take the serial path (whole input on the device before the first result travels back): correct
    results for the non-overlapped... whole batch."""
    import ctypes as C
    from poulpy_amd.hal import GlweOpParams, PoulpyHipError
    n, size, rank, base2k, batch = 4096, 3, 1, 12, 6
    ref, hip = mods(n)
    rng = seeded(9157)
    cols = rank + 1

    def pinned(shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = hip.lib.pz_alloc_bytes(C.c_size_t(nbytes))
        assert ptr
        return np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype).reshape(shape), ptr
    hp = lambda arr: arr.ctypes.data_as(C.c_void_p)
    held = []
    try:
        mat = MatZnx(n, size, cols, cols, size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(size, cols, cols, size), hip.vmp_pmat_alloc(size, cols, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        key_host, kp = pinned(ph.data.shape, np.float64)
        held.append(kp)
        key_host[...] = ph.data
        # (1) in place with res_size > a_size: refused, nothing written
        big, bp = pinned((batch, size + 1, cols, n), np.int64)
        held.append(bp)
        big[...] = 0x3C3C
        p_bad = GlweOpParams(rank=rank, dnum=size, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size + 1,
                             res_base2k=base2k, rank_out=rank)
        with pytest.raises(PoulpyHipError, match="in-place"):
            hip.glwe_external_product_batched(hp(big), hp(big), hp(key_host), p_bad, batch)
        assert (big == 0x3C3C).all()
        # (2) res overlaps a, shifted by one ciphertext: the serial path, every result against the oracle
        p = GlweOpParams(rank=rank, dnum=size, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                         res_base2k=base2k, rank_out=rank)
        buf, pp = pinned((batch + 1, size, cols, n), np.int64)
        held.append(pp)
        want = np.empty((batch, size, cols, n), dtype=np.int64)
        for t in range(batch):
            a = VecZnx(n, cols, size).fill_uniform(base2k, rng)
            buf[t + 1] = a.data
            r = VecZnx(n, cols, size)
            ref.glwe_external_product(r, base2k, a, base2k, pr, 1, base2k)
            want[t] = r.data
        hip.glwe_external_product_batched(hp(buf[0:]), hp(buf[1:]), hp(key_host), p, batch)   # res = a - one ciphertext
        assert np.array_equal(buf[:batch], want), "partially overlapping pinned host ranges"
    finally:
        hip.sync()
        for ptr in held:
            hip.lib.pz_free_bytes(C.c_void_p(ptr))


@pytest.mark.parametrize("n,rank,blk,dnum,bsz,rsz,k", [
    (2048, 1, 3, 2, 3, 2, 15),     # small-ring path (chained forward transform, 32-bit digits between blocks)
    (1024, 2, 3, 2, 3, 3, 14),     # rank 2 at N = 1024: small-ring path with the 9 / 6-row block step
    (4096, 1, 3, 2, 3, 2, 15),     # pipeline path, plan 16 x 128
    (8192, 1, 2, 3, 3, 3, 13),     # pipeline path, plan 32 x 128
])
@pytest.mark.parametrize("nblocks,extra", [(1, 0), (2, 0), (3, 1), (4, 2)])
def test_blind_rotation_block_counts_and_accumulator_forms(mods, n, rank, blk, dnum, bsz, rsz, k, nblocks, extra):
    """Round 5: between two blocks the accumulator lives as 32-bit digits in the workspace - the caller's `res` is the operand of the FIRST block and
    the destination of the LAST.  One block (no 32-bit form at all), two (first -> last), three and four (middle blocks read and write the digits), with a
    dropped partial block behind them (chunks_exact, algorithm.rs:303); ragged batch.  Every limb against the oracle."""
    from tests.test_gpu_parity import _run_blind_rotation
    ref, hip = mods(n)
    n_lwe = nblocks * blk + extra
    got, want = _run_blind_rotation(hip, ref, n, rank, n_lwe, blk, dnum, bsz, rsz, k, batch=5, seed=n + 31 * nblocks + rank)
    assert np.array_equal(got, want)
