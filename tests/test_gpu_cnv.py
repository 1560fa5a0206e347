"""GPU parity of the convolution family and of GLWE tensoring / relinearization (BASELINE configs[4])
against the CPU oracle, through the C ABI, bit-exact on the normalized i64 limbs.

Mirrors poulpy-hal/src/test_suite/convolution.rs (test_convolution, test_convolution_pairwise, test_convolution_by_const:
a_size = b_size = 15, every cnv_offset, every column pair) — DFT-domain values are never compared, results go through
idft + normalize — and poulpy-core's tensoring callers (operations/glwe.rs:541-913).
"""
import os

import numpy as np
import pytest

from poulpy_amd.layouts import MatZnx, VecZnx, VecZnxBig, VecZnxDft
from tests.helpers import seeded

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


def _finish(mod, d, base2k, res_size):
    big = VecZnxBig(d.n, 1, res_size)
    mod.vec_znx_idft_apply_tmpa(big, 0, d, 0)
    out = VecZnx(d.n, 1, res_size)
    out.data[...] = 0x5A5A
    mod.vec_znx_big_normalize(out, base2k, 0, 0, big, base2k, 0)
    return out.data


@pytest.mark.parametrize("n,base2k", [(8, 17), (8, 12), (16, 17), (256, 12), (64, 17), (4096, 12)])   # (8, 17): poulpy-cpu-ref/src/tests.rs:11-27
def test_convolution_and_pairwise(mods, n, base2k):
    """test_suite/convolution.rs:91-252 shapes: 2 columns, a_size = b_size = 15, res_size = 30, every offset / column pair."""
    ref, hip = mods(n)
    rng = seeded(n + base2k)
    cols, a_size, b_size = 2, 15, 15
    res_size = a_size + b_size
    a = VecZnx(n, cols, a_size).fill_uniform(17, rng)
    b = VecZnx(n, cols, b_size).fill_uniform(17, rng)
    ar, br = ref.cnv_pvec_left_alloc(cols, a_size), ref.cnv_pvec_right_alloc(cols, b_size)
    ah, bh = hip.cnv_pvec_left_alloc(cols, a_size), hip.cnv_pvec_right_alloc(cols, b_size)
    for (m_, x, y) in ((ref, ar, br), (hip, ah, bh)):
        m_.cnv_prepare_left(x, a, -1)
        m_.cnv_prepare_right(y, b, -1)
    offsets = range(res_size) if n <= 256 else (0, 1, 7, 14, 29)
    for i in range(cols):
        for j in range(cols):
            for off in offsets:
                dr, dh = VecZnxDft(n, 1, res_size), VecZnxDft(n, 1, res_size)
                dh.data[...] = 3.25     # garbage: every limb must be written
                ref.cnv_apply_dft(off, dr, 0, ar, i, br, j)
                hip.cnv_apply_dft(off, dh, 0, ah, i, bh, j)
                assert np.array_equal(_finish(hip, dh, base2k, res_size), _finish(ref, dr, base2k, res_size)), ("apply", i, j, off)
                dr, dh = VecZnxDft(n, 1, res_size), VecZnxDft(n, 1, res_size)
                ref.cnv_pairwise_apply_dft(off, dr, 0, ar, br, i, j)
                hip.cnv_pairwise_apply_dft(off, dh, 0, ah, bh, i, j)
                assert np.array_equal(_finish(hip, dh, base2k, res_size), _finish(ref, dr, base2k, res_size)), ("pairwise", i, j, off)


@pytest.mark.parametrize("n", [8, 512])
def test_convolution_shapes_masks_and_columns(mods, n):
    """res shorter / longer than the product, prepared operands longer than the input (zero limbs) or shorter (the mask moves to
    the last ACTIVE limb), masks, prepare_self, a res with two columns (only res_col written), offsets beyond the product."""
    ref, hip = mods(n)
    rng = seeded(99)
    base2k = 13
    for (cols, a_size, b_size, pa_size, pb_size, res_size, mask) in ((1, 4, 3, 4, 3, 5, -1), (3, 5, 5, 3, 7, 9, -(1 << 4)),
                                                                    (2, 2, 6, 2, 6, 12, -(1 << 9)), (2, 6, 1, 6, 1, 3, -1)):
        a = VecZnx(n, cols, a_size).fill_uniform(base2k + 2, rng)
        b = VecZnx(n, cols, b_size).fill_uniform(base2k + 2, rng)
        ar, br = ref.cnv_pvec_left_alloc(cols, pa_size), ref.cnv_pvec_right_alloc(cols, pb_size)
        ah, bh = hip.cnv_pvec_left_alloc(cols, pa_size), hip.cnv_pvec_right_alloc(cols, pb_size)
        ah.data[...] = 1e9
        bh.data[...] = -1e9
        for (m_, x, y) in ((ref, ar, br), (hip, ah, bh)):
            m_.cnv_prepare_left(x, a, mask)
            m_.cnv_prepare_right(y, b, mask)
        for off in (0, 1, pa_size + pb_size - 2, pa_size + pb_size + 5):
            for (i, j) in ((0, 0), (cols - 1, 0), (0, cols - 1)):
                dr = VecZnxDft(n, 1, res_size)
                dh = VecZnxDft(n, 2, res_size)
                dh.data[...] = 42.0
                ref.cnv_apply_dft(off, dr, 0, ar, i, br, j)
                hip.cnv_apply_dft(off, dh, 1, ah, i, bh, j)
                assert (dh.data[:, 0] == 42.0).all()
                one = VecZnxDft(n, 1, res_size, np.ascontiguousarray(dh.data[:, 1:2]))
                assert np.array_equal(_finish(hip, one, base2k, res_size), _finish(ref, dr, base2k, res_size)), (cols, a_size, off, i, j)
        if pa_size == pb_size:
            sl, sr = hip.cnv_pvec_left_alloc(cols, pa_size), hip.cnv_pvec_right_alloc(cols, pa_size)
            hip.cnv_prepare_self(sl, sr, a, mask)
            assert np.array_equal(sl.data, ah.data) and np.array_equal(sr.data, ah.data)
    assert hip.cnv_apply_dft_tmp_bytes(0, 30, 15, 15) == ref.cnv_apply_dft_tmp_bytes(0, 30, 15, 15)
    assert hip.cnv_pairwise_apply_dft_tmp_bytes(0, 30, 15, 15) == ref.cnv_pairwise_apply_dft_tmp_bytes(0, 30, 15, 15)
    assert hip.cnv_by_const_apply_tmp_bytes(0, 30, 15, 15) == ref.cnv_by_const_apply_tmp_bytes(0, 30, 15, 15)
    assert hip.cnv_prepare_left_tmp_bytes(5, 9) == ref.cnv_prepare_left_tmp_bytes(5, 9)


@pytest.mark.parametrize("n", [8, 16, 64, 2048])
def test_convolution_by_const(mods, n):
    """test_suite/convolution.rs:22-89 (i64 domain, wrapping)."""
    ref, hip = mods(n)
    rng = seeded(n + 1)
    a_size, b_size = 15, 15
    res_size = a_size + b_size
    a = VecZnx(n, 2, a_size).fill_uniform(17, rng)
    for b_const in (rng.integers(-(1 << 16), 1 << 16, b_size, dtype=np.int64), rng.integers(-(1 << 62), 1 << 62, 3, dtype=np.int64)):
        for a_col in range(2):
            for off in (0, 1, 5, a_size, res_size - 1, res_size + 2):
                wr, wh = VecZnxBig(n, 2, res_size), VecZnxBig(n, 2, res_size)
                wr.data[...] = 7
                wh.data[...] = 7
                ref.cnv_by_const_apply(off, wr, 1, a, a_col, b_const)
                hip.cnv_by_const_apply(off, wh, 1, a, a_col, b_const)
                assert np.array_equal(wh.data, wr.data), (a_col, off)


def _run_tensor(hip, ref, n, rank, a_size, b_size, res_size, ab_base2k, res_base2k, cnv_offset, mode, batch, seed, chunk=0, a_bits_off=0):
    from poulpy_amd.hal import GlweTensorParams
    rng = seeded(seed)
    cols = rank + 1
    tcols = cols * (cols + 1) // 2
    square = mode == "square"
    a_k = ab_base2k * a_size - a_bits_off
    b_k = a_k if square else ab_base2k * b_size
    a_all = np.empty((batch, a_size, cols, n), dtype=np.int64)
    b_all = np.empty((batch, (a_size if square else b_size), cols, n), dtype=np.int64)
    prev = rng.integers(-(1 << (res_base2k - 1)), 1 << (res_base2k - 1), (batch, res_size, tcols, n), dtype=np.int64)
    want = np.empty_like(prev)
    for t in range(batch):
        a = VecZnx(n, cols, a_size).fill_uniform(ab_base2k, rng)
        b = a if square else VecZnx(n, cols, b_size).fill_uniform(ab_base2k, rng)
        a_all[t], b_all[t] = a.data, b.data
        r = VecZnx(n, tcols, res_size, prev[t].copy())
        if square:
            ref.glwe_tensor_square_apply(cnv_offset, r, res_base2k, a, a_k, ab_base2k)
        else:
            ref.glwe_tensor_apply(cnv_offset, r, res_base2k, a, a_k, b, b_k, ab_base2k, add_assign=(mode == "add_assign"))
        want[t] = r.data
    d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
    d_b = hip.device_alloc(b_all.nbytes).upload(b_all)
    d_r = hip.device_alloc(prev.nbytes).upload(prev)
    p = GlweTensorParams(rank=rank, a_size=a_size, b_size=b_size, ab_base2k=ab_base2k, a_effective_k=a_k, b_effective_k=b_k,
                         res_size=res_size, res_base2k=res_base2k, cnv_offset=cnv_offset)
    hip.set_chunk(chunk)
    hip.glwe_tensor_apply_batched(d_r.ptr, d_a.ptr, None if square else d_b.ptr, p, mode, batch)
    hip.sync()
    hip.set_chunk(0)
    got = d_r.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_a, d_b, d_r):
        buf.free()
    return got, want


@pytest.mark.parametrize("n", [8, 256])
@pytest.mark.parametrize("mode", ["apply", "add_assign", "square"])
@pytest.mark.parametrize("rank", [1, 2])
def test_glwe_tensor_apply_batched(mods, rank, mode, n):
    """poulpy-core operations/glwe.rs:609-913 on a batch: offsets below / at / above base2k (negative and positive cnv_offset_lo),
    masked bottom limb, different output base, chunked."""
    ref, hip = mods(n)
    for (a_size, b_size, res_size, ab, rb, off, abo) in ((4, 3, 5, 12, 12, 5, 3), (3, 3, 4, 14, 14, 14, 0), (4, 4, 6, 12, 15, 30, 7),
                                                         (2, 5, 3, 13, 11, 20, 0)):
        got, want = _run_tensor(hip, ref, n, rank, a_size, b_size, res_size, ab, rb, off, mode, batch=5, seed=rank * 100 + off, chunk=2,
                                a_bits_off=abo)
        assert np.array_equal(got, want), (rank, mode, a_size, b_size, res_size, ab, rb, off)


@pytest.mark.parametrize("mode", ["apply", "add_assign", "square"])
@pytest.mark.parametrize("rank", [1, 2])
def test_glwe_tensor_apply_fused_row_pass(mods, rank, mode):
    """N = 8192 (m = 32 x 128: the row-major pipeline layout): one base2k and rank <= 2 run the tensoring as pass 1 + k_mid_cnv (forward
    row transform + limb convolution + inverse row transform in one kernel) + raw inverse column pass + normalize with the Karatsuba
    combination in its stores (round 3).  Ragged limb counts, offsets below / at / above base2k, a masked bottom limb, chunks, a result
    with more limbs than the product reaches; a two-base shape on the same ring takes the per-op composition.  Bit-exact vs the oracle."""
    n = 8192
    ref, hip = mods(n)
    for (a_size, b_size, res_size, ab, rb, off, abo) in ((4, 3, 5, 12, 12, 5, 3), (3, 3, 4, 14, 14, 14, 0), (4, 4, 6, 12, 12, 30, 7),
                                                         (2, 5, 3, 13, 13, 20, 0), (1, 1, 2, 12, 12, 0, 0), (5, 2, 9, 12, 12, 13, 0)):
        hip.dispatch_notes(reset=True)
        got, want = _run_tensor(hip, ref, n, rank, a_size, b_size, res_size, ab, rb, off, mode, batch=3, seed=rank * 1000 + off + a_size, chunk=2,
                                a_bits_off=abo)
        assert np.array_equal(got, want), (rank, mode, a_size, b_size, res_size, ab, rb, off)
        if os.environ.get("POULPY_DBG_TENSOR_FUSED") != "0" and os.environ.get("POULPY_DBG_TENSOR_COMBINE") != "0":
            assert "k_mid_cnv" in hip.dispatch_notes(), (rank, mode, a_size, b_size, res_size, off)
    hip.dispatch_notes(reset=True)
    got, want = _run_tensor(hip, ref, n, rank, 4, 4, 6, 12, 15, 30, mode, batch=3, seed=77 + rank, chunk=2, a_bits_off=7)   # two bases
    assert np.array_equal(got, want)
    assert "k_mid_cnv" not in hip.dispatch_notes()


def test_glwe_tensor_apply_n65536_16_limbs(mods):
    """BASELINE configs[4] shape: N = 2^16, 16 limbs, rank 1 (CKKS multiplication, base2k 12 so that FFT64 represents it)."""
    n = 65536
    ref, hip = mods(n)
    got, want = _run_tensor(hip, ref, n, 1, 16, 16, 16, 12, 12, 16 * 12 - 20, "apply", batch=2, seed=65536)
    assert np.array_equal(got, want)


def _run_relinearize(hip, ref, n, rank, a_size, a_base2k, key_size, key_base2k, dnum, dsize, res_size, res_base2k, batch, seed, fuse=(True, True),
                     chunk=0, pin=False):
    from poulpy_amd.hal import GlweOpParams
    rng = seeded(seed)
    cols, pairs = rank + 1, rank * (rank + 1) // 2
    mat = MatZnx(n, dnum, pairs, cols, key_size).fill_uniform(key_base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, pairs, cols, key_size), hip.vmp_pmat_alloc(dnum, pairs, cols, key_size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    a_all = np.empty((batch, a_size, cols + pairs, n), dtype=np.int64)
    want = np.empty((batch, res_size, cols, n), dtype=np.int64)
    for t in range(batch):
        a = VecZnx(n, cols + pairs, a_size).fill_uniform(a_base2k, rng)
        a_all[t] = a.data
        r = VecZnx(n, cols, res_size)
        ref.glwe_tensor_relinearize(r, res_base2k, a, a_base2k, pr, dsize, key_base2k)
        want[t] = r.data
    d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    d_r = hip.device_alloc(want.nbytes)
    hip.lib.pz_memset_d(hip.handle, d_r.ptr, 0x5A, want.nbytes)
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=dsize, key_size=key_size, key_base2k=key_base2k, a_size=a_size, a_base2k=a_base2k,
                     res_size=res_size, res_base2k=res_base2k, rank_out=rank)
    hip.set_chunk(chunk)
    hip.set_fusion(*fuse)
    if pin:
        hip.pin_key(d_k.ptr, dnum, pairs, cols, key_size)
    hip.glwe_tensor_relinearize_batched(d_r.ptr, d_a.ptr, d_k.ptr, p, batch)
    hip.sync()
    if pin:
        hip.unpin_key(d_k.ptr)
    hip.set_chunk(0)
    hip.set_fusion(True, True)
    got = d_r.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_a, d_k, d_r):
        buf.free()
    return got, want


@pytest.mark.parametrize("fuse", [(True, True), (False, False)], ids=["fused", "unfused"])
@pytest.mark.parametrize("rank", [1, 2])
def test_glwe_tensor_relinearize_batched(mods, rank, fuse):
    """operations/glwe.rs:541-607: equal bases (fused pipeline), dsize 2, every base different, and the reference's
    res_base2k == key_base2k != a_base2k case (the un-normalized a is added, :588-592)."""
    for n in (256, 4096):
        ref, hip = mods(n)
        for (a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b) in ((4, 12, 5, 12, 4, 1, 4, 12), (5, 12, 4, 12, 5, 1, 3, 12),
                                                                         (4, 13, 5, 13, 2, 2, 4, 13), (4, 15, 5, 12, 5, 1, 4, 13),
                                                                         (3, 16, 5, 12, 4, 1, 4, 12)):
            got, want = _run_relinearize(hip, ref, n, rank, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b, batch=5,
                                         seed=n + rank + a_size + a_b, fuse=fuse, chunk=3)
            assert np.array_equal(got, want), (n, rank, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b)


def test_config5_relinearize_n65536_16_limbs(mods):
    """BASELINE configs[4], relinearize half: N = 2^16, 16 limbs, rank 1 (tensor of 3 columns, tensor key 1 -> 1 of 16 rows x 16 limbs);
    pinned and unpinned key."""
    n = 65536
    ref, hip = mods(n)
    for pin in (False, True):
        got, want = _run_relinearize(hip, ref, n, 1, 16, 12, 16, 12, 16, 1, 16, 12, batch=3, seed=5 + pin, pin=pin)
        assert np.array_equal(got, want), pin


def _run_mul_relinearize(hip, ref, n, rank, a_size, b_size, t_size, base2k, cnv_offset, key_size, key_base2k, dnum, dsize, res_size, res_base2k, mode, batch,
                         seed, chunk=0, a_bits_off=0, pin=False):
    """glwe_tensor_apply / _square_apply into a scratch tensor (t_size limbs, base2k) + glwe_tensor_relinearize: poulpy-ckks's
    ckks_mul_into_default / square (leveled/default/mul.rs:49-85, :131-170), oracle = the two reference operations one after the other."""
    from poulpy_amd.hal import GlweOpParams, GlweTensorParams
    rng = seeded(seed)
    cols, pairs = rank + 1, rank * (rank + 1) // 2
    tcols = cols + pairs
    square = mode == "square"
    a_k = base2k * a_size - a_bits_off
    b_k = a_k if square else base2k * b_size
    mat = MatZnx(n, dnum, pairs, cols, key_size).fill_uniform(key_base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, pairs, cols, key_size), hip.vmp_pmat_alloc(dnum, pairs, cols, key_size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    a_all = np.empty((batch, a_size, cols, n), dtype=np.int64)
    b_all = np.empty((batch, (a_size if square else b_size), cols, n), dtype=np.int64)
    want = np.empty((batch, res_size, cols, n), dtype=np.int64)
    for t in range(batch):
        a = VecZnx(n, cols, a_size).fill_uniform(base2k, rng)
        b = a if square else VecZnx(n, cols, b_size).fill_uniform(base2k, rng)
        a_all[t], b_all[t] = a.data, b.data
        tmp = VecZnx(n, tcols, t_size)
        if square:
            ref.glwe_tensor_square_apply(cnv_offset, tmp, base2k, a, a_k, base2k)
        else:
            ref.glwe_tensor_apply(cnv_offset, tmp, base2k, a, a_k, b, b_k, base2k)
        r = VecZnx(n, cols, res_size)
        ref.glwe_tensor_relinearize(r, res_base2k, tmp, base2k, pr, dsize, key_base2k)
        want[t] = r.data
    d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
    d_b = hip.device_alloc(b_all.nbytes).upload(b_all)
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    d_r = hip.device_alloc(want.nbytes)
    hip.lib.pz_memset_d(hip.handle, d_r.ptr, 0x5A, want.nbytes)
    tp = GlweTensorParams(rank=rank, a_size=a_size, b_size=b_size, ab_base2k=base2k, a_effective_k=a_k, b_effective_k=b_k, res_size=t_size,
                          res_base2k=base2k, cnv_offset=cnv_offset)
    rp = GlweOpParams(rank=rank, dnum=dnum, dsize=dsize, key_size=key_size, key_base2k=key_base2k, a_size=t_size, a_base2k=base2k,
                      res_size=res_size, res_base2k=res_base2k, rank_out=rank)
    hip.set_chunk(chunk)
    if pin:
        hip.pin_key(d_k.ptr, dnum, pairs, cols, key_size)
    hip.glwe_tensor_mul_relinearize_batched(d_r.ptr, d_a.ptr, None if square else d_b.ptr, d_k.ptr, tp, rp, mode, batch)
    hip.sync()
    if pin:
        hip.unpin_key(d_k.ptr)
    hip.set_chunk(0)
    got = d_r.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_a, d_b, d_k, d_r):
        buf.free()
    return got, want


@pytest.mark.parametrize("mode", ["apply", "square"])
@pytest.mark.parametrize("rank", [1, 2])
def test_glwe_tensor_mul_relinearize_batched(mods, rank, mode):
    """pz_glwe_tensor_mul_relinearize_batched == glwe_tensor_apply / _square_apply then glwe_tensor_relinearize of the oracle, bit for bit.
    N = 8192 (pipeline plan): one base2k <= 14 keeps the tensor as 16-bit digits in the workspace (tensoring tails NZF 5 / 6, forward pass 1 on
    the tile-order copies, the relinearization's tail adding them: k_fwd_pass1_t16 in the dispatch notes' place is checked through the result
    alone) - ragged limb counts, offsets below / at / above base2k, a masked bottom limb, chunks, base2k 14 (the widest the pairwise column's
    pair - d_i - d_j fits); base2k 15, a key in another base, dsize 2 and the small rings take the i64 tensor in the workspace."""
    for n in (8192, 256):
        ref, hip = mods(n)
        for (a_size, b_size, t_size, k, off, key_size, key_k, dnum, dsize, res_size, res_k, abo) in (
                (4, 3, 5, 12, 5, 5, 12, 5, 1, 4, 12, 3), (3, 3, 4, 14, 14, 4, 14, 4, 1, 4, 14, 0), (4, 4, 6, 12, 30, 6, 12, 6, 1, 5, 12, 7),
                (2, 5, 3, 13, 20, 4, 13, 3, 1, 3, 13, 0), (3, 3, 4, 15, 14, 4, 15, 4, 1, 4, 15, 0), (4, 4, 4, 12, 16, 5, 13, 5, 1, 4, 12, 0),
                (4, 4, 4, 13, 16, 5, 13, 2, 2, 4, 13, 0)):
            if n == 256 and a_size == 2:
                continue
            got, want = _run_mul_relinearize(hip, ref, n, rank, a_size, b_size, t_size, k, off, key_size, key_k, dnum, dsize, res_size, res_k, mode,
                                             batch=5, seed=n + 31 * rank + off + k, chunk=2, a_bits_off=abo)
            assert np.array_equal(got, want), (n, rank, mode, a_size, b_size, t_size, k, off, key_size, key_k, dnum, dsize, res_size, res_k)


def test_config5_mul_relinearize_n65536_16_limbs(mods):
    """BASELINE configs[4] as ONE call: N = 2^16, 16 limbs, rank 1, base2k 12 - tensoring + relinearization with the tensor as 16-bit digits;
    apply and square, pinned and unpinned tensor key."""
    n = 65536
    ref, hip = mods(n)
    for mode, pin in (("apply", False), ("square", True)):
        got, want = _run_mul_relinearize(hip, ref, n, 1, 16, 16, 16, 12, 16 * 12 - 20, 16, 12, 16, 1, 16, 12, mode, batch=3, seed=6 + pin, pin=pin)
        assert np.array_equal(got, want), (mode, pin)


# ------------------------------------------------------------------------------------------
# batched i64 VecZnx family (pz_vec_znx_*_batched): one launch for `batch` device-resident containers
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [64, 4096, 65536])
def test_vec_znx_family_batched(mods, n):
    import ctypes as C
    ref, hip = mods(n)
    rng = seeded(n + 9)
    batch = 5 if n < 65536 else 3
    sz = lambda *xs: [C.c_size_t(int(x)) for x in xs]
    for (rc, rs, ac, as_, bc, bs) in ((2, 3, 2, 3, 2, 3), (3, 4, 2, 2, 1, 5), (1, 2, 2, 4, 3, 3)):
        a = rng.integers(-(1 << 40), 1 << 40, (batch, as_, ac, n), dtype=np.int64)
        b = rng.integers(-(1 << 40), 1 << 40, (batch, bs, bc, n), dtype=np.int64)
        r0 = rng.integers(-(1 << 40), 1 << 40, (batch, rs, rc, n), dtype=np.int64)
        d_a, d_b = hip.device_alloc(a.nbytes).upload(a), hip.device_alloc(b.nbytes).upload(b)
        d_r = hip.device_alloc(r0.nbytes)
        rcol, acol, bcol = rc - 1, ac - 1, 0

        def oracle(fn):
            out = r0.copy()
            for t in range(batch):
                fn(VecZnx(n, rc, rs, out[t]), VecZnx(n, ac, as_, a[t].copy()), VecZnx(n, bc, bs, b[t].copy()))
            return out

        def device(call):
            d_r.upload(r0)
            hip._ck(call())
            hip.sync()
            return d_r.download(np.int64, r0.size).reshape(r0.shape)

        L = hip.lib
        cases = {
            "add_into": (lambda r, x, y: ref.vec_znx_add_into(r, rcol, x, acol, y, bcol),
                         lambda: L.pz_vec_znx_add_into_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol), d_b.ptr, *sz(bc, bs, bcol))),
            "sub": (lambda r, x, y: ref.vec_znx_sub(r, rcol, x, acol, y, bcol),
                    lambda: L.pz_vec_znx_sub_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol), d_b.ptr, *sz(bc, bs, bcol))),
            "add_assign": (lambda r, x, y: ref.vec_znx_add_assign(r, rcol, x, acol),
                           lambda: L.pz_vec_znx_add_assign_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
            "sub_assign": (lambda r, x, y: ref.vec_znx_sub_assign(r, rcol, x, acol),
                           lambda: L.pz_vec_znx_sub_assign_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
            "sub_negate_assign": (lambda r, x, y: ref.vec_znx_sub_negate_assign(r, rcol, x, acol),
                                  lambda: L.pz_vec_znx_sub_negate_assign_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
            "negate": (lambda r, x, y: ref.vec_znx_negate(r, rcol, x, acol),
                       lambda: L.pz_vec_znx_negate_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
            "copy": (lambda r, x, y: ref.vec_znx_copy(r, rcol, x, acol),
                     lambda: L.pz_vec_znx_copy_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
            "rotate": (lambda r, x, y: ref.vec_znx_rotate(-7 - n, r, rcol, x, acol),
                       lambda: L.pz_vec_znx_rotate_batched(hip.handle, C.c_size_t(batch), C.c_int64(-7 - n), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
            "normalize": (lambda r, x, y: ref.vec_znx_normalize(r, 13, -5, rcol, x, 17, acol),
                          lambda: L.pz_vec_znx_normalize_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, 13), C.c_int64(-5), C.c_size_t(rcol), d_a.ptr, *sz(ac, as_, 17, acol))),
            "lsh": (lambda r, x, y: ref.vec_znx_lsh(14, 19, r, rcol, x, acol),
                    lambda: L.pz_vec_znx_lsh_batched(hip.handle, C.c_size_t(batch), *sz(14, 19), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
            "rsh": (lambda r, x, y: ref.vec_znx_rsh(14, 23, r, rcol, x, acol),
                    lambda: L.pz_vec_znx_rsh_batched(hip.handle, C.c_size_t(batch), *sz(14, 23), d_r.ptr, *sz(rc, rs, rcol), d_a.ptr, *sz(ac, as_, acol))),
        }
        for name, (of, df) in cases.items():
            assert np.array_equal(device(df), oracle(of)), (n, name, rc, rs, ac, as_)
        want = r0.copy()
        want[:, :, rcol] = 0
        assert np.array_equal(device(lambda: L.pz_vec_znx_zero_batched(hip.handle, C.c_size_t(batch), d_r.ptr, *sz(rc, rs, rcol))), want)
        for buf in (d_a, d_b, d_r):
            buf.free()
