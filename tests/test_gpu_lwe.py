"""GPU parity of the LWE glue of the gate bootstrap (BASELINE configs[3]: mod_switch_2n -> blind rotation -> LWE key switch) against
the CPU oracle, through the C ABI, bit-exact.  Shapes follow poulpy-core/src/test_suite/keyswitch/lwe_ct.rs and conversion.rs
(LWE dimension below the ring degree, rank-1 / rank-2 keys) and poulpy-bin-fhe's blind-rotation tests (block sizes, bases)."""
import numpy as np
import pytest

from poulpy_amd.layouts import MatZnx, VecZnx
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


def rand_lwe(rng, batch, size, n_lwe, base2k):
    return rng.integers(-(1 << (base2k - 1)), 1 << (base2k - 1), size=(batch, size, n_lwe + 1), dtype=np.int64)


def prepared(ref, hip, rng, n, dnum, cols_in, cols_out, size, base2k):
    mat = MatZnx(n, dnum, cols_in, cols_out, size).fill_uniform(base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, cols_in, cols_out, size), hip.vmp_pmat_alloc(dnum, cols_in, cols_out, size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    return pr, ph


@pytest.mark.parametrize("n2,base2k,size", [(2048, 17, 2), (2048, 12, 2), (2048, 5, 4), (1 << 15, 13, 3), (64, 7, 1), (4096, 13, 1), (1024, 11, 2)])
@pytest.mark.parametrize("negate", [False, True])
def test_mod_switch_2n(mods, n2, base2k, size, negate):
    """both branches of mod.rs:136-171 (one limb rounded / limbs concatenated, incl. the partial last limb and base2k == log2n),
    Left and Right, a ragged batch"""
    ref, hip = mods(256)
    rng = seeded(n2 + base2k + size)
    batch, n_lwe = 37, 101
    lwe = rand_lwe(rng, batch, size, n_lwe, base2k)
    want = np.stack([ref.mod_switch_2n(n2, lwe[b], base2k, negate) for b in range(batch)])
    d_l = hip.device_alloc(lwe.nbytes).upload(lwe)
    d_r = hip.device_alloc(want.nbytes)
    hip.lwe_mod_switch_2n_batched(d_r.ptr, d_l.ptr, n_lwe, size, base2k, n2, negate, batch)
    hip.sync()
    got = d_r.download(np.int64, want.size).reshape(want.shape)
    d_l.free(); d_r.free()
    assert np.array_equal(got, want)


def test_mod_switch_2n_rejects_short_lwes(mods):
    from poulpy_amd.hal import PoulpyHipError
    _, hip = mods(256)
    d = hip.device_alloc(1 << 16)
    with pytest.raises(PoulpyHipError):
        hip.lwe_mod_switch_2n_batched(d.ptr, d.ptr, 10, 1, 5, 2048, False, 1)   # 12 bits wanted, one limb of 5
    d.free()


@pytest.mark.parametrize("n,n_lwe", [(256, 100), (1024, 1024), (4096, 77)])
def test_sample_extract(mods, n, n_lwe):
    ref, hip = mods(n)
    rng = seeded(n + n_lwe)
    for cols, a_size, res_size, batch in ((2, 3, 3, 5), (2, 4, 2, 3), (3, 2, 4, 2)):
        a = rng.integers(-2048, 2048, (batch, a_size, cols, n), dtype=np.int64)
        want = np.stack([ref.lwe_sample_extract(n_lwe, res_size, VecZnx(n, cols, a_size, a[b].copy())) for b in range(batch)])
        d_a = hip.device_alloc(a.nbytes).upload(a)
        d_r = hip.device_alloc(want.nbytes)
        hip.lib.pz_memset_d(hip.handle, d_r.ptr, 0x33, want.nbytes)
        hip.lwe_sample_extract_batched(d_r.ptr, n_lwe, res_size, d_a.ptr, cols, a_size, batch)
        hip.sync()
        got = d_r.download(np.int64, want.size).reshape(want.shape)
        d_a.free(); d_r.free()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("n,n_in,n_out,a_size,res_size,a_b,key_b,res_b,dsize", [
    (256, 100, 77, 3, 3, 12, 12, 12, 1),        # five-kernel path (N = 256)
    (4096, 600, 500, 3, 2, 13, 13, 13, 1),      # fused pipeline
    (4096, 4096, 17, 4, 4, 12, 12, 12, 2),      # dsize 2
    (8192, 300, 300, 2, 3, 17, 12, 15, 1),      # three different bases
])
def test_lwe_keyswitch(mods, n, n_in, n_out, a_size, res_size, a_b, key_b, res_b, dsize):
    """keyswitching/lwe.rs:49-94 (the shapes of poulpy-core/src/test_suite/keyswitch/lwe_ct.rs: LWE dimensions below the ring degree)"""
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(n + n_in + n_out)
    batch = 5
    key_size = -(-a_size * a_b // key_b) + 1
    dnum = -(-(key_size - 1) // dsize)
    pr, ph = prepared(ref, hip, rng, n, dnum, 1, 2, key_size, key_b)
    lwe = rand_lwe(rng, batch, a_size, n_in, a_b)
    want = np.stack([ref.lwe_keyswitch(n_out, res_size, res_b, lwe[b], a_b, pr, dsize, key_b) for b in range(batch)])
    p = GlweOpParams(rank=1, dnum=dnum, dsize=dsize, key_size=key_size, key_base2k=key_b, a_size=a_size, a_base2k=a_b, res_size=res_size,
                     res_base2k=res_b, rank_out=1)
    d_l = hip.device_alloc(lwe.nbytes).upload(lwe)
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    d_r = hip.device_alloc(want.nbytes)
    hip.lib.pz_memset_d(hip.handle, d_r.ptr, 0x33, want.nbytes)
    hip.lwe_keyswitch_batched(d_r.ptr, n_out, d_l.ptr, n_in, d_k.ptr, p, batch)
    hip.sync()
    got = d_r.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_l, d_k, d_r):
        buf.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,n_lwe,lwe_size,lwe_b,key_b,rank_out", [(256, 100, 3, 12, 12, 1), (4096, 512, 2, 13, 13, 2), (4096, 700, 2, 17, 12, 1),
                                                                  (2048, 33, 3, 11, 14, 2)])
def test_glwe_from_lwe(mods, n, n_lwe, lwe_size, lwe_b, key_b, rank_out):
    """conversion/lwe_to_glwe.rs:46-121, same base and cross-base embedding, rank-1 and rank-2 outputs"""
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(n + n_lwe + rank_out)
    batch = 4
    glwe_size = -(-lwe_size * lwe_b // key_b)
    key_size, dnum, res_size = glwe_size + 1, glwe_size, glwe_size
    pr, ph = prepared(ref, hip, rng, n, dnum, 1, rank_out + 1, key_size, key_b)
    lwe = rand_lwe(rng, batch, lwe_size, n_lwe, lwe_b)
    want = np.empty((batch, res_size, rank_out + 1, n), dtype=np.int64)
    for b in range(batch):
        r = VecZnx(n, rank_out + 1, res_size)
        ref.glwe_from_lwe(r, key_b, lwe[b], lwe_b, glwe_size, pr, 1, key_b)
        want[b] = r.data
    p = GlweOpParams(rank=1, dnum=dnum, dsize=1, key_size=key_size, key_base2k=key_b, a_size=glwe_size, a_base2k=key_b, res_size=res_size,
                     res_base2k=key_b, rank_out=rank_out)
    d_l = hip.device_alloc(lwe.nbytes).upload(lwe)
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    d_r = hip.device_alloc(want.nbytes)
    hip.glwe_from_lwe_batched(d_r.ptr, d_l.ptr, n_lwe, lwe_size, lwe_b, d_k.ptr, p, batch)
    hip.sync()
    got = d_r.download(np.int64, want.size).reshape(want.shape)
    for buf in (d_l, d_k, d_r):
        buf.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,rank,a_idx,n_lwe", [(256, 1, 0, 100), (4096, 1, 5, 640), (4096, 2, 4095, 33), (2048, 3, 1, 2048)])
def test_lwe_from_glwe(mods, n, rank, a_idx, n_lwe):
    """conversion/glwe_to_lwe.rs:42-90: coefficient a_idx of a rank-`rank` GLWE to an LWE"""
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rng = seeded(n + rank + a_idx)
    batch, size, base2k = 3, 3, 12
    key_size, dnum = size + 1, size
    pr, ph = prepared(ref, hip, rng, n, dnum, rank, 2, key_size, base2k)
    a = rng.integers(-2048, 2048, (batch, size, rank + 1, n), dtype=np.int64)
    want = np.stack([ref.lwe_from_glwe(n_lwe, size, base2k, VecZnx(n, rank + 1, size, a[b].copy()), base2k, a_idx, pr, 1, base2k)
                     for b in range(batch)])
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=key_size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=1)
    d_a = hip.device_alloc(a.nbytes).upload(a)
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    d_r = hip.device_alloc(want.nbytes)
    hip.lwe_from_glwe_batched(d_r.ptr, n_lwe, d_a.ptr, a_idx, d_k.ptr, p, batch)
    hip.sync()
    got = d_r.download(np.int64, want.size).reshape(want.shape)
    a_after = d_a.download(np.int64, a.size).reshape(a.shape)
    for buf in (d_a, d_k, d_r):
        buf.free()
    assert np.array_equal(got, want) and np.array_equal(a_after, a)


@pytest.mark.parametrize("block_size", [7, 1])
def test_gate_bootstrap_chain(mods, block_size):
    """BASELINE configs[3], every step on the device: LWE (2 limbs of 17 bits) -> mod_switch_2n -> CGGI blind rotation at N = 1024
    (rank 1, dnum 2) -> sample extract + LWE key switch back to the short LWE dimension (lwe_from_glwe at coefficient 0), compared
    step by step and end to end with the oracle on the same key material (short LWE: 28 coefficients, so that the CPU finishes)."""
    from poulpy_amd.hal import BlindRotationParams, GlweOpParams
    n, k, n_lwe, rank, dnum, bsz, rsz, batch = 1024, 17, 28, 1, 2, 2, 2, 6
    ref, hip = mods(n)
    rng = seeded(4242 + block_size)
    cols = rank + 1
    lut = VecZnx(n, 1, rsz).fill_uniform(k, rng)
    brk_r = np.empty((n_lwe, n * dnum * cols * cols * bsz), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        pr, ph = prepared(ref, hip, rng, n, dnum, cols, cols, bsz, k)
        brk_r[i], brk_h[i] = pr.data.reshape(-1), ph.data.reshape(-1)
    ksk_r, ksk_h = prepared(ref, hip, rng, n, rsz, rank, 2, rsz + 1, k)
    lwe = rand_lwe(rng, batch, 2, n_lwe, k)
    # oracle chain
    xpa = ref.blind_rotation_x_pow_a() if block_size > 1 else np.zeros((1, 1))
    want_2n = np.stack([ref.mod_switch_2n(2 * n, lwe[b], k, False) for b in range(batch)])
    want_acc = np.empty((batch, rsz, cols, n), dtype=np.int64)
    want_out = np.empty((batch, rsz, n_lwe + 1), dtype=np.int64)
    for b in range(batch):
        acc = VecZnx(n, cols, rsz)
        ref.blind_rotation_execute(acc, k, np.ascontiguousarray(want_2n[b]), lut, brk_r, dnum, bsz, block_size, xpa)
        want_acc[b] = acc.data
        want_out[b] = ref.lwe_from_glwe(n_lwe, rsz, k, acc, k, 0, ksk_r, 1, k)
    # device chain
    d_lwe = hip.device_alloc(lwe.nbytes).upload(lwe)
    d_2n = hip.device_alloc(want_2n.nbytes)
    d_lut = hip.device_alloc(lut.data.nbytes).upload(lut.data)
    d_brk = hip.device_alloc(brk_h.nbytes).upload(brk_h)
    d_ksk = hip.device_alloc(ksk_h.data.nbytes).upload(ksk_h.data)
    d_acc = hip.device_alloc(want_acc.nbytes)
    d_out = hip.device_alloc(want_out.nbytes)
    hip.lwe_mod_switch_2n_batched(d_2n.ptr, d_lwe.ptr, n_lwe, 2, k, 2 * n, False, batch)
    bp = BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block_size, dnum=dnum, brk_size=bsz, base2k=k, res_size=rsz, lut_size=rsz)
    hip.blind_rotation_execute_batched(d_acc.ptr, d_2n.ptr, d_lut.ptr, d_brk.ptr, bp, batch)
    kp = GlweOpParams(rank=rank, dnum=rsz, dsize=1, key_size=rsz + 1, key_base2k=k, a_size=rsz, a_base2k=k, res_size=rsz, res_base2k=k, rank_out=1)
    hip.lwe_from_glwe_batched(d_out.ptr, n_lwe, d_acc.ptr, 0, d_ksk.ptr, kp, batch)
    hip.sync()
    got_2n = d_2n.download(np.int64, want_2n.size).reshape(want_2n.shape)
    got_acc = d_acc.download(np.int64, want_acc.size).reshape(want_acc.shape)
    got_out = d_out.download(np.int64, want_out.size).reshape(want_out.shape)
    for buf in (d_lwe, d_2n, d_lut, d_brk, d_ksk, d_acc, d_out):
        buf.free()
    assert np.array_equal(got_2n, want_2n)
    assert np.array_equal(got_acc, want_acc)
    assert np.array_equal(got_out, want_out)
