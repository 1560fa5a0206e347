"""GPU parity at the LAUNCH GEOMETRY of bench.py / BASELINE.json.

The small-batch parity tests run the persistent middle kernel with one tile per workgroup; the multi-tile software pipeline
(prefetch of tile t+1 under the inverse pass of t, XCD-aware tile order, per-workgroup row rotation, dummy stores of partial
tiles) only runs at hundreds of ciphertexts per launch.  Here the batch is BASELINE's: a pool of P distinct ciphertexts (P
coprime with every tile size) is replicated on the device to `batch` ciphertexts, the oracle computes the P expected results,
and EVERY output ciphertext of the launch is compared with the expected result of its pool entry — every tile, every XCD slot,
the last partial wave.  Bit-exact.

PyTorch is plumbing only here (large device buffers, gather / compare on the device); the op under test goes through the C ABI.
"""
import ctypes as C

import os

import numpy as np
import pytest

from poulpy_amd.layouts import MatZnx, VecZnx
from tests.helpers import MARGIN_MAX, probed_margin, seeded

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


def _pool_parity(hip, ref, ks, n, rank, size, base2k, dnum, batch, pool, seed, chunk=0, pin=False, in_chunks=64):
    """external product (ks = False) / key switch (ks = True) on `batch` device-resident ciphertexts; returns the number of
    output ciphertexts that differ from the oracle (0 = parity)."""
    import torch
    from poulpy_amd.hal import GlweOpParams
    rng = seeded(seed)
    cols = rank + 1
    cols_in = rank if ks else cols
    mat = MatZnx(n, dnum, cols_in, cols, size).fill_uniform(base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, cols_in, cols, size), hip.vmp_pmat_alloc(dnum, cols_in, cols, size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    a_pool = np.empty((pool, size, cols, n), dtype=np.int64)
    want_pool = np.empty((pool, size, cols, n), dtype=np.int64)
    for i in range(pool):
        a = VecZnx(n, cols, size).fill_uniform(base2k, rng)
        a_pool[i] = a.data
        r = VecZnx(n, cols, size)
        (ref.glwe_keyswitch if ks else ref.glwe_external_product)(r, base2k, a, base2k, pr, 1, base2k)
        want_pool[i] = r.data
    dev = torch.device("cuda", 0)
    d_pool = torch.from_numpy(a_pool).to(dev)
    d_want = torch.from_numpy(want_pool).to(dev)
    idx = torch.arange(batch, device=dev) % pool
    a_all = d_pool[idx].contiguous()                      # ciphertext b = pool entry b mod P
    res = torch.full((batch, size, cols, n), 0x5A5A5A5A, dtype=torch.int64, device=dev)
    key = torch.from_numpy(ph.data).to(dev)
    torch.cuda.synchronize()
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k,
                     res_size=size, res_base2k=base2k, rank_out=rank)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    hip.set_chunk(chunk)
    if pin:
        hip.pin_key(ptr(key), dnum, cols_in, cols, size)
    def count_bad():
        bad = 0
        for b0 in range(0, batch, in_chunks):             # compare on the device, a slab at a time
            b1 = min(batch, b0 + in_chunks)
            eq = (res[b0:b1] == d_want[idx[b0:b1]]).reshape(b1 - b0, -1).all(dim=1)
            bad += int((~eq).sum().item())
        return bad
    run = lambda: (hip.glwe_keyswitch_batched if ks else hip.glwe_external_product_batched)(ptr(res), ptr(a_all), ptr(key), p, batch)
    try:
        run()
        hip.sync()
        bad = count_bad()
        # the same call with the rounding margin measured on the kernels it dispatches: thin margins are reported before they become wrong limbs
        res.fill_(0x5A5A5A5A)
        torch.cuda.synchronize()   # torch's stream and the module's stream are not ordered
        margin = probed_margin(hip, run)
        bad_probe = count_bad()
    finally:
        if pin:
            hip.unpin_key(ptr(key))
        hip.set_chunk(0)
    assert bad_probe == 0 or bad != 0, f"the probing instantiations changed {bad_probe} results (the product kernels were right)"
    bad += bad_probe
    assert margin < MARGIN_MAX, f"rounding margin too thin: max |x - round(x)| = {margin} (N = {n}, base2k {base2k}, {size} limbs)"
    del a_all, res, d_pool, d_want, key
    torch.cuda.empty_cache()
    return bad


@pytest.mark.parametrize("pin,chunk", [(True, 0), (False, 0), (True, 384)], ids=["pinned-one-wave", "unpinned-one-wave", "pinned-chunked"])
def test_metric_config_at_bench_batch(mods, pin, chunk):
    """The metric configuration (N = 2^16, 8 limbs, rank 1, base2k 12, dnum 8) at 1030 ciphertexts per call: k_mid128<4,16> walks
    >= 3 tiles per workgroup (258 ciphertext tiles x 32 rows per XCD over 32 workgroups), the last tile is half empty; pinned and
    unpinned key; one wave and three waves of 384 / 384 / 262."""
    n = 65536
    ref, hip = mods(n)
    assert _pool_parity(hip, ref, False, n, 1, 8, 12, 8, batch=1030, pool=37, seed=1030 + chunk, chunk=chunk, pin=pin) == 0


def test_config3_keyswitch_at_batch_4096(mods):
    """BASELINE configs[2]: GLWE key-switch via VmpPMat, N = 2^16, 8 limbs, batch 4096 (32 GiB in, 32 GiB out; the pipeline splits
    it into waves by itself)."""
    n = 65536
    ref, hip = mods(n)
    assert _pool_parity(hip, ref, True, n, 1, 8, 12, 8, batch=4096, pool=29, seed=4096, pin=True) == 0


def test_config2_external_product_at_batch_1024(mods):
    """BASELINE configs[1]: GGSW external product, N = 2^12, 4 limbs, base2k 17, batch 1024."""
    n = 4096
    ref, hip = mods(n)
    assert _pool_parity(hip, ref, False, n, 1, 4, 17, 4, batch=1024, pool=41, seed=1024) == 0
    assert _pool_parity(hip, ref, False, n, 1, 4, 17, 4, batch=1027, pool=41, seed=1027, pin=True) == 0


@pytest.mark.parametrize("n,size,base2k,ks,one_kernel", [
    (1024, 4, 17, False, True),    # external product, N = 1024: two workgroups per CU, both output columns through the tile at once
    (1024, 4, 17, True, True),     # key switch, N = 1024: 4 input polynomials, body operand on column 0
    (1024, 2, 20, False, True),    # 2 limbs
    (2048, 4, 17, False, True),    # external product, N = 2048: one workgroup per CU, the chain in two rounds
    (2048, 3, 18, False, True),    # 3 limbs (KS = 3)
    (2048, 4, 17, True, False),    # key switch at N = 2048 stays on the two-kernel pipeline (measured - 4 % on one kernel)
])
def test_one_kernel_product_pool_parity_at_bench_batch(mods, n, size, base2k, ks, one_kernel):
    """Round 6, k_small_one (device_small_one.hpp): forward transforms, product, inverse transforms and carry chains of a ciphertext in ONE workgroup at N = 1024 /
    2048 - 1027 ciphertexts per call (ragged against nothing: one workgroup per ciphertext, the last CU round is partial), pinned and unpinned key, every
    output against its pool entry, the margin probe re-run on the same dispatch; the dispatch note names the form."""
    ref, hip = mods(n)
    for pin in (False, True):
        hip.dispatch_notes(reset=True)
        assert _pool_parity(hip, ref, ks, n, 1, size, base2k, size, batch=1027, pool=23, seed=7000 + n + size + int(ks), pin=pin) == 0
        notes = hip.dispatch_notes()
        if os.environ.get("POULPY_DBG_SMALL_ONE", "1") != "0":
            assert ("k_small_one<M1=%d,KS=%d>" % (n // 256, size) in notes) == one_kernel, notes


def test_config5_shape_keyswitch_16_limbs_at_batch_256(mods):
    """BASELINE configs[4] shape (N = 2^16, 16 limbs: 32 output polynomials, k_mid128<2,32>) at 259 ciphertexts per call."""
    n = 65536
    ref, hip = mods(n)
    assert _pool_parity(hip, ref, True, n, 1, 16, 12, 16, batch=259, pool=7, seed=259, pin=True) == 0


def _auto_pool_parity(hip, ref, n, rank, size, base2k, dnum, gal, mode, batch, pool, seed, pin=True, in_place=False, in_chunks=32):
    """the glwe_automorphism family (key switch + X -> X^gal, poulpy-core automorphism/glwe_ct.rs:51-275) on `batch` device-resident
    ciphertexts drawn from a pool of `pool` distinct ones; every output compared with its pool entry's oracle result on the device.
    Returns the number of mismatching ciphertexts."""
    import torch
    from poulpy_amd.hal import GlweOpParams
    rng = seeded(seed)
    cols = rank + 1
    mat = MatZnx(n, dnum, rank, cols, size).fill_uniform(base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, rank, cols, size), hip.vmp_pmat_alloc(dnum, rank, cols, size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    a_pool = np.empty((pool, size, cols, n), dtype=np.int64)
    want_pool = np.empty((pool, size, cols, n), dtype=np.int64)
    for i in range(pool):
        a = VecZnx(n, cols, size).fill_uniform(base2k, rng)
        a_pool[i] = a.data
        r = VecZnx(n, cols, size)
        ref.glwe_automorphism(r, base2k, a, base2k, pr, 1, base2k, gal, mode)
        want_pool[i] = r.data
    dev = torch.device("cuda", 0)
    d_pool = torch.from_numpy(a_pool).to(dev)
    d_want = torch.from_numpy(want_pool).to(dev)
    idx = torch.arange(batch, device=dev) % pool
    a_all = d_pool[idx].contiguous()
    res = a_all if in_place else torch.full((batch, size, cols, n), 0x5A5A5A5A, dtype=torch.int64, device=dev)
    key = torch.from_numpy(ph.data).to(dev)
    torch.cuda.synchronize()
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k,
                     res_size=size, res_base2k=base2k, rank_out=rank)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    def count_bad():
        bad = 0
        for b0 in range(0, batch, in_chunks):
            b1 = min(batch, b0 + in_chunks)
            eq = (res[b0:b1] == d_want[idx[b0:b1]]).reshape(b1 - b0, -1).all(dim=1)
            bad += int((~eq).sum().item())
        return bad
    run = lambda: hip.glwe_automorphism_batched(ptr(res), ptr(a_all), ptr(key), p, gal, mode, batch)
    if pin:
        hip.pin_key(ptr(key), dnum, rank, cols, size)
    try:
        run()
        hip.sync()
        bad = count_bad()
        # once more with the rounding margin measured on the dispatched kernels (fresh inputs: res may alias a)
        a_all.copy_(d_pool[idx])
        if not in_place:
            res.fill_(0x5A5A5A5A)
        torch.cuda.synchronize()   # torch's stream and the module's stream are not ordered
        margin = probed_margin(hip, run)
        bad_probe = count_bad()
    finally:
        if pin:
            hip.unpin_key(ptr(key))
    assert bad_probe == 0 or bad != 0, f"the probing instantiations changed {bad_probe} results (the product kernels were right)"
    bad += bad_probe
    assert margin < MARGIN_MAX, f"rounding margin too thin: max |x - round(x)| = {margin} (N = {n}, base2k {base2k}, {size} limbs, automorphism {mode})"
    del a_all, res, d_pool, d_want, key
    torch.cuda.empty_cache()
    return bad


@pytest.mark.parametrize("gal,mode", [(5, "automorphism"), (-1, "automorphism"), (5, "add"), (-1, "add")])
def test_config5_rotate_16_limbs_pool_parity_at_bench_batch(mods, gal, mode):
    """BASELINE configs[4], rotate half (CKKS rotate = glwe_automorphism, poulpy-ckks leveled/default/rotate.rs:47-50) at its own limb
    count: N = 2^16, 16 limbs, rank 1, 259 ciphertexts per call, pinned key - the 32-slot tile k_mid128r<2,32,PERM=1,..> with the
    permuted spectrum position and the permuting / sign-restoring tail (only 8 limbs were covered).  Galois
    elements 5 (= 1 mod 4) and -1 (= 3 mod 4: conjugated spectrum); the plain form and `_add` (the trace's step), every output checked."""
    n = 65536
    ref, hip = mods(n)
    assert _auto_pool_parity(hip, ref, n, 1, 16, 12, 16, gal, mode, batch=259, pool=5, seed=5000 + gal + len(mode)) == 0


def test_config5_rotate_16_limbs_in_place(mods):
    """glwe_automorphism_assign / _add_assign (res == a, glwe_ct.rs:74-94, :142-183) at the same geometry."""
    n = 65536
    ref, hip = mods(n)
    assert _auto_pool_parity(hip, ref, n, 1, 16, 12, 16, 5, "add", batch=131, pool=5, seed=5131, in_place=True) == 0


def _tensor_pool_parity(hip, ref, n, rank, size, base2k, cnv_offset, mode, batch, pool, seed, in_chunks=16, res_size=None):
    """glwe_tensor_apply / _add_assign / _square (poulpy-core operations/glwe.rs:609-913) on `batch` pairs drawn from a pool; every
    tensor compared with its pool entry's oracle result on the device.  Returns the number of mismatching tensors."""
    import torch
    from poulpy_amd.hal import GlweTensorParams
    rng = seeded(seed)
    cols = rank + 1
    tcols = cols * (cols + 1) // 2
    square = mode == "square"
    k = base2k * size
    res_size = res_size or size
    a_pool = np.empty((pool, size, cols, n), dtype=np.int64)
    b_pool = np.empty((pool, size, cols, n), dtype=np.int64)
    prev_pool = rng.integers(-(1 << (base2k - 1)), 1 << (base2k - 1), (pool, res_size, tcols, n), dtype=np.int64)
    want_pool = np.empty_like(prev_pool)
    for i in range(pool):
        a = VecZnx(n, cols, size).fill_uniform(base2k, rng)
        b = a if square else VecZnx(n, cols, size).fill_uniform(base2k, rng)
        a_pool[i], b_pool[i] = a.data, b.data
        r = VecZnx(n, tcols, res_size, prev_pool[i].copy())
        if square:
            ref.glwe_tensor_square_apply(cnv_offset, r, base2k, a, k, base2k)
        else:
            ref.glwe_tensor_apply(cnv_offset, r, base2k, a, k, b, k, base2k, add_assign=(mode == "add_assign"))
        want_pool[i] = r.data
    dev = torch.device("cuda", 0)
    idx = torch.arange(batch, device=dev) % pool
    d_a = torch.from_numpy(a_pool).to(dev)[idx].contiguous()
    d_b = None if square else torch.from_numpy(b_pool).to(dev)[idx].contiguous()
    d_prev = torch.from_numpy(prev_pool).to(dev)
    d_r = d_prev[idx].contiguous()
    d_want = torch.from_numpy(want_pool).to(dev)
    torch.cuda.synchronize()
    p = GlweTensorParams(rank=rank, a_size=size, b_size=size, ab_base2k=base2k, a_effective_k=k, b_effective_k=k, res_size=res_size,
                         res_base2k=base2k, cnv_offset=cnv_offset)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    def count_bad():
        bad = 0
        for b0 in range(0, batch, in_chunks):
            b1 = min(batch, b0 + in_chunks)
            eq = (d_r[b0:b1] == d_want[idx[b0:b1]]).reshape(b1 - b0, -1).all(dim=1)
            bad += int((~eq).sum().item())
        return bad
    run = lambda: hip.glwe_tensor_apply_batched(ptr(d_r), ptr(d_a), None if square else ptr(d_b), p, mode, batch)
    hip.dispatch_notes(reset=True)
    run()
    hip.sync()
    notes = hip.dispatch_notes()
    bad = count_bad()
    # once more with the rounding margin measured on the tensoring tails (the previous content of res again: add_assign accumulates)
    d_r.copy_(d_prev[idx])
    torch.cuda.synchronize()   # torch's stream and the module's stream are not ordered
    margin = probed_margin(hip, run)
    bad_probe = count_bad()
    assert bad_probe == 0 or bad != 0, f"the probing instantiations changed {bad_probe} results (the product kernels were right)"
    bad += bad_probe
    assert margin < MARGIN_MAX, f"rounding margin too thin: max |x - round(x)| = {margin} (N = {n}, base2k {base2k}, {size} limbs, tensor {mode})"
    del d_a, d_b, d_r, d_want, d_prev
    torch.cuda.empty_cache()
    return bad, notes


@pytest.mark.parametrize("mode", ["apply", "add_assign", "square"])
def test_config5_tensoring_16_limbs_pool_parity_at_bench_batch(mods, mode):
    """BASELINE configs[4], multiply half at the bench geometry of tools/bench_tensor.py: N = 2^16, 16 limbs, rank 1, 259 pairs per call -
    every k_mid_cnv tile of both workgroups per CU over many tiles per workgroup, every NzCombine mode of the normalizing inverse column
    pass (plain / negated / read-modify-write stores, mode 5 for apply and square) - every tensor checked (batch 2-3
    only)."""
    n = 65536
    ref, hip = mods(n)
    bad, notes = _tensor_pool_parity(hip, ref, n, 1, 16, 12, 16 * 12 - 20, mode, batch=259, pool=3, seed=6000 + len(mode))
    assert bad == 0, (mode, bad)
    if os.environ.get("POULPY_DBG_TENSOR_FUSED") != "0" and os.environ.get("POULPY_DBG_TENSOR_COMBINE") != "0":
        assert "k_mid_cnv" in notes, notes
        if os.environ.get("POULPY_DBG_TENSOR_ALLTERMS") != "0":
            assert "k_mid_cnv3<16,16>" in notes, notes


@pytest.mark.parametrize("mode", ["apply", "add_assign", "square"])
def test_tensoring_all_terms_kernel_shapes(mods, mode):
    """k_mid_cnv3 (round 4: the three terms of a rank-1 tensoring in one launch, one operand vector in registers) on its 8-limb instantiation
    and on the windows it selects at write-back: offsets below base2k (negative bit offset, window at limb 0), around the middle, near the top
    of the product (zero rows beyond a_size + b_size - 2), results shorter and longer than the operands; batch 67 at N = 2^13 (many tiles per
    CU).  The per-term kernel (POULPY_DBG_TENSOR_ALLTERMS=0) must give the same digits - both against the oracle."""
    n = 8192
    ref, hip = mods(n)
    for (size, res_size, off) in ((8, 8, 8 * 12 - 9), (8, 5, 40), (8, 12, 7), (8, 16, 3 * 12), (8, 8, 15 * 12 + 5), (8, 3, 14 * 12)):
        bad, notes = _tensor_pool_parity(hip, ref, n, 1, size, 12, off, mode, batch=67, pool=3, seed=6300 + off + res_size, res_size=res_size)
        assert bad == 0, (mode, size, res_size, off, bad)
        if os.environ.get("POULPY_DBG_TENSOR_FUSED") != "0" and os.environ.get("POULPY_DBG_TENSOR_COMBINE") != "0" and os.environ.get("POULPY_DBG_TENSOR_ALLTERMS") != "0":
            assert "k_mid_cnv3<8,8>" in notes, (notes, size, res_size, off)


def test_config5_tensoring_rank2_pool_parity(mods):
    """rank 2 (6 tensor columns, 3 pairwise terms) on the fused tensoring at a batch with many tiles per workgroup: N = 2^13, 8 limbs, 515
    pairs."""
    n = 8192
    ref, hip = mods(n)
    for mode in ("apply", "add_assign", "square"):
        bad, notes = _tensor_pool_parity(hip, ref, n, 2, 8, 12, 8 * 12 - 9, mode, batch=515, pool=3, seed=6100 + len(mode))
        assert bad == 0, (mode, bad)


def test_config5_relinearize_16_limbs_pool_parity_at_bench_batch(mods):
    """BASELINE configs[4]: tensor + relinearization at the bench geometry - the GLWETensors the tensoring produced at 16 limbs
    (3 columns) through the tensor key (1 -> 1, dnum 16, 16 limbs), 259 per call, pinned key: k_mid128r<2,32,..> with one input column and the
    tail that adds an operand to every column (`small_all`).  Every output checked."""
    import torch
    from poulpy_amd.hal import GlweOpParams
    n, rank, size, base2k, dnum, batch, pool = 65536, 1, 16, 12, 16, 259, 5
    ref, hip = mods(n)
    rng = seeded(6200)
    cols, pairs = rank + 1, rank * (rank + 1) // 2
    mat = MatZnx(n, dnum, pairs, cols, size).fill_uniform(base2k, rng)
    pr, ph = ref.vmp_pmat_alloc(dnum, pairs, cols, size), hip.vmp_pmat_alloc(dnum, pairs, cols, size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    a_pool = np.empty((pool, size, cols + pairs, n), dtype=np.int64)
    want_pool = np.empty((pool, size, cols, n), dtype=np.int64)
    for i in range(pool):
        a = VecZnx(n, cols + pairs, size).fill_uniform(base2k, rng)
        a_pool[i] = a.data
        r = VecZnx(n, cols, size)
        ref.glwe_tensor_relinearize(r, base2k, a, base2k, pr, 1, base2k)
        want_pool[i] = r.data
    dev = torch.device("cuda", 0)
    idx = torch.arange(batch, device=dev) % pool
    d_a = torch.from_numpy(a_pool).to(dev)[idx].contiguous()
    d_want = torch.from_numpy(want_pool).to(dev)
    d_r = torch.full((batch, size, cols, n), 0x5A5A5A5A, dtype=torch.int64, device=dev)
    key = torch.from_numpy(ph.data).to(dev)
    torch.cuda.synchronize()
    p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k,
                     res_size=size, res_base2k=base2k, rank_out=rank)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    hip.pin_key(ptr(key), dnum, pairs, cols, size)
    try:
        hip.glwe_tensor_relinearize_batched(ptr(d_r), ptr(d_a), ptr(key), p, batch)
        hip.sync()
    finally:
        hip.unpin_key(ptr(key))
    bad = int((d_r != d_want[idx]).flatten(1).any(dim=1).sum().item())
    del d_a, d_r, d_want, key
    torch.cuda.empty_cache()
    assert bad == 0


def test_config4_blind_rotation_n16384(mods):
    """BASELINE configs[3], N = 2^14 leg: CGGI block-binary blind rotation at N = 16384 (accumulators do not fit LDS: the composed
    path with the LDS-staged block step) on a short LWE so that the CPU oracle finishes in about a minute; batch 9 (ragged
    against the 8-ciphertext tile of the block step)."""
    from tests.test_gpu_parity import _run_blind_rotation
    n = 16384
    ref, hip = mods(n)
    got, want = _run_blind_rotation(hip, ref, n, 1, 8, 4, 2, 2, 2, 17, batch=9, seed=16384)
    assert np.array_equal(got, want)
    # standard variant (block size 1) on the same ring
    got, want = _run_blind_rotation(hip, ref, n, 1, 3, 1, 2, 2, 2, 15, batch=3, seed=16385)
    assert np.array_equal(got, want)


def _br_pool_parity(hip, ref, n, rank, n_lwe, block_size, dnum, brk_size, res_size, base2k, batch, pool, seed):
    """Blind rotation at the launch geometry of tools/bench_blind_rotation.py: a pool of `pool` distinct LWE
    ciphertexts (coprime with every tile size: 2, 4, 8 ciphertexts per workgroup) whose rotations the oracle computes, replicated on
    the device to `batch` ciphertexts; EVERY output is compared with its pool entry's oracle result on the device.  Returns the number
    of mismatching ciphertexts."""
    import torch
    from poulpy_amd.hal import BlindRotationParams
    from poulpy_amd.layouts import MatZnx, VecZnx
    rng = np.random.default_rng(seed)
    cols = rank + 1
    lut = VecZnx(n, 1, res_size).fill_uniform(base2k, rng)
    brk_r = np.empty((n_lwe, n * dnum * cols * cols * brk_size), dtype=np.float64)
    brk_h = np.empty_like(brk_r)
    for i in range(n_lwe):
        mat = MatZnx(n, dnum, cols, cols, brk_size).fill_uniform(base2k, rng)
        pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, brk_size), hip.vmp_pmat_alloc(dnum, cols, cols, brk_size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        brk_r[i], brk_h[i] = pr.data.reshape(-1), ph.data.reshape(-1)
    lwe_pool = rng.integers(-n, n, (pool, n_lwe + 1), dtype=np.int64)
    lwe_pool[0, 1] = 0
    xpa = ref.blind_rotation_x_pow_a()
    want_pool = np.empty((pool, res_size, cols, n), dtype=np.int64)
    for b in range(pool):
        r = VecZnx(n, cols, res_size)
        ref.blind_rotation_execute(r, base2k, np.ascontiguousarray(lwe_pool[b]), lut, brk_r, dnum, brk_size, block_size, xpa)
        want_pool[b] = r.data
    dev = torch.device("cuda", 0)
    idx = torch.arange(batch, device=dev) % pool
    d_lwe = torch.from_numpy(lwe_pool).to(dev)[idx].contiguous()
    d_want = torch.from_numpy(want_pool).to(dev)
    d_lut = torch.from_numpy(lut.data).to(dev)
    d_brk = torch.from_numpy(brk_h).to(dev)
    d_res = torch.full((batch, res_size, cols, n), 0x3333, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    p = BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block_size, dnum=dnum, brk_size=brk_size, base2k=base2k,
                            res_size=res_size, lut_size=res_size)
    C = __import__("ctypes")
    run = lambda: hip.blind_rotation_execute_batched(C.c_void_p(d_res.data_ptr()), C.c_void_p(d_lwe.data_ptr()), C.c_void_p(d_lut.data_ptr()),
                                                     C.c_void_p(d_brk.data_ptr()), p, batch)
    run()
    hip.sync()
    bad = int((d_res != d_want[idx]).flatten(1).any(dim=1).sum().item())
    # once more with the rounding margin measured on the rotation's own kernels (one-kernel carry phase / small-ring tail / pipeline tail)
    d_res.fill_(0x3333)
    torch.cuda.synchronize()   # torch's stream and the module's stream are not ordered
    margin = probed_margin(hip, run)
    bad_probe = int((d_res != d_want[idx]).flatten(1).any(dim=1).sum().item())
    assert bad_probe == 0 or bad != 0, f"the probing instantiations changed {bad_probe} results (the product kernels were right)"
    bad += bad_probe
    assert margin < MARGIN_MAX, f"rounding margin too thin: max |x - round(x)| = {margin} (blind rotation, N = {n}, base2k {base2k})"
    return bad


@pytest.mark.parametrize("n,rank,note", [(1024, 1, "one-kernel path, two ciphertexts per workgroup"),
                                         (1024, 2, "rank 2 (circuit-bootstrapping key): small-ring transforms + LDS-staged block step"),
                                         (2048, 1, "small-ring transforms + LDS-staged block step"),
                                         (16384, 1, "block step on the three-kernel pipeline (k_mid128 BR)")])
def test_config4_blind_rotation_pool_parity_at_bench_batch(mods, n, rank, note):
    """BASELINE configs[3] at the batch the blind-rotation bench runs (>= 1027 ciphertexts: ragged against every tile size), every
    output checked.  Short LWE (two blocks + a dropped partial one) so that the oracle's pool stays in seconds; the launch geometry
    over the batch - tiles, XCD slots, last partial tile - is the bench's."""
    ref, hip = mods(n)
    dnum, bsz, rsz = (3, 4, 4) if rank == 2 else (3, 3, 3)
    batch = 1027                            # N = 2^14 too: 128 tiles of 8 + 3, the bench's per-GPU batch is 1024
    bad = _br_pool_parity(hip, ref, n, rank, n_lwe=15, block_size=7, dnum=dnum, brk_size=bsz, res_size=rsz, base2k=13, batch=batch, pool=7,
                          seed=4000 + n + rank)
    assert bad == 0, (note, bad)


@pytest.mark.parametrize("n,rank,blk,dnum,bsz,rsz,k,note", [
    (512, 3, 3, 1, 2, 1, 18, "the reference bench's shape: 4 rows, 8 output polynomials"),
    (512, 1, 3, 3, 3, 3, 13, "6 rows, 6 output polynomials in two groups of 3"),
    (1024, 1, 7, 3, 3, 3, 13, "N = 1024: 32-bit accumulators when two ciphertexts share a workgroup"),
    (1024, 1, 2, 2, 2, 2, 17, "N = 1024, 4 rows, 4 output polynomials: 64-bit accumulators for two ciphertexts"),
    (256, 1, 3, 2, 3, 2, 15, "N = 256: always one ciphertext per workgroup"),
])
def test_one_kernel_rotation_forms_by_batch(mods, n, rank, blk, dnum, bsz, rsz, k, note):
    """Round 5: the one-kernel rotation picks its form from the batch (br_forms.hpp) - one ciphertext per 512-thread workgroup while every
    ciphertext can have a CU of its own, two above that, and at N = 512 one per 256-thread workgroup above two per CU.  Each regime at a
    ragged batch, every output against the oracle's pool, the dispatched form read back from the module."""
    import torch
    ref, hip = mods(n)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    regimes = [(ncu // 3 + 2, "CT=1,NT=512"), (ncu + 45, "CT=1,NT=512" if n == 256 else "CT=2,NT=512"),
               (2 * ncu + 91, "CT=1,NT=256" if n == 512 else ("CT=1,NT=512" if n == 256 else "CT=2,NT=512"))]
    for batch, form in regimes:
        hip.dispatch_notes(reset=True)
        bad = _br_pool_parity(hip, ref, n, rank, n_lwe=2 * blk + 1, block_size=blk, dnum=dnum, brk_size=bsz, res_size=rsz, base2k=k, batch=batch,
                              pool=7, seed=5200 + n + rank + batch)
        notes = hip.dispatch_notes()
        assert bad == 0, (note, batch, bad)
        assert "k_br_fused" in notes and form in notes, (note, batch, form, notes)


@pytest.mark.parametrize("n,rank,blk,dnum,bsz,rsz,k,form", [
    (512, 3, 3, 1, 2, 3, 18, "R0=4,CT=2,NT=512,PJ=1,MR=4,CG=4,A32=1"),     # 12 accumulator polynomials per ciphertext: 32-bit digits in LDS
    (512, 1, 3, 3, 3, 7, 13, "R0=4,CT=2,NT=512,PJ=1,MR=6,CG=3,A32=1"),     # 6 rows, 7 result limbs
    (1024, 1, 2, 3, 3, 1, 13, "R0=8,CT=2,NT=512,PJ=2,MR=6,CG=3,A32=0"),    # one result limb: 64-bit accumulators fit beside 12 work polynomials
    (1024, 1, 2, 2, 2, 4, 14, "R0=8,CT=2,NT=512,PJ=1,MR=4,CG=4,A32=1"),    # 4 result limbs
    (1024, 2, 2, 2, 1, 1, 14, "R0=8,CT=2,NT=512,PJ=1,MR=6,CG=3,A32=0"),    # rank 2 with a one-limb key: 3 output polynomials, one column group
    (1024, 2, 2, 1, 1, 4, 14, "R0=8,CT=2,NT=512,PJ=1,MR=6,CG=3,A32=1"),
])
def test_one_kernel_rotation_two_ciphertext_forms(mods, n, rank, blk, dnum, bsz, rsz, k, form):
    """The two-ciphertext forms of the one-kernel rotation that only shapes with many accumulator limbs (32-bit digits in LDS) or a single one reach -
    since round 5 a batch above one ciphertext per CU is what selects them, so the small-batch shape tests no longer do.  Ragged batch, every output
    against the oracle's pool; the same shape at a small batch runs the one-ciphertext form."""
    import torch
    ref, hip = mods(n)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for batch, want in ((ncu + 45, form), (7, form.replace("CT=2", "CT=1").replace("A32=1", "A32=0"))):
        hip.dispatch_notes(reset=True)
        bad = _br_pool_parity(hip, ref, n, rank, n_lwe=2 * blk + 1, block_size=blk, dnum=dnum, brk_size=bsz, res_size=rsz, base2k=k, batch=batch,
                              pool=5, seed=5300 + n + rank + rsz + batch)
        notes = hip.dispatch_notes()
        assert bad == 0, (form, batch, bad)
        assert want in notes, (batch, want, notes)


def test_seeded_one_kernel_rotation_sweep_above_one_per_cu(mods):
    """16 random blind-rotation shapes (fixed seed; POULPY_SWEEP_SEED / POULPY_SWEEP_CASES for others) at N = 256 / 512 / 1024 and a ragged batch above
    one ciphertext per CU - where the one-kernel rotation puts two ciphertexts into a workgroup (or, at N = 512, uses the 256-thread form): the
    small-batch sweep of test_gpu_parity.py only reaches the one-ciphertext forms since round 5.  Every output against the oracle's pool; shapes
    the one-kernel path does not take run the composed path and are checked all the same."""
    import torch
    rng = np.random.default_rng(int(os.environ.get("POULPY_SWEEP_SEED", "5151")))
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    fused = 0
    for case in range(int(os.environ.get("POULPY_SWEEP_CASES", "16"))):
        n = int(2 ** rng.integers(8, 11))
        rank = int(rng.integers(1, 4))
        blk = int(rng.integers(1, 6))
        n_lwe = int(rng.integers(1, 3)) * blk + int(rng.integers(0, blk))
        dnum, bsz, rsz = (int(x) for x in rng.integers(1, 4, 3))
        k = int(rng.integers(10, 16))
        batch = ncu + int(rng.integers(1, 2 * ncu))
        ref, hip = mods(n)
        hip.dispatch_notes(reset=True)
        desc = dict(case=case, n=n, rank=rank, blk=blk, n_lwe=n_lwe, dnum=dnum, bsz=bsz, rsz=rsz, k=k, batch=batch)
        if os.environ.get("POULPY_SWEEP_VERBOSE"):
            print(desc, flush=True)
        bad = _br_pool_parity(hip, ref, n, rank, n_lwe=n_lwe, block_size=blk, dnum=dnum, brk_size=bsz, res_size=rsz, base2k=k, batch=batch, pool=5,
                              seed=5400 + case)
        assert bad == 0, (desc, bad, hip.dispatch_notes())
        fused += "k_br_fused" in hip.dispatch_notes()
    assert fused >= 4, f"only {fused} of the sweep's shapes ran the one-kernel path"


def test_blind_rotation_block_larger_than_a_wave(mods):
    """Block sizes above 64: the one-kernel rotation keeps a block's rotation amounts one per lane (round 5) and leaves such shapes to the composed path;
    70 coefficients in one block and 2 x 65 + a dropped partial block, small and above-one-per-CU batches."""
    import torch
    n = 256
    ref, hip = mods(n)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for blk, n_lwe, batch in ((70, 70, 5), (65, 133, ncu + 9)):
        hip.dispatch_notes(reset=True)
        bad = _br_pool_parity(hip, ref, n, 1, n_lwe=n_lwe, block_size=blk, dnum=2, brk_size=2, res_size=2, base2k=14, batch=batch, pool=3,
                              seed=5500 + blk)
        assert bad == 0, (blk, n_lwe, batch, bad)
        assert "k_br_fused" not in hip.dispatch_notes(), hip.dispatch_notes()


# ------------------------------------------------------------------------------------------
# conversion quirks: round half away, saturating `as i64`, NaN -> 0, and the >= 2^51 slow path of the tail
# reference: poulpy-cpu-ref/src/reference/fft64/reim/conversion.rs:43-60
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [64, 4096, 65536])
def test_saturating_and_large_conversions(mods, n):
    """Inverse transforms whose values leave +-2^51 (exact slow path) and +-2^63 (saturation), and NaN (-> 0).

    A CONSTANT spectrum c (every frequency slot = c_re + i c_im) is exact in any FFT ordering and any butterfly schedule: each
    butterfly sees equal inputs, so sums double exactly and differences are exactly zero; the inverse transform is c_re at
    coefficient 0, c_im at coefficient n/2 and exact zeros elsewhere.  Constant fills also mean the same thing in the oracle's
    [re | im] bit-reversed layout and in this backend's interleaved natural-order layout, so both sides get identical inputs."""
    from poulpy_amd.layouts import VecZnxBig, VecZnxDft
    ref, hip = mods(n)
    m = n // 2
    cases = [
        (2.0 ** 52 + 3.0, -(2.0 ** 51 + 1.0)),     # beyond the 3-instruction fast conversion, exactly representable
        (2.0 ** 62 + 2.0 ** 11, -(2.0 ** 62)),     # large, in range
        (1.0e19, -1.0e19),                         # > 2^63: saturates to i64::MAX / i64::MIN
        (2.0 ** 63, -(2.0 ** 63)),                 # boundary: 2^63 saturates, -2^63 is i64::MIN exactly
        (0.5, -0.5), (1.5, -2.5), (0.49999999999999994, -0.49999999999999994), (2.0 ** 51 - 0.5, -(2.0 ** 51) + 0.5),   # half away from zero
        (float("nan"), float("nan")),              # NaN -> 0 (a NaN in one component only taints a backend-specific set of outputs)
    ]
    import math

    def rust_round_as_i64(x):
        if x != x:
            return 0
        t = math.trunc(x)
        r = t + (int(math.copysign(1, x)) if abs(x - t) >= 0.5 else 0)
        return max(-(2 ** 63), min(2 ** 63 - 1, r))

    for (cre, cim) in cases:
        dr, dh = VecZnxDft(n, 1, 2), VecZnxDft(n, 1, 2)
        for d in (dr, dh):
            d.data[...] = 0.0
        rr = dr.at(0, 0)           # the inverse transform divides by m: a constant spectrum c comes out as c at coefficient 0 / m
        rr[:m] = cre
        rr[m:] = cim
        hh = dh.at(0, 0).view(np.complex128)
        hh[:] = complex(cre, cim)
        want, got = VecZnxBig(n, 1, 2), VecZnxBig(n, 1, 2)
        want.data[...] = 7
        got.data[...] = 7
        ref.vec_znx_idft_apply(want, 0, dr, 0)
        hip.vec_znx_idft_apply(got, 0, dh, 0)
        assert np.array_equal(got.data, want.data), (n, cre, cim, got.data[0, 0, :2], want.data[0, 0, :2])
        # the expected values, stated directly (Rust: f64::round() as i64)
        assert int(got.data[0, 0, 0]) == rust_round_as_i64(cre) and int(got.data[0, 0, m]) == rust_round_as_i64(cim), (cre, cim)
        assert not got.data[0, 0, 1:m].any() and not got.data[0, 0, m + 1:].any()
    # the consuming form goes through the same epilogue
    d = VecZnxDft(n, 1, 1)
    d.at(0, 0).view(np.complex128)[:] = complex(1.0e19, -(2.0 ** 52 + 1.0))
    big = hip.vec_znx_idft_apply_consume(d)
    assert int(big.data[0, 0, 0]) == np.iinfo(np.int64).max and int(big.data[0, 0, m]) == -(2 ** 52 + 1)


@pytest.mark.parametrize("n", [4096, 65536])
def test_fused_tail_large_and_saturating_values(mods, n):
    """The fused tail's slow conversion path (`big >= 2^51`) and its saturation, through the batched external product: constant
    polynomials (only coefficient 0 non-zero) have constant spectra, and ONE non-zero input row keeps the product a single
    exactly-representable term, so GPU and oracle agree bit for bit although the values exceed 2^51 / 2^63."""
    from poulpy_amd.hal import GlweOpParams
    ref, hip = mods(n)
    rank, size, dnum, base2k = 1, 4, 4, 20
    cols = rank + 1
    for (ca, ck) in ((2 ** 30 + 1, 2 ** 22 + 1), (2 ** 40, 2 ** 30), (-(2 ** 40), 2 ** 30), (2 ** 31 + 5, -(2 ** 21 + 3))):
        mat = MatZnx(n, dnum, cols, cols, size)
        mat.data[:, :, :, :, 0] = ck                       # every key entry = the constant ck
        mat.data[0, 0, 1, 1, 0] = ck + 1
        pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, size), hip.vmp_pmat_alloc(dnum, cols, cols, size)
        ref.vmp_prepare(pr, mat)
        hip.vmp_prepare(ph, mat)
        batch = 5
        a_all = np.zeros((batch, size, cols, n), dtype=np.int64)
        want = np.empty_like(a_all)
        for b in range(batch):
            a_all[b, b % size, b % cols, 0] = ca + b       # one non-zero input polynomial -> one product term per output
            a = VecZnx(n, cols, size, a_all[b].copy())
            r = VecZnx(n, cols, size)
            ref.glwe_external_product(r, base2k, a, base2k, pr, 1, base2k)
            want[b] = r.data
        d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
        d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
        d_r = hip.device_alloc(want.nbytes)
        p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k,
                         res_size=size, res_base2k=base2k, rank_out=rank)
        for fuse in ((True, True), (False, False)):
            hip.lib.pz_memset_d(hip.handle, d_r.ptr, 0x11, want.nbytes)
            hip.set_fusion(*fuse)
            hip.glwe_external_product_batched(d_r.ptr, d_a.ptr, d_k.ptr, p, batch)
            hip.sync()
            hip.set_fusion(True, True)
            got = d_r.download(np.int64, want.size).reshape(want.shape)
            assert np.array_equal(got, want), (n, ca, ck, fuse)
        assert np.abs(want).max() > 0
        for buf in (d_a, d_k, d_r):
            buf.free()


# ------------------------------------------------------------------------------------------
# multi-GPU plumbing at the C-ABI level, single rank (the only size a 1-GPU box can run): RCCL communicator + broadcast
# ------------------------------------------------------------------------------------------
def test_pz_bcast_key_single_rank_rccl(mods):
    """pz_comm_unique_id / pz_comm_init_rank / pz_bcast_key / pz_comm_destroy with world size 1 (RCCL loads, the communicator
    comes up on the module's device, the in-place broadcast leaves the key intact and is ordered on the module stream), then an
    external product with the broadcast key is still bit-exact."""
    import ctypes as C
    from poulpy_amd.hal import PoulpyHipError
    n = 4096
    ref, hip = mods(n)
    assert hip.lib.pz_comm_world_size(hip.handle) == 0 and hip.lib.pz_comm_rank(hip.handle) == -1
    buf = hip.device_alloc(1 << 20)
    with pytest.raises(PoulpyHipError):
        hip.bcast_key(buf.ptr, 1 << 20, 0)            # no communicator yet
    uid = hip.comm_unique_id()
    assert len(uid) == 128
    hip.comm_init_rank(1, 0, uid)
    assert hip.lib.pz_comm_world_size(hip.handle) == 1 and hip.lib.pz_comm_rank(hip.handle) == 0
    with pytest.raises(PoulpyHipError):
        hip.comm_init_rank(1, 0, uid)                 # one communicator per module
    data = np.arange((200 << 20) // 8, dtype=np.float64)   # 200 MiB: four buckets
    big = hip.device_alloc(data.nbytes).upload(data)
    hip.bcast_key(big.ptr, data.nbytes, 0)
    with pytest.raises(PoulpyHipError):
        hip.bcast_key(big.ptr, data.nbytes, 3)        # root out of range
    hip.sync()
    assert np.array_equal(big.download(np.float64, data.size), data)
    assert _pool_parity(hip, ref, False, n, 1, 4, 17, 4, batch=40, pool=5, seed=77) == 0
    hip.comm_destroy()
    hip.comm_destroy()                                # idempotent
    assert hip.lib.pz_comm_world_size(hip.handle) == 0
    for b in (buf, big):
        b.free()


# ------------------------------------------------------------------------------------------
# dsize > 1 and res_base2k != key_base2k on the fused three-kernel pipeline
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [4096, 8192, 65536])
def test_fused_pipeline_digits_and_cross_base_output(mods, n):
    """poulpy-core's own sweeps (test_suite/external_product/glwe_ct.rs:34-40, keyswitching) vary dsize and use different base2k for
    input / key / output; on the 128-point-row plans those shapes now run the fused pipeline: digit-selected rows and column offsets
    inside the middle kernel's product (k_mid128<.., DS>), cross-base output as tail (key base) + one cross-base pass.  Fused and
    five-kernel paths against the oracle, bit-exact; includes dsize > a.size (empty product), dnum smaller than the digit count, and
    the tensor relinearization."""
    from tests.test_gpu_parity import _run_glwe_op
    from tests.test_gpu_cnv import _run_relinearize
    ref, hip = mods(n)
    big = n == 65536
    shapes = [  # ks, rank, rank_out, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b
        (False, 1, 1, 5, 12, 6, 12, 3, 2, 4, 12), (False, 1, 1, 5, 12, 6, 12, 2, 3, 5, 12), (False, 2, 2, 4, 13, 5, 13, 1, 4, 4, 13),
        (True, 1, 1, 6, 12, 6, 12, 3, 2, 5, 12), (True, 2, 1, 5, 12, 7, 12, 2, 3, 6, 12), (True, 1, 2, 4, 12, 5, 12, 1, 5, 4, 12),
        (False, 1, 1, 5, 12, 6, 12, 5, 1, 4, 15), (True, 1, 1, 5, 14, 6, 14, 5, 1, 6, 11), (False, 1, 1, 4, 17, 6, 13, 3, 2, 4, 15),
        (True, 2, 2, 4, 15, 5, 12, 2, 2, 4, 14),
    ]
    if big:
        shapes = shapes[:2] + shapes[3:4] + shapes[6:9]
    for (ks, rank, rank_out, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b) in shapes:
        for fuse in ((True, True), (False, False)):
            got, want = _run_glwe_op(hip, ref, ks, n, rank, rank_out, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b,
                                     batch=3 if big else 9, seed=n + dsize * 7 + a_b + r_b, fuse=fuse)
            assert np.array_equal(got, want), (n, ks, rank, rank_out, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b, fuse)
    for (rank, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b) in ((1, 5, 12, 6, 12, 3, 2, 5, 12), (2, 4, 13, 5, 13, 4, 1, 4, 11),
                                                                          (1, 4, 12, 5, 12, 2, 3, 4, 14)):
        if big and rank == 2:
            continue
        got, want = _run_relinearize(hip, ref, n, rank, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b, batch=3 if big else 7,
                                     seed=n + rank + dsize)
        assert np.array_equal(got, want), (n, "relinearize", rank, a_size, a_b, key_size, k_b, dnum, dsize, res_size, r_b)


@pytest.mark.parametrize("n", [8192, 65536])
def test_digit_groups_on_the_interleaved_middle_kernel(mods, n):
    """dsize 2 at the bench's shapes (8 limbs, dnum 4: 16 product terms on 16 input polynomials for the external product, 8 terms on 8 for the
    key switch) runs k_mid128r<.., DS> since round 3 (the digit-group addressing on the interleaved schedule, terms in program order);
    smaller digit shapes stay on k_mid128<.., DS>.  Bit-exact against the oracle with a partial last tile (5 = 4 + 1 ciphertexts) and more
    than two tiles per workgroup row (9); the kernel that ran is read back from the library."""
    from tests.test_gpu_parity import _run_glwe_op
    ref, hip = mods(n)
    for ks, want_note in ((False, "k_mid128r<CT=4,NP=16,NR=16,HALFIN=0"), (True, "k_mid128r<CT=4,NP=16,NR=8,HALFIN=1")):
        for batch in ((5,) if n == 65536 else (5, 9)):
            hip.dispatch_notes(reset=True)
            got, want = _run_glwe_op(hip, ref, ks, n, 1, 1, 8, 12, 8, 12, 4, 2, 8, 12, batch=batch, seed=n + batch + int(ks), fuse=(True, True))
            assert np.array_equal(got, want), (n, ks, batch)
            notes = hip.dispatch_notes()
            if os.environ.get("POULPY_DBG_MID_R") != "0":   # (the A/B knob that keeps every plain product on k_mid128)
                assert want_note in notes and "DS=1" in notes, notes
    # a digit shape outside the ring kernel's two forms (10 input polynomials): the older kernel, same answer
    hip.dispatch_notes(reset=True)
    got, want = _run_glwe_op(hip, ref, False, n, 1, 1, 5, 12, 6, 12, 3, 2, 4, 12, batch=5, seed=n + 1, fuse=(True, True))
    assert np.array_equal(got, want)
    assert "k_mid128<CT=4,NP=16,DS=1>" in hip.dispatch_notes()
