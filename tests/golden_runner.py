"""Runs one golden fixture (tests/golden/*.npz) through a module that has the hal method names
(oracle.ref.RefModule on CPU, poulpy_amd.hal.Module on the GPU) and compares bit for bit."""
from __future__ import annotations

import glob
import os

import numpy as np

from poulpy_amd.layouts import MatZnx, ScalarZnx, SvpPPol, VecZnx, VecZnxBig

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def run_fixture(path: str, module_factory) -> None:
    z = np.load(path)
    kind, n, base2k = str(z["kind"]), int(z["n"]), int(z["base2k"])
    mod = module_factory(n)
    if kind == "vmp":
        a_np, mat_np = z["a"], z["mat"]
        a_size, cols_in, _ = a_np.shape
        rows, _, size, cols_out, _ = mat_np.shape
        res_size = z["res_big"].shape[0]
        a = VecZnx(n, cols_in, a_size, np.ascontiguousarray(a_np))
        mat = MatZnx(n, rows, cols_in, cols_out, size, np.ascontiguousarray(mat_np))
        ad = mod.vec_znx_dft_alloc(cols_in, a_size)
        for j in range(cols_in):
            mod.vec_znx_dft_apply(1, 0, ad, j, a, j)
        pm = mod.vmp_pmat_alloc(rows, cols_in, cols_out, size)
        mod.vmp_prepare(pm, mat)
        rd = mod.vec_znx_dft_alloc(cols_out, res_size)
        mod.vmp_apply_dft_to_dft(rd, ad, pm, int(z["limb_offset"]))
        big = mod.vec_znx_idft_apply_consume(rd)
        assert np.array_equal(big.data, z["res_big"]), f"{path}: res_big differs"
        res = VecZnx(n, cols_out, res_size)
        res.data[...] = -7
        for c in range(cols_out):
            mod.vec_znx_big_normalize(res, base2k, 0, c, big, base2k, c)
        assert np.array_equal(res.data, z["res_norm"]), f"{path}: normalized result differs"
    elif kind == "svp":
        s_np, b_np = z["s"], z["b"]
        size, cols, _ = b_np.shape
        s = ScalarZnx(n, cols, 1, np.ascontiguousarray(s_np))
        b = VecZnx(n, cols, size, np.ascontiguousarray(b_np))
        pp = SvpPPol(n, cols)
        d = mod.vec_znx_dft_alloc(cols, size)
        for c in range(cols):
            mod.svp_prepare(pp, c, s, c)
        for c in range(cols):
            mod.svp_apply_dft(d, c, pp, c, b, c)
        big = mod.vec_znx_idft_apply_consume(d)
        assert np.array_equal(big.data, z["res_big"]), f"{path}: svp product differs"
    elif kind == "normalize":
        a_np = z["a"]
        a = VecZnxBig(n, 1, a_np.shape[0], np.ascontiguousarray(a_np))
        res = VecZnx(n, 1, z["res"].shape[0])
        res.data[...] = 3
        mod.vec_znx_big_normalize(res, base2k, 0, 0, a, base2k, 0)
        assert np.array_equal(res.data, z["res"]), f"{path}: normalize differs"
    elif kind == "external_product":
        a_np, mat_np, want = z["a"], z["mat"], z["res"]
        a_size, cols, _ = a_np.shape
        dnum, _, key_size, _, _ = mat_np.shape
        a = VecZnx(n, cols, a_size, np.ascontiguousarray(a_np))
        mat = MatZnx(n, dnum, cols, cols, key_size, np.ascontiguousarray(mat_np))
        pm = mod.vmp_pmat_alloc(dnum, cols, cols, key_size)
        mod.vmp_prepare(pm, mat)
        # the op sequence of poulpy-core/src/external_product/glwe.rs:99-141,197-271 (dsize = 1) through the HAL methods
        ad = mod.vec_znx_dft_alloc(cols, a_size)
        for j in range(cols):
            mod.vec_znx_dft_apply(1, 0, ad, j, a, j)
        rd = mod.vec_znx_dft_alloc(cols, key_size)
        mod.vmp_apply_dft_to_dft(rd, ad, pm, 0)
        big = mod.vec_znx_idft_apply_consume(rd)
        res = VecZnx(n, cols, want.shape[0])
        for c in range(cols):
            mod.vec_znx_big_normalize(res, base2k, 0, c, big, base2k, c)
        assert np.array_equal(res.data, want), f"{path}: external product differs"
    elif kind == "glwe_automorphism":
        a_np, mat_np, gal = z["a"], z["mat"], int(z["p"])
        a_size, cols, _ = a_np.shape
        dnum, rank, key_size, _, _ = mat_np.shape
        a = VecZnx(n, cols, a_size, np.ascontiguousarray(a_np))
        mat = MatZnx(n, dnum, rank, cols, key_size, np.ascontiguousarray(mat_np))
        pm = mod.vmp_pmat_alloc(dnum, rank, cols, key_size)
        mod.vmp_prepare(pm, mat)

        def keyswitch_big():  # keyswitching/glwe.rs:207-239 through the HAL methods
            ad = mod.vec_znx_dft_alloc(rank, a_size)
            for j in range(rank):
                mod.vec_znx_dft_apply(1, 0, ad, j, a, j + 1)
            rd = mod.vec_znx_dft_alloc(cols, key_size)
            mod.vmp_apply_dft_to_dft(rd, ad, pm, 0)
            big = mod.vec_znx_idft_apply_consume(rd)
            mod.vec_znx_big_add_small_assign(big, 0, a, 0)
            return big

        # glwe_automorphism (automorphism/glwe_ct.rs:51-72): normalize, then the automorphism of the result
        big = keyswitch_big()
        res = VecZnx(n, cols, z["res_auto"].shape[0])
        for c in range(cols):
            mod.vec_znx_big_normalize(res, base2k, 0, c, big, base2k, c)
            mod.vec_znx_automorphism_assign(gal, res, c)
        assert np.array_equal(res.data, z["res_auto"]), f"{path}: glwe_automorphism differs"
        # glwe_automorphism_add (:96-140): automorphism of the big value, + a, normalize
        big = keyswitch_big()
        res = VecZnx(n, cols, z["res_add"].shape[0])
        for c in range(cols):
            mod.vec_znx_big_automorphism_assign(gal, big, c)
            mod.vec_znx_big_add_small_assign(big, c, a, c)
            mod.vec_znx_big_normalize(res, base2k, 0, c, big, base2k, c)
        assert np.array_equal(res.data, z["res_add"]), f"{path}: glwe_automorphism_add differs"
    elif kind == "glwe_batched":
        a_np, mat_np, want = z["a"], z["mat"], z["res"]
        ks, rank = bool(int(z["keyswitch"])), int(z["rank"])
        batch, a_size, cols, _ = a_np.shape
        dnum, cols_in, key_size, _, _ = mat_np.shape
        res_size = want.shape[1]
        mat = MatZnx(n, dnum, cols_in, cols, key_size, np.ascontiguousarray(mat_np))
        pm = mod.vmp_pmat_alloc(dnum, cols_in, cols, key_size)
        mod.vmp_prepare(pm, mat)
        if hasattr(mod, "glwe_external_product_batched"):   # the device library: ONE batched call on device-resident ciphertexts
            from poulpy_amd.hal import GlweOpParams
            p = GlweOpParams(rank=rank, dnum=dnum, dsize=1, key_size=key_size, key_base2k=base2k, a_size=a_size, a_base2k=base2k,
                             res_size=res_size, res_base2k=base2k, rank_out=rank)
            d_a = mod.device_alloc(a_np.nbytes).upload(np.ascontiguousarray(a_np))
            d_k = mod.device_alloc(pm.data.nbytes).upload(pm.data)
            d_r = mod.device_alloc(want.nbytes)
            (mod.glwe_keyswitch_batched if ks else mod.glwe_external_product_batched)(d_r.ptr, d_a.ptr, d_k.ptr, p, batch)
            mod.sync()
            got = d_r.download(np.int64, want.size).reshape(want.shape)
            for buf in (d_a, d_k, d_r):
                buf.free()
        else:                                                # the oracle: the reference's per-ciphertext op
            got = np.empty_like(want)
            for b in range(batch):
                a = VecZnx(n, cols, a_size, np.ascontiguousarray(a_np[b]))
                res = VecZnx(n, cols, res_size)
                (mod.glwe_keyswitch if ks else mod.glwe_external_product)(res, base2k, a, base2k, pm, 1, base2k)
                got[b] = res.data
        assert np.array_equal(got, want), f"{path}: batched GLWE product differs"
    elif kind == "lwe_keyswitch":
        lwe, mat_np, want, want_ms, n2 = z["lwe"], z["mat"], z["res"], z["mod_switched"], int(z["n2"])
        batch, size, len_in = lwe.shape
        n_out = want.shape[2] - 1
        dnum, _, key_size, _, _ = mat_np.shape
        mat = MatZnx(n, dnum, 1, 2, key_size, np.ascontiguousarray(mat_np))
        pm = mod.vmp_pmat_alloc(dnum, 1, 2, key_size)
        mod.vmp_prepare(pm, mat)
        if hasattr(mod, "lwe_keyswitch_batched"):
            from poulpy_amd.hal import GlweOpParams
            p = GlweOpParams(rank=1, dnum=dnum, dsize=1, key_size=key_size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                             res_base2k=base2k, rank_out=1)
            d_l = mod.device_alloc(lwe.nbytes).upload(np.ascontiguousarray(lwe))
            d_k = mod.device_alloc(pm.data.nbytes).upload(pm.data)
            d_r = mod.device_alloc(want.nbytes)
            d_m = mod.device_alloc(want_ms.nbytes)
            mod.lwe_keyswitch_batched(d_r.ptr, n_out, d_l.ptr, len_in - 1, d_k.ptr, p, batch)
            mod.lwe_mod_switch_2n_batched(d_m.ptr, d_l.ptr, len_in - 1, size, base2k, n2, False, batch)
            mod.sync()
            got = d_r.download(np.int64, want.size).reshape(want.shape)
            got_ms = d_m.download(np.int64, want_ms.size).reshape(want_ms.shape)
            for buf in (d_l, d_k, d_r, d_m):
                buf.free()
        else:
            got = np.stack([mod.lwe_keyswitch(n_out, size, base2k, lwe[b], base2k, pm, 1, base2k) for b in range(batch)])
            got_ms = np.stack([mod.mod_switch_2n(n2, lwe[b], base2k, False) for b in range(batch)])
        assert np.array_equal(got_ms, want_ms), f"{path}: mod_switch_2n differs"
        assert np.array_equal(got, want), f"{path}: lwe_keyswitch differs"
    else:
        raise AssertionError(f"unknown fixture kind {kind}")
