"""ORACLE — TEST INFRASTRUCTURE ONLY ("parity unpinned" by reference fixtures, see fft64_ref.h).

ctypes wrapper over oracle/_build/libpoulpy_oracle.so exposing the C restatement of
poulpy-cpu-ref's FFT64 family through the same method names and container classes as
poulpy_amd.hal.Module, so that parity tests read `ref.op(...)` / `hip.op(...)` like the
reference's cross_backend_test_suite (poulpy-hal/src/test_suite/mod.rs:64-95).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from ctypes import c_double, c_int64, c_size_t, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_build", "libpoulpy_oracle.so")
SO_FAST = os.path.join(_HERE, "_build", "libpoulpy_oracle_fast.so")


def build(fast: bool = False) -> str:
    target = "fast" if fast else "all"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return SO_FAST if fast else SO


def load(fast: bool = False) -> C.CDLL:
    path = SO_FAST if fast else SO
    # always through make: a prebuilt library that is older than fft64_ref.c (it travels with the snapshot to the GPU box) is rebuilt, a fresh
    # one costs a no-op; without a compiler the existing file is used as it is
    try:
        build(fast)
    except (subprocess.CalledProcessError, OSError):
        if not os.path.exists(path):
            raise
    lib = C.CDLL(path)
    lib.pzr_tables_new.restype = c_void_p
    lib.pzr_tables_new.argtypes = [c_uint64]
    lib.pzr_tables_free.argtypes = [c_void_p]
    lib.pzr_tables_omg_fft.restype = C.POINTER(c_double)
    lib.pzr_tables_omg_ifft.restype = C.POINTER(c_double)
    lib.pzr_tables_omg_fft.argtypes = [c_void_p]
    lib.pzr_tables_omg_ifft.argtypes = [c_void_p]
    for name in ("pzr_vmp_prepare_tmp_bytes", "pzr_vmp_apply_dft_to_dft_tmp_bytes", "pzr_vmp_apply_dft_tmp_bytes",
                 "pzr_vec_znx_normalize_tmp_bytes", "pzr_cnv_prepare_tmp_bytes", "pzr_cnv_apply_dft_tmp_bytes",
                 "pzr_cnv_pairwise_apply_dft_tmp_bytes", "pzr_cnv_by_const_apply_tmp_bytes"):
        getattr(lib, name).restype = c_size_t
    lib.pzr_msb_mask_bottom_limb.restype = c_int64
    lib.pzr_margin_probe_get.restype = c_double
    return lib


def _p(arr):
    return arr.ctypes.data_as(c_void_p)


def _sz(*xs):
    return [c_size_t(int(x)) for x in xs]


class RefModule:
    """``Module<FFT64Ref>`` restated (poulpy-cpu-ref/src/fft64/module.rs)."""

    def __init__(self, n: int, fast: bool = False):
        self.lib = load(fast)
        self._n = int(n)
        self.t = c_void_p(self.lib.pzr_tables_new(c_uint64(n)))
        if not self.t.value:
            raise ValueError(f"n must be a power of two >= 2 but is {n}")

    def n(self):
        return self._n

    def rounding_margin_of(self, run) -> float:
        """max |x - round(x)| over every value the oracle rounds (reim_to_znx_i64[_assign]) while `run()` executes: the oracle's own
        distance from a wrong limb on these inputs (pzr_margin_probe_*: diagnostic, single-threaded; 0.5 = a wrong limb)."""
        self.lib.pzr_margin_probe_set(1)
        try:
            run()
            return float(self.lib.pzr_margin_probe_get())
        finally:
            self.lib.pzr_margin_probe_set(0)

    def __del__(self):
        try:
            if self.t is not None and self.t.value:
                self.lib.pzr_tables_free(self.t)
                self.t = None
        except Exception:
            pass

    # raw transforms on [re | im] arrays
    def fft(self, data: np.ndarray):
        assert data.dtype == np.float64 and data.size == self._n
        self.lib.pzr_fft(self.t, _p(data))

    def ifft(self, data: np.ndarray):
        assert data.dtype == np.float64 and data.size == self._n
        self.lib.pzr_ifft(self.t, _p(data))

    def omg_fft(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.lib.pzr_tables_omg_fft(self.t), shape=(self._n,)).copy()

    def omg_ifft(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.lib.pzr_tables_omg_ifft(self.t), shape=(self._n,)).copy()

    # allocation helpers with the reference's layouts
    def vec_znx_dft_alloc(self, cols, size):
        from poulpy_amd.layouts import VecZnxDft
        return VecZnxDft(self._n, cols, size)

    def vec_znx_big_alloc(self, cols, size):
        from poulpy_amd.layouts import VecZnxBig
        return VecZnxBig(self._n, cols, size)

    def svp_ppol_alloc(self, cols):
        from poulpy_amd.layouts import SvpPPol
        return SvpPPol(self._n, cols)

    def vmp_pmat_alloc(self, rows, cols_in, cols_out, size):
        from poulpy_amd.layouts import VmpPMat
        return VmpPMat(self._n, rows, cols_in, cols_out, size)

    # VecZnxDft
    def vec_znx_dft_apply(self, step, offset, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_dft_apply(self.t, *_sz(step, offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                       *_sz(a.cols, a.size, a_col))

    def vec_znx_idft_apply(self, res, res_col, a, a_col, scratch=None):
        self.lib.pzr_vec_znx_idft_apply(self.t, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a.size, a_col))

    def vec_znx_idft_apply_tmpa(self, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_idft_apply_tmpa(self.t, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                             *_sz(a.cols, a.size, a_col))

    def vec_znx_idft_apply_consume(self, a):
        self.lib.pzr_vec_znx_idft_apply_consume(self.t, _p(a.data), *_sz(a.cols, a.size))
        return a.into_big()

    def _dft3(self, fn, res, res_col, a, a_col, b, b_col):
        fn(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a.size, a_col), _p(b.data),
           *_sz(b.cols, b.size, b_col))

    def _dft2(self, fn, res, res_col, a, a_col, *extra):
        fn(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a.size, a_col), *extra)

    def vec_znx_dft_add_into(self, res, res_col, a, a_col, b, b_col):
        self._dft3(self.lib.pzr_vec_znx_dft_add_into, res, res_col, a, a_col, b, b_col)

    def vec_znx_dft_sub(self, res, res_col, a, a_col, b, b_col):
        self._dft3(self.lib.pzr_vec_znx_dft_sub, res, res_col, a, a_col, b, b_col)

    def vec_znx_dft_add_assign(self, res, res_col, a, a_col):
        self._dft2(self.lib.pzr_vec_znx_dft_add_assign, res, res_col, a, a_col)

    def vec_znx_dft_add_scaled_assign(self, res, res_col, a, a_col, a_scale):
        self._dft2(self.lib.pzr_vec_znx_dft_add_scaled_assign, res, res_col, a, a_col, c_int64(a_scale))

    def vec_znx_dft_sub_assign(self, res, res_col, a, a_col):
        self._dft2(self.lib.pzr_vec_znx_dft_sub_assign, res, res_col, a, a_col)

    def vec_znx_dft_sub_negate_assign(self, res, res_col, a, a_col):
        self._dft2(self.lib.pzr_vec_znx_dft_sub_negate_assign, res, res_col, a, a_col)

    def vec_znx_dft_copy(self, step, offset, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_dft_copy(c_size_t(self._n), *_sz(step, offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                      *_sz(a.cols, a.size, a_col))

    def vec_znx_dft_zero(self, res, res_col):
        self.lib.pzr_vec_znx_dft_zero(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col))

    # SVP
    def svp_prepare(self, res, res_col, a, a_col):
        self.lib.pzr_svp_prepare(self.t, _p(res.data), *_sz(res.cols, res_col), _p(a.data), *_sz(a.cols, a_col))

    def svp_apply_dft(self, res, res_col, a, a_col, b, b_col):
        self.lib.pzr_svp_apply_dft(self.t, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a_col), _p(b.data),
                                   *_sz(b.cols, b.size, b_col))

    def svp_apply_dft_to_dft(self, res, res_col, a, a_col, b, b_col):
        self.lib.pzr_svp_apply_dft_to_dft(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                          *_sz(a.cols, a_col), _p(b.data), *_sz(b.cols, b.size, b_col))

    def svp_apply_dft_to_dft_assign(self, res, res_col, a, a_col):
        self.lib.pzr_svp_apply_dft_to_dft_assign(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                 *_sz(a.cols, a_col))

    # VMP
    def vmp_prepare(self, res, a, scratch=None):
        self.lib.pzr_vmp_prepare(self.t, _p(res.data), _p(a.data), *_sz(a.rows, a.cols_in, a.cols_out, a.size))

    def vmp_apply_dft(self, res, a, b, scratch=None):
        self.lib.pzr_vmp_apply_dft(self.t, _p(res.data), *_sz(res.cols, res.size), _p(a.data), *_sz(a.cols, a.size), _p(b.data),
                                   *_sz(b.rows, b.cols_in, b.cols_out, b.size))

    def vmp_apply_dft_to_dft(self, res, a, b, limb_offset=0, scratch=None):
        self.lib.pzr_vmp_apply_dft_to_dft(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size), _p(a.data), *_sz(a.cols, a.size),
                                          _p(b.data), *_sz(b.rows, b.cols_in, b.cols_out, b.size), c_size_t(limb_offset))

    def vmp_apply_dft_to_dft_tmp_bytes(self, res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size):
        return self.lib.pzr_vmp_apply_dft_to_dft_tmp_bytes(*_sz(a_size, b_rows, b_cols_in))

    def vmp_prepare_tmp_bytes(self, rows, cols_in, cols_out, size):
        return self.lib.pzr_vmp_prepare_tmp_bytes(c_size_t(self._n))

    def vec_znx_big_normalize_tmp_bytes(self):
        return self.lib.pzr_vec_znx_normalize_tmp_bytes(c_size_t(self._n))

    # VecZnxBig
    def vec_znx_big_normalize(self, res, res_base2k, res_offset, res_col, a, a_base2k, a_col, scratch=None):
        self.lib.pzr_vec_znx_normalize(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_base2k), c_int64(res_offset),
                                       c_size_t(res_col), _p(a.data), *_sz(a.cols, a.size, a_base2k, a_col))

    vec_znx_normalize = vec_znx_big_normalize

    def vec_znx_big_add_small_assign(self, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_big_add_small_assign(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                  *_sz(a.cols, a.size, a_col))

    # poulpy-core callers
    def glwe_external_product(self, res, res_base2k, a, a_base2k, pmat, dsize, ggsw_base2k):
        rank = a.cols - 1
        self.lib.pzr_glwe_external_product(self.t, c_size_t(rank), _p(res.data), *_sz(res.size, res_base2k), _p(a.data),
                                           *_sz(a.size, a_base2k), _p(pmat.data), *_sz(pmat.rows, pmat.size, dsize, ggsw_base2k))

    def glwe_keyswitch(self, res, res_base2k, a, a_base2k, pmat, dsize, key_base2k):
        rank_in, rank_out = a.cols - 1, res.cols - 1
        self.lib.pzr_glwe_keyswitch(self.t, *_sz(rank_in, rank_out), _p(res.data), *_sz(res.size, res_base2k), _p(a.data),
                                    *_sz(a.size, a_base2k), _p(pmat.data), *_sz(pmat.rows, pmat.size, dsize, key_base2k))

    # automorphism family (reference/vec_znx/automorphism.rs, fft64/vec_znx_big.rs:144-188, poulpy-core automorphism/glwe_ct.rs)
    def vec_znx_automorphism(self, p, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_automorphism(c_size_t(self._n), c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col),
                                          _p(a.data), *_sz(a.cols, a.size, a_col))

    def vec_znx_automorphism_assign(self, p, res, res_col, scratch=None):
        self.lib.pzr_vec_znx_automorphism_assign(c_size_t(self._n), c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col))

    vec_znx_big_automorphism = vec_znx_automorphism
    vec_znx_big_automorphism_assign = vec_znx_automorphism_assign

    def vec_znx_big_sub_small_assign(self, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_big_sub_small_assign(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                  *_sz(a.cols, a.size, a_col))

    def vec_znx_big_sub_small_negate_assign(self, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_big_sub_small_negate_assign(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col),
                                                         _p(a.data), *_sz(a.cols, a.size, a_col))

    AUTO_MODES = {"automorphism": 1, "add": 2, "sub": 3, "sub_negate": 4}

    def glwe_automorphism(self, res, res_base2k, a, a_base2k, pmat, dsize, key_base2k, p, mode="automorphism"):
        rank = a.cols - 1
        self.lib.pzr_glwe_automorphism(self.t, c_size_t(rank), C.c_int(self.AUTO_MODES[mode]), c_int64(p), _p(res.data),
                                       *_sz(res.size, res_base2k), _p(a.data), *_sz(a.size, a_base2k), _p(pmat.data),
                                       *_sz(pmat.rows, pmat.size, dsize, key_base2k))

    # i64 VecZnx limb-wise family (reference/vec_znx/add.rs, sub.rs, negate.rs, copy.rs)
    def vec_znx_add_into(self, res, res_col, a, a_col, b, b_col):
        self.lib.pzr_vec_znx_add_into(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                      *_sz(a.cols, a.size, a_col), _p(b.data), *_sz(b.cols, b.size, b_col))

    def vec_znx_sub(self, res, res_col, a, a_col, b, b_col):
        self.lib.pzr_vec_znx_sub(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                 *_sz(a.cols, a.size, a_col), _p(b.data), *_sz(b.cols, b.size, b_col))

    def _vec_znx_assign_op(self, mode, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_assign_op(C.c_int(mode), c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                       *_sz(a.cols, a.size, a_col))

    def vec_znx_add_assign(self, res, res_col, a, a_col):
        self._vec_znx_assign_op(0, res, res_col, a, a_col)

    def vec_znx_sub_assign(self, res, res_col, a, a_col):
        self._vec_znx_assign_op(1, res, res_col, a, a_col)

    def vec_znx_sub_negate_assign(self, res, res_col, a, a_col):
        self._vec_znx_assign_op(2, res, res_col, a, a_col)

    def vec_znx_negate(self, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_negate(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                    *_sz(a.cols, a.size, a_col))

    def vec_znx_negate_assign(self, res, res_col):
        self.lib.pzr_vec_znx_negate(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), None, *_sz(0, 0, 0))

    def vec_znx_copy(self, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_copy(c_size_t(self._n), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                  *_sz(a.cols, a.size, a_col))

    def vec_znx_lsh(self, base2k, k, res, res_col, a, a_col, scratch=None):
        self.lib.pzr_vec_znx_lsh(c_size_t(self._n), *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                 *_sz(a.cols, a.size, a_col))

    def vec_znx_rsh(self, base2k, k, res, res_col, a, a_col, scratch=None):
        self.lib.pzr_vec_znx_rsh(c_size_t(self._n), *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                 *_sz(a.cols, a.size, a_col))

    def vec_znx_lsh_assign(self, base2k, k, res, res_col, scratch=None):
        self.lib.pzr_vec_znx_lsh_assign(c_size_t(self._n), *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col))

    def ggsw_expand_row(self, ggsw, base2k, keys, dsize, key_base2k):
        """conversion/gglwe_to_ggsw.rs:116-268; ggsw: MatZnx (rows, cols_in = cols_out = rank+1), keys: rank prepared GGLWEs."""
        rank = ggsw.cols_out - 1
        arr = (C.c_void_p * len(keys))(*[k.data.ctypes.data for k in keys])
        self.lib.pzr_ggsw_expand_row(self.t, c_size_t(rank), _p(ggsw.data), *_sz(ggsw.rows, ggsw.size, base2k), arr,
                                     *_sz(keys[0].rows, keys[0].size, dsize, key_base2k))

    def ggsw_from_gglwe(self, ggsw, base2k, a, keys, dsize, key_base2k):
        """conversion/gglwe_to_ggsw.rs:32-61; a: the GGLWE as a MatZnx (rows = dnum, cols_in, cols_out = rank+1)."""
        rank = ggsw.cols_out - 1
        assert a.rows == ggsw.rows and a.cols_out == ggsw.cols_out
        arr = (C.c_void_p * len(keys))(*[k.data.ctypes.data for k in keys])
        self.lib.pzr_ggsw_from_gglwe(self.t, c_size_t(rank), _p(ggsw.data), *_sz(ggsw.rows, ggsw.size, base2k), _p(a.data),
                                     *_sz(a.cols_in, a.size), arr, *_sz(keys[0].rows, keys[0].size, dsize, key_base2k))

    # blind rotation (poulpy-bin-fhe/src/blind_rotation/algorithms/cggi), SURVEY.md 8f rank 2
    def vec_znx_rotate(self, p, res, res_col, a, a_col):
        self.lib.pzr_vec_znx_rotate(c_size_t(self._n), c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                    *_sz(a.cols, a.size, a_col))

    def vec_znx_mul_xp_minus_one_assign(self, p, res, res_col, scratch=None):
        self.lib.pzr_vec_znx_mul_xp_minus_one_assign(c_size_t(self._n), c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col))

    def vec_znx_normalize_assign(self, base2k, res, res_col, scratch=None):
        self.lib.pzr_vec_znx_normalize_assign(c_size_t(self._n), c_size_t(base2k), _p(res.data), *_sz(res.cols, res.size, res_col))

    def blind_rotation_x_pow_a(self) -> np.ndarray:
        """key_prepared.rs:66-74: the 2n prepared monomials X^i (SvpPPol, cols = 1), as one (2n, n) f64 array."""
        out = np.zeros((2 * self._n, self._n), dtype=np.float64)
        self.lib.pzr_blind_rotation_x_pow_a(self.t, _p(out))
        return out

    def blind_rotation_execute(self, res, base2k, lwe_2n: np.ndarray, lut, brk: np.ndarray, dnum, brk_size, block_size,
                               x_pow_a: np.ndarray):
        """algorithm.rs:265-440 on one LWE ciphertext: res GLWE (VecZnx cols = rank+1), lwe_2n = mod-switched [b, a_1..a_n],
        lut VecZnx(1, size), brk = (n_lwe, prepared GGSW doubles) array."""
        rank = res.cols - 1
        n_lwe = lwe_2n.shape[0] - 1
        assert brk.dtype == np.float64 and brk.flags["C_CONTIGUOUS"] and lwe_2n.dtype == np.int64
        self.lib.pzr_blind_rotation_execute(self.t, *_sz(rank, n_lwe, block_size), _p(res.data), *_sz(res.size, base2k), _p(lwe_2n),
                                            _p(lut.data), c_size_t(lut.size), _p(brk), *_sz(dnum, brk_size), _p(x_pow_a))

    def circuit_bootstrap_to_constant(self, ggsw, base2k, lwe_2n, lut, brk, brk_dnum, brk_size, glwe_size, block_size, x_pow_a, gals,
                                      atk, tsk, gap):
        """circuit_bootstrapping/circuit.rs:219-370 (to_exponent = false, one base2k); ggsw: MatZnx(res_dnum, cols, cols, res_size),
        atk: prepared automorphism keys of the trace steps (with gals), tsk: the rank prepared tensor keys."""
        rank = ggsw.cols_out - 1
        n_lwe = lwe_2n.shape[0] - 1
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ap = (c_void_p * ns)(*[pm.data.ctypes.data for pm in atk])
        tp = (c_void_p * len(tsk))(*[pm.data.ctypes.data for pm in tsk])
        self.lib.pzr_circuit_bootstrap_to_constant(
            self.t, *_sz(rank, base2k, n_lwe, block_size), _p(lwe_2n), _p(lut.data), c_size_t(lut.size), _p(brk),
            *_sz(brk_dnum, brk_size, glwe_size), _p(x_pow_a), c_size_t(ns), g, ap, *_sz(atk[0].rows, atk[0].size), _p(ggsw.data),
            *_sz(ggsw.rows, ggsw.size, gap), tp, *_sz(tsk[0].rows, tsk[0].size))

    def glwe_pack(self, res, base2k, cts: dict, log_gap_out, gals, pmats):
        """glwe_packing.rs:122-176 at one base2k / size: cts {index: VecZnx GLWE} (clobbered), gals / pmats for the log_n trace steps."""
        rank = res.cols - 1
        n = self._n
        slots = (c_void_p * n)()
        for j, ct in cts.items():
            assert ct.cols == res.cols and ct.size == res.size
            slots[j] = ct.data.ctypes.data
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ptrs = (c_void_p * ns)(*[pm.data.ctypes.data for pm in pmats])
        self.lib.pzr_glwe_pack(self.t, c_size_t(rank), _p(res.data), slots, *_sz(res.size, base2k, log_gap_out), g, ptrs,
                               *_sz(pmats[0].rows, pmats[0].size))

    def glwe_pack_bases(self, res, base2k, key_base2k, trace_size, cts: dict, log_gap_out, gals, pmats):
        """glwe_packing.rs:122-176 with the automorphism keys in their own base; trace_size: limbs of the closing trace's temporary."""
        rank = res.cols - 1
        n = self._n
        slots = (c_void_p * n)()
        for j, ct in cts.items():
            assert ct.cols == res.cols and ct.size == res.size
            slots[j] = ct.data.ctypes.data
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ptrs = (c_void_p * ns)(*[pm.data.ctypes.data for pm in pmats])
        self.lib.pzr_glwe_pack_bases(self.t, c_size_t(rank), _p(res.data), slots, *_sz(res.size, base2k, key_base2k, trace_size, log_gap_out),
                                     g, ptrs, *_sz(pmats[0].rows, pmats[0].size))

    def glwe_trace_assign_bases(self, res, res_base2k, conv_size, key_base2k, gals, pmats, dsize=1):
        """glwe_trace.rs:129-176 with res in another base than the keys (:153-163)."""
        rank = res.cols - 1
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ptrs = (c_void_p * ns)(*[pm.data.ctypes.data for pm in pmats])
        self.lib.pzr_glwe_trace_assign_bases(self.t, c_size_t(rank), _p(res.data), *_sz(res.size, res_base2k, conv_size, key_base2k),
                                             c_size_t(ns), g, ptrs, *_sz(pmats[0].rows, pmats[0].size, dsize))

    def circuit_bootstrap_to_exponent(self, ggsw, base2k, lwe_2n, lut, brk, brk_dnum, brk_size, glwe_size, block_size, x_pow_a, gals,
                                      atk, tsk, gap, log_gap_in, log_gap_out, log_domain):
        """circuit.rs:219-421 (to_exponent = true, one base2k); gals / atk: all log_n trace steps."""
        rank = ggsw.cols_out - 1
        n_lwe = lwe_2n.shape[0] - 1
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ap = (c_void_p * ns)(*[pm.data.ctypes.data for pm in atk])
        tp = (c_void_p * len(tsk))(*[pm.data.ctypes.data for pm in tsk])
        self.lib.pzr_circuit_bootstrap_to_exponent(
            self.t, *_sz(rank, base2k, n_lwe, block_size), _p(lwe_2n), _p(lut.data), c_size_t(lut.size), _p(brk),
            *_sz(brk_dnum, brk_size, glwe_size), _p(x_pow_a), g, ap, *_sz(atk[0].rows, atk[0].size), _p(ggsw.data),
            *_sz(ggsw.rows, ggsw.size, gap, log_gap_in, log_gap_out, log_domain), tp, *_sz(tsk[0].rows, tsk[0].size))

    def circuit_bootstrap_bases(self, ggsw, bases, to_exponent, lwe_2n, lut, brk, brk_dnum, brk_size, glwe_size, atk_glwe_size, trace_size,
                                block_size, x_pow_a, gals, atk, tsk, gap, log_gap_in=0, log_gap_out=0, log_domain=0):
        """circuit.rs:219-421 with one base2k per object: bases = (brk, atk, tsk, res); gals / atk: the steps of the full trace (constant
        mode) or all log_n steps (exponent mode)."""
        rank = ggsw.cols_out - 1
        n_lwe = lwe_2n.shape[0] - 1
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ap = (c_void_p * ns)(*[pm.data.ctypes.data for pm in atk])
        tp = (c_void_p * len(tsk))(*[pm.data.ctypes.data for pm in tsk])
        b = (c_size_t * 4)(*[int(x) for x in bases])
        self.lib.pzr_circuit_bootstrap_bases(
            self.t, c_size_t(rank), b, C.c_int(1 if to_exponent else 0), *_sz(n_lwe, block_size), _p(lwe_2n), _p(lut.data), c_size_t(lut.size),
            _p(brk), *_sz(brk_dnum, brk_size, glwe_size, atk_glwe_size, trace_size), _p(x_pow_a), c_size_t(ns), g, ap,
            *_sz(atk[0].rows, atk[0].size), _p(ggsw.data), *_sz(ggsw.rows, ggsw.size, gap, log_gap_in, log_gap_out, log_domain), tp,
            *_sz(tsk[0].rows, tsk[0].size))

    def blind_rotation_execute_extended(self, res, base2k, lwe_2n, luts, brk, dnum, brk_size, block_size, x_pow_a):
        """algorithm.rs:121-273: luts = (ext, lut_size, 1, n) i64 array (lut.data[j]), lwe_2n switched to 2*n*ext."""
        rank = res.cols - 1
        n_lwe = lwe_2n.shape[0] - 1
        ext, lut_size = luts.shape[0], luts.shape[1]
        assert luts.dtype == np.int64 and luts.flags["C_CONTIGUOUS"]
        self.lib.pzr_blind_rotation_execute_extended(self.t, *_sz(rank, n_lwe, block_size, ext), _p(res.data), *_sz(res.size, base2k),
                                                     _p(lwe_2n), _p(luts), c_size_t(lut_size), _p(brk), *_sz(dnum, brk_size), _p(x_pow_a))

    # glwe_trace (poulpy-core/src/glwe_trace.rs) and the shift it uses
    def vec_znx_rsh_assign(self, base2k, k, res, res_col, scratch=None):
        self.lib.pzr_vec_znx_rsh_assign(c_size_t(self._n), *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col))

    def glwe_trace_assign(self, res, base2k, gals, pmats, dsize=1):
        """glwe_trace.rs:129-176 at equal bases: gals[s], pmats[s] (prepared automorphism keys) for the steps skip..log_n."""
        rank = res.cols - 1
        ns = len(gals)
        if ns == 0:
            return
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ptrs = (c_void_p * ns)(*[pm.data.ctypes.data for pm in pmats])
        self.lib.pzr_glwe_trace_assign(self.t, c_size_t(rank), _p(res.data), *_sz(res.size, base2k), c_size_t(ns), g, ptrs,
                                       *_sz(pmats[0].rows, pmats[0].size, dsize))

    # convolution family (reference/fft64/convolution.rs; HalImpl cnv_*, hal_impl.rs:670-754)
    def cnv_pvec_left_alloc(self, cols, size):
        from poulpy_amd.layouts import CnvPVecL
        return CnvPVecL(self._n, cols, size)

    def cnv_pvec_right_alloc(self, cols, size):
        from poulpy_amd.layouts import CnvPVecR
        return CnvPVecR(self._n, cols, size)

    def cnv_prepare_left_tmp_bytes(self, res_size, a_size):
        return self.lib.pzr_cnv_prepare_tmp_bytes(*_sz(self._n, res_size, a_size))

    cnv_prepare_right_tmp_bytes = cnv_prepare_left_tmp_bytes
    cnv_prepare_self_tmp_bytes = cnv_prepare_left_tmp_bytes

    def cnv_apply_dft_tmp_bytes(self, cnv_offset, res_size, a_size, b_size):
        return self.lib.pzr_cnv_apply_dft_tmp_bytes(*_sz(res_size, a_size, b_size))

    def cnv_pairwise_apply_dft_tmp_bytes(self, cnv_offset, res_size, a_size, b_size):
        return self.lib.pzr_cnv_pairwise_apply_dft_tmp_bytes(*_sz(res_size, a_size, b_size))

    def cnv_by_const_apply_tmp_bytes(self, cnv_offset, res_size, a_size, b_size):
        return self.lib.pzr_cnv_by_const_apply_tmp_bytes(*_sz(res_size, a_size, b_size))

    def cnv_prepare_left(self, res, a, mask=-1, scratch=None):
        assert res.cols == a.cols
        self.lib.pzr_cnv_prepare(self.t, _p(res.data), *_sz(res.cols, res.size), _p(a.data), *_sz(a.cols, a.size), c_int64(mask))

    cnv_prepare_right = cnv_prepare_left

    def cnv_prepare_self(self, left, right, a, mask=-1, scratch=None):
        assert left.cols == right.cols == a.cols and left.size == right.size
        self.lib.pzr_cnv_prepare_self(self.t, _p(left.data), _p(right.data), *_sz(left.cols, left.size), _p(a.data),
                                      *_sz(a.cols, a.size), c_int64(mask))

    def cnv_apply_dft(self, cnv_offset, res, res_col, a, a_col, b, b_col, scratch=None):
        self.lib.pzr_cnv_apply_dft(*_sz(self._n, cnv_offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                   *_sz(a.size, a_col), _p(b.data), *_sz(b.size, b_col))

    def cnv_pairwise_apply_dft(self, cnv_offset, res, res_col, a, b, i, j, scratch=None):
        self.lib.pzr_cnv_pairwise_apply_dft(*_sz(self._n, cnv_offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                            c_size_t(a.size), _p(b.data), *_sz(b.size, i, j))

    def cnv_by_const_apply(self, cnv_offset, res, res_col, a, a_col, b, scratch=None):
        b = np.ascontiguousarray(b, dtype=np.int64)
        self.lib.pzr_cnv_by_const_apply(*_sz(self._n, cnv_offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                        *_sz(a.cols, a.size, a_col), _p(b), c_size_t(b.size))

    # GLWE tensoring (poulpy-core/src/operations/glwe.rs:541-913)
    def msb_mask_bottom_limb(self, base2k, k) -> int:
        return int(self.lib.pzr_msb_mask_bottom_limb(*_sz(base2k, k)))

    def glwe_tensor_apply(self, cnv_offset, res, res_base2k, a, a_effective_k, b, b_effective_k, ab_base2k, add_assign=False):
        rank = a.cols - 1
        assert res.cols == (rank + 1) * (rank + 2) // 2 and b.cols == a.cols
        self.lib.pzr_glwe_tensor_apply(self.t, *_sz(rank, cnv_offset), C.c_int(int(add_assign)), _p(res.data), *_sz(res.size, res_base2k),
                                       _p(a.data), *_sz(a.size, a_effective_k), _p(b.data), *_sz(b.size, b_effective_k, ab_base2k))

    def glwe_tensor_square_apply(self, cnv_offset, res, res_base2k, a, a_effective_k, a_base2k):
        rank = a.cols - 1
        assert res.cols == (rank + 1) * (rank + 2) // 2
        self.lib.pzr_glwe_tensor_square_apply(self.t, *_sz(rank, cnv_offset), _p(res.data), *_sz(res.size, res_base2k), _p(a.data),
                                              *_sz(a.size, a_effective_k, a_base2k))

    def glwe_tensor_relinearize(self, res, res_base2k, a, a_base2k, tsk_pmat, dsize, key_base2k):
        rank = res.cols - 1
        assert a.cols == (rank + 1) * (rank + 2) // 2
        self.lib.pzr_glwe_tensor_relinearize(self.t, c_size_t(rank), _p(res.data), *_sz(res.size, res_base2k), _p(a.data),
                                             *_sz(a.size, a_base2k), _p(tsk_pmat.data), *_sz(tsk_pmat.rows, tsk_pmat.size, dsize, key_base2k))

    # ---- LWE glue of the gate bootstrap (mod_switch_2n, sample extract, LWE key switch, LWE <-> GLWE) ----
    # an LWE is a numpy int64 array of shape (size, n_lwe + 1): limb i = [b, a_0, ..., a_{n_lwe-1}]  (VecZnx(n_lwe + 1, 1, size))
    def mod_switch_2n(self, n2, lwe, base2k, negate=False):
        lwe = np.ascontiguousarray(lwe, dtype=np.int64)
        size, length = lwe.shape
        res = np.zeros(length, dtype=np.int64)
        self.lib.pzr_mod_switch_2n(c_size_t(n2), _p(res), _p(lwe), *_sz(length - 1, size, base2k), C.c_int(1 if negate else 0))
        return res

    def lwe_sample_extract(self, res_n_lwe, res_size, a):
        res = np.zeros((res_size, res_n_lwe + 1), dtype=np.int64)
        self.lib.pzr_lwe_sample_extract(c_size_t(self._n), _p(res), *_sz(res_n_lwe, res_size), _p(a.data), *_sz(a.cols, a.size))
        return res

    def lwe_keyswitch(self, res_n_lwe, res_size, res_base2k, a, a_base2k, pmat, dsize, key_base2k):
        a = np.ascontiguousarray(a, dtype=np.int64)
        res = np.zeros((res_size, res_n_lwe + 1), dtype=np.int64)
        self.lib.pzr_lwe_keyswitch(self.t, _p(res), *_sz(res_n_lwe, res_size, res_base2k), _p(a),
                                   *_sz(a.shape[1] - 1, a.shape[0], a_base2k), _p(pmat.data), *_sz(pmat.rows, pmat.size, dsize, key_base2k))
        return res

    def glwe_from_lwe(self, res, res_base2k, lwe, lwe_base2k, glwe_size, pmat, dsize, key_base2k):
        lwe = np.ascontiguousarray(lwe, dtype=np.int64)
        self.lib.pzr_glwe_from_lwe(self.t, c_size_t(res.cols - 1), _p(res.data), *_sz(res.size, res_base2k), _p(lwe),
                                   *_sz(lwe.shape[1] - 1, lwe.shape[0], lwe_base2k, glwe_size), _p(pmat.data),
                                   *_sz(pmat.rows, pmat.size, dsize, key_base2k))

    def lwe_from_glwe(self, res_n_lwe, res_size, res_base2k, a, a_base2k, a_idx, pmat, dsize, key_base2k):
        res = np.zeros((res_size, res_n_lwe + 1), dtype=np.int64)
        self.lib.pzr_lwe_from_glwe(self.t, c_size_t(a.cols - 1), _p(res), *_sz(res_n_lwe, res_size, res_base2k), _p(a.data),
                                   *_sz(a.size, a_base2k, a_idx), _p(pmat.data), *_sz(pmat.rows, pmat.size, dsize, key_base2k))
        return res
