"""ctypes binding of libpoulpy_hip.so + a ``Module`` with poulpy-hal's method names.

Method names / argument order follow the api traits of the reference
(poulpy-hal/src/api/{vec_znx_dft,svp_ppol,vmp_pmat,vec_znx_big}.rs), minus the
``scratch`` argument (the device path keeps its own workspace; the `*_tmp_bytes`
functions still return the reference's sizes).  A non-zero status from the C ABI raises
``PoulpyHipError`` — the Rust shim panics in the same places.

This module NEVER falls back to a CPU implementation: without the shared library or
without a HIP device every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, c_double, c_int, c_int64, c_size_t, c_uint64, c_void_p

import numpy as np

from .layouts import CnvPVecL, CnvPVecR, MatZnx, ScalarZnx, SvpPPol, VecZnx, VecZnxBig, VecZnxDft, VmpPMat

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpoulpy_hip.so")


class PoulpyHipError(RuntimeError):
    pass


class GlweOpParams(C.Structure):
    """pz_glwe_op_params (include/poulpy_hip.h)"""
    _fields_ = [(k, c_uint64) for k in (
        "rank", "dnum", "dsize", "key_size", "key_base2k", "a_size", "a_base2k", "res_size", "res_base2k", "rank_out")]


class BlindRotationParams(C.Structure):
    """pz_blind_rotation_params (include/poulpy_hip.h)"""
    _fields_ = [(k, c_uint64) for k in ("rank", "n_lwe", "block_size", "dnum", "brk_size", "base2k", "res_size", "lut_size")]


class CircuitBootstrappingParams(C.Structure):
    """pz_circuit_bootstrapping_params (include/poulpy_hip.h)"""
    _fields_ = [("br", BlindRotationParams)] + [(k, c_uint64) for k in ("atk_dnum", "atk_size", "tsk_dnum", "tsk_size", "res_dnum",
                                                                         "res_size", "gap", "extension_factor", "atk_base2k",
                                                                         "tsk_base2k", "res_base2k", "atk_glwe_size", "trace_size")]


class GlweTensorParams(C.Structure):
    """pz_glwe_tensor_params (include/poulpy_hip.h)"""
    _fields_ = [(k, c_uint64) for k in ("rank", "a_size", "b_size", "ab_base2k", "a_effective_k", "b_effective_k", "res_size",
                                        "res_base2k", "cnv_offset")]


_lib = None
PZ_ABI_VERSION = 4   # pz_abi_version() of include/poulpy_hip.h this mirror was written against


def load_library(path: str | None = None) -> C.CDLL:
    """Load libpoulpy_hip.so (built by ``__graft_entry__.build()``); raises if missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("POULPY_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise PoulpyHipError(
            f"{p} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback.")
    lib = C.CDLL(p)
    lib.pz_last_error.restype = C.c_char_p
    lib.pz_abi_version.restype = C.c_uint32
    if lib.pz_abi_version() != PZ_ABI_VERSION:   # struct layouts and entry points of this mirror belong to ONE generation of the header
        raise PoulpyHipError(f"{p} has ABI version {lib.pz_abi_version()}, poulpy_amd/hal.py mirrors version {PZ_ABI_VERSION} "
                             "(a stale or variant build: rebuild with __graft_entry__.build())")
    lib.pz_module_n.restype = c_uint64
    lib.pz_alloc_bytes.restype = c_void_p
    lib.pz_alloc_bytes.argtypes = [c_size_t]
    lib.pz_free_bytes.argtypes = [c_void_p]
    lib.pz_module_stream.restype = c_void_p
    lib.pz_module_graph_launches.restype = c_uint64
    for name in ("pz_bytes_of_vec_znx", "pz_bytes_of_vec_znx_dft", "pz_bytes_of_vec_znx_big", "pz_bytes_of_svp_ppol",
                 "pz_bytes_of_vmp_pmat", "pz_vec_znx_idft_apply_tmp_bytes", "pz_vmp_prepare_tmp_bytes",
                 "pz_vmp_apply_dft_tmp_bytes", "pz_vmp_apply_dft_to_dft_tmp_bytes", "pz_vec_znx_big_normalize_tmp_bytes",
                 "pz_glwe_op_workspace_bytes", "pz_vec_znx_automorphism_assign_tmp_bytes",
                 "pz_vec_znx_big_automorphism_assign_tmp_bytes", "pz_blind_rotation_workspace_bytes", "pz_vec_znx_rsh_tmp_bytes", "pz_vec_znx_rotate_assign_tmp_bytes",
                 "pz_circuit_bootstrapping_tmp_bytes", "pz_vec_znx_normalize_tmp_bytes", "pz_vec_znx_lsh_tmp_bytes", "pz_glwe_pack_tmp_bytes", "pz_glwe_pack_bases_tmp_bytes",
                 "pz_circuit_bootstrapping_to_exponent_tmp_bytes", "pz_blind_rotation_extended_tmp_bytes",
                 "pz_cnv_prepare_left_tmp_bytes", "pz_cnv_prepare_right_tmp_bytes", "pz_cnv_prepare_self_tmp_bytes", "pz_cnv_apply_dft_tmp_bytes",
                 "pz_cnv_pairwise_apply_dft_tmp_bytes", "pz_cnv_by_const_apply_tmp_bytes", "pz_glwe_tensor_apply_workspace_bytes",
                 "pz_comm_unique_id_bytes"):
        getattr(lib, name).restype = c_size_t
    if path is None:
        _lib = lib
    return lib


def _p(arr: np.ndarray):
    return arr.ctypes.data_as(c_void_p)


def _sz(*xs):
    return [c_size_t(int(x)) for x in xs]


class DeviceBuffer:
    """A raw HBM allocation owned by a Module (pz_device_alloc)."""

    def __init__(self, module: "Module", nbytes: int):
        self.module, self.nbytes = module, int(nbytes)
        out = c_void_p()
        module._ck(module.lib.pz_device_alloc(module.handle, c_size_t(self.nbytes), C.byref(out)))
        self.ptr = out

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.module._ck(self.module.lib.pz_memcpy_h2d(self.module.handle, self.ptr, _p(arr), c_size_t(arr.nbytes)))
        return self

    def download(self, dtype, count: int, offset_bytes: int = 0) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        src = c_void_p(self.ptr.value + offset_bytes)
        self.module._ck(self.module.lib.pz_memcpy_d2h(self.module.handle, _p(out), src, c_size_t(out.nbytes)))
        return out

    def at(self, offset_bytes: int) -> c_void_p:
        return c_void_p(self.ptr.value + int(offset_bytes))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self.module.lib.pz_device_free(self.module.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Module:
    """``Module<FFT64Hip>`` — poulpy-hal/src/layouts/module.rs:97-189."""

    def __init__(self, n: int, device: int | None = None, lib: C.CDLL | None = None):
        self.lib = lib or load_library()
        h = c_void_p()
        if device is None:
            st = self.lib.pz_module_new(c_uint64(n), C.byref(h))
        else:
            st = self.lib.pz_module_new_on_device(c_uint64(n), c_int(device), C.byref(h))
        self.handle = h
        self._n = int(n)
        if st != 0:
            self.handle = None
            raise PoulpyHipError(f"pz_module_new({n}) failed [{st}]: {self.lib.pz_last_error().decode()}")

    # -- plumbing -------------------------------------------------------------
    def _ck(self, st: int):
        if st != 0:
            raise PoulpyHipError(f"[{st}] {self.lib.pz_last_error().decode()}")

    def n(self) -> int:
        return self._n

    def sync(self):
        self._ck(self.lib.pz_module_sync(self.handle))

    def clone(self) -> "Module":
        """A sibling for another host thread (pz_module_clone): shares the device tables, owns its stream / workspaces / lock."""
        sib = object.__new__(Module)
        sib.lib, sib._n = self.lib, self._n
        h = c_void_p()
        st = self.lib.pz_module_clone(self.handle, C.byref(h))
        sib.handle = h if st == 0 else None
        if st != 0:
            raise PoulpyHipError(f"pz_module_clone failed [{st}]: {self.lib.pz_last_error().decode()}")
        return sib

    def close(self):
        if self.handle is not None:
            self.lib.pz_module_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def set_chunk(self, cts: int):
        self._ck(self.lib.pz_module_set_chunk(self.handle, c_size_t(cts)))

    def set_fusion(self, fuse_tail: bool = True, fuse_mid: bool = True):
        self._ck(self.lib.pz_module_set_fusion(self.handle, c_int(int(fuse_tail)), c_int(int(fuse_mid))))

    def set_small_path(self, enable: bool = True):
        """N = 4096: the two-kernel pipeline (device_small.hpp) on / off (on by default)."""
        self._ck(self.lib.pz_module_set_small_path(self.handle, c_int(1 if enable else 0)))

    def set_margin_probe(self, enable: bool):
        self._ck(self.lib.pz_module_set_margin_probe(self.handle, c_int(1 if enable else 0)))

    def get_margin(self) -> float:
        out = c_double()
        self._ck(self.lib.pz_module_get_margin(self.handle, C.byref(out)))
        return out.value

    def rounding_margin_of(self, run) -> float:
        """max |x - round(x)| over every value the inverse transforms of `run()` round (0.5 = a wrong limb): one more execution of `run`
        with the probing instantiations of the rounding kernels in the same dispatch (pz_module_set_margin_probe)."""
        self.sync()
        self.set_margin_probe(True)
        try:
            run()
            self.sync()
            return self.get_margin()
        finally:
            self.set_margin_probe(False)

    KERNEL_CLASSES = ("fwd_pass1", "fwd_pass2", "vmp", "inv_pass2", "inv_pass1", "normalize", "elementwise", "fused_mid",
                      "fused_tail")

    def dispatch_notes(self, reset: bool = False) -> str:
        """Kernel instantiations chosen by the hot dispatch sites since the last reset (include/poulpy_hip.h)."""
        buf = C.create_string_buffer(4096)
        self._ck(self.lib.pz_module_dispatch_notes(self.handle, buf, c_size_t(4096), c_int(1 if reset else 0)))
        return buf.value.decode()

    def set_kernel_timing(self, enable: bool):
        self._ck(self.lib.pz_module_set_kernel_timing(self.handle, c_int(1 if enable else 0)))

    def kernel_stats(self) -> dict:
        """{class name: (launches, total_ms)} measured with HIP events on the module stream."""
        out = {}
        for k, name in enumerate(self.KERNEL_CLASSES):
            cnt, ms = c_uint64(), c_double()
            self._ck(self.lib.pz_module_get_kernel_stats(self.handle, c_int(k), C.byref(cnt), C.byref(ms)))
            out[name] = (cnt.value, ms.value)
        return out

    # -- allocation (api/*Alloc traits) -----------------------------------------
    def vec_znx_dft_alloc(self, cols, size) -> VecZnxDft:
        return VecZnxDft(self._n, cols, size)

    def vec_znx_big_alloc(self, cols, size) -> VecZnxBig:
        return VecZnxBig(self._n, cols, size)

    def svp_ppol_alloc(self, cols) -> SvpPPol:
        return SvpPPol(self._n, cols)

    def vmp_pmat_alloc(self, rows, cols_in, cols_out, size) -> VmpPMat:
        return VmpPMat(self._n, rows, cols_in, cols_out, size)

    def bytes_of_vec_znx_dft(self, cols, size) -> int:
        return self.lib.pz_bytes_of_vec_znx_dft(c_uint64(self._n), *_sz(cols, size))

    def bytes_of_vmp_pmat(self, rows, cols_in, cols_out, size) -> int:
        return self.lib.pz_bytes_of_vmp_pmat(c_uint64(self._n), *_sz(rows, cols_in, cols_out, size))

    # -- VecZnxDft (api/vec_znx_dft.rs) -------------------------------------------
    def vec_znx_dft_apply(self, step, offset, res: VecZnxDft, res_col, a: VecZnx, a_col):
        self._ck(self.lib.pz_vec_znx_dft_apply(self.handle, *_sz(step, offset), _p(res.data), *_sz(res.cols, res.size, res_col),
                                               _p(a.data), *_sz(a.cols, a.size, a_col)))

    def vec_znx_idft_apply_tmp_bytes(self) -> int:
        return self.lib.pz_vec_znx_idft_apply_tmp_bytes(self.handle)

    def vec_znx_idft_apply(self, res: VecZnxBig, res_col, a: VecZnxDft, a_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_idft_apply(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                *_sz(a.cols, a.size, a_col)))

    def vec_znx_idft_apply_tmpa(self, res: VecZnxBig, res_col, a: VecZnxDft, a_col):
        self._ck(self.lib.pz_vec_znx_idft_apply_tmpa(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                     *_sz(a.cols, a.size, a_col)))

    def vec_znx_idft_apply_consume(self, a: VecZnxDft) -> VecZnxBig:
        self._ck(self.lib.pz_vec_znx_idft_apply_consume(self.handle, _p(a.data), *_sz(a.cols, a.size)))
        return a.into_big()

    def _dft3(self, fn, res, res_col, a, a_col, b, b_col):
        self._ck(fn(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a.size, a_col),
                    _p(b.data), *_sz(b.cols, b.size, b_col)))

    def _dft2(self, fn, res, res_col, a, a_col, *extra):
        self._ck(fn(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a.size, a_col), *extra))

    def vec_znx_dft_add_into(self, res, res_col, a, a_col, b, b_col):
        self._dft3(self.lib.pz_vec_znx_dft_add_into, res, res_col, a, a_col, b, b_col)

    def vec_znx_dft_sub(self, res, res_col, a, a_col, b, b_col):
        self._dft3(self.lib.pz_vec_znx_dft_sub, res, res_col, a, a_col, b, b_col)

    def vec_znx_dft_add_assign(self, res, res_col, a, a_col):
        self._dft2(self.lib.pz_vec_znx_dft_add_assign, res, res_col, a, a_col)

    def vec_znx_dft_add_scaled_assign(self, res, res_col, a, a_col, a_scale):
        self._dft2(self.lib.pz_vec_znx_dft_add_scaled_assign, res, res_col, a, a_col, c_int64(a_scale))

    def vec_znx_dft_sub_assign(self, res, res_col, a, a_col):
        self._dft2(self.lib.pz_vec_znx_dft_sub_assign, res, res_col, a, a_col)

    def vec_znx_dft_sub_negate_assign(self, res, res_col, a, a_col):
        self._dft2(self.lib.pz_vec_znx_dft_sub_negate_assign, res, res_col, a, a_col)

    def vec_znx_dft_copy(self, step, offset, res, res_col, a, a_col):
        self._ck(self.lib.pz_vec_znx_dft_copy(self.handle, *_sz(step, offset), _p(res.data), *_sz(res.cols, res.size, res_col),
                                              _p(a.data), *_sz(a.cols, a.size, a_col)))

    def vec_znx_dft_zero(self, res, res_col):
        self._ck(self.lib.pz_vec_znx_dft_zero(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col)))

    # -- SVP (api/svp_ppol.rs) -------------------------------------------------------
    def svp_prepare(self, res: SvpPPol, res_col, a: ScalarZnx, a_col):
        self._ck(self.lib.pz_svp_prepare(self.handle, _p(res.data), *_sz(res.cols, res_col), _p(a.data), *_sz(a.cols, a_col)))

    def svp_apply_dft(self, res: VecZnxDft, res_col, a: SvpPPol, a_col, b: VecZnx, b_col):
        self._ck(self.lib.pz_svp_apply_dft(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                           *_sz(a.cols, a_col), _p(b.data), *_sz(b.cols, b.size, b_col)))

    def svp_apply_dft_to_dft(self, res: VecZnxDft, res_col, a: SvpPPol, a_col, b: VecZnxDft, b_col):
        self._ck(self.lib.pz_svp_apply_dft_to_dft(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                  *_sz(a.cols, a_col), _p(b.data), *_sz(b.cols, b.size, b_col)))

    def svp_apply_dft_to_dft_assign(self, res: VecZnxDft, res_col, a: SvpPPol, a_col):
        self._ck(self.lib.pz_svp_apply_dft_to_dft_assign(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                         *_sz(a.cols, a_col)))

    # -- VMP (api/vmp_pmat.rs) -------------------------------------------------------
    def vmp_prepare_tmp_bytes(self, rows, cols_in, cols_out, size) -> int:
        return self.lib.pz_vmp_prepare_tmp_bytes(self.handle, *_sz(rows, cols_in, cols_out, size))

    def vmp_prepare(self, res: VmpPMat, a: MatZnx, scratch=None):
        assert (res.rows, res.cols_in, res.cols_out, res.size) == (a.rows, a.cols_in, a.cols_out, a.size)
        self._ck(self.lib.pz_vmp_prepare(self.handle, _p(res.data), _p(a.data), *_sz(a.rows, a.cols_in, a.cols_out, a.size)))

    def vmp_apply_dft_tmp_bytes(self, res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size) -> int:
        return self.lib.pz_vmp_apply_dft_tmp_bytes(self.handle, *_sz(res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size))

    def vmp_apply_dft(self, res: VecZnxDft, a: VecZnx, b: VmpPMat, scratch=None):
        self._ck(self.lib.pz_vmp_apply_dft(self.handle, _p(res.data), *_sz(res.cols, res.size), _p(a.data), *_sz(a.cols, a.size),
                                           _p(b.data), *_sz(b.rows, b.cols_in, b.cols_out, b.size)))

    def vmp_apply_dft_to_dft_tmp_bytes(self, res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size) -> int:
        return self.lib.pz_vmp_apply_dft_to_dft_tmp_bytes(self.handle, *_sz(res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size))

    def vmp_apply_dft_to_dft(self, res: VecZnxDft, a: VecZnxDft, b: VmpPMat, limb_offset=0, scratch=None):
        self._ck(self.lib.pz_vmp_apply_dft_to_dft(self.handle, _p(res.data), *_sz(res.cols, res.size), _p(a.data),
                                                  *_sz(a.cols, a.size), _p(b.data), *_sz(b.rows, b.cols_in, b.cols_out, b.size),
                                                  c_size_t(limb_offset)))

    def vmp_zero(self, res: VmpPMat):
        self._ck(self.lib.pz_vmp_zero(self.handle, _p(res.data), *_sz(res.rows, res.cols_in, res.cols_out, res.size)))

    # -- VecZnxBig (api/vec_znx_big.rs) ------------------------------------------------
    def vec_znx_big_normalize_tmp_bytes(self) -> int:
        return self.lib.pz_vec_znx_big_normalize_tmp_bytes(self.handle)

    def vec_znx_big_normalize(self, res: VecZnx, res_base2k, res_offset, res_col, a: VecZnxBig, a_base2k, a_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_big_normalize(self.handle, _p(res.data), *_sz(res.cols, res.size, res_base2k), c_int64(res_offset),
                                                   c_size_t(res_col), _p(a.data), *_sz(a.cols, a.size, a_base2k, a_col)))

    def vec_znx_big_add_small_assign(self, res: VecZnxBig, res_col, a: VecZnx, a_col):
        self._ck(self.lib.pz_vec_znx_big_add_small_assign(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                          *_sz(a.cols, a.size, a_col)))

    # -- i64 VecZnx limb-wise family (hal_impl.rs:34-131, :289) ---------------------------------------
    def _znx3(self, fn, res, res_col, a, a_col, b, b_col):
        self._ck(fn(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a.size, a_col), _p(b.data),
                    *_sz(b.cols, b.size, b_col)))

    def _znx2(self, fn, res, res_col, a, a_col):
        self._ck(fn(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data), *_sz(a.cols, a.size, a_col)))

    def vec_znx_add_into(self, res: VecZnx, res_col, a: VecZnx, a_col, b: VecZnx, b_col):
        self._znx3(self.lib.pz_vec_znx_add_into, res, res_col, a, a_col, b, b_col)

    def vec_znx_sub(self, res: VecZnx, res_col, a: VecZnx, a_col, b: VecZnx, b_col):
        self._znx3(self.lib.pz_vec_znx_sub, res, res_col, a, a_col, b, b_col)

    def vec_znx_add_assign(self, res: VecZnx, res_col, a: VecZnx, a_col):
        self._znx2(self.lib.pz_vec_znx_add_assign, res, res_col, a, a_col)

    def vec_znx_sub_assign(self, res: VecZnx, res_col, a: VecZnx, a_col):
        self._znx2(self.lib.pz_vec_znx_sub_assign, res, res_col, a, a_col)

    def vec_znx_sub_negate_assign(self, res: VecZnx, res_col, a: VecZnx, a_col):
        self._znx2(self.lib.pz_vec_znx_sub_negate_assign, res, res_col, a, a_col)

    def vec_znx_negate(self, res: VecZnx, res_col, a: VecZnx, a_col):
        self._znx2(self.lib.pz_vec_znx_negate, res, res_col, a, a_col)

    def vec_znx_copy(self, res: VecZnx, res_col, a: VecZnx, a_col):
        self._znx2(self.lib.pz_vec_znx_copy, res, res_col, a, a_col)

    def vec_znx_negate_assign(self, res: VecZnx, res_col):
        self._ck(self.lib.pz_vec_znx_negate_assign(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col)))

    def vec_znx_zero(self, res: VecZnx, res_col):
        self._ck(self.lib.pz_vec_znx_zero(self.handle, _p(res.data), *_sz(res.cols, res.size, res_col)))

    def vec_znx_normalize_tmp_bytes(self) -> int:
        return self.lib.pz_vec_znx_normalize_tmp_bytes(self.handle)

    def vec_znx_normalize(self, res: VecZnx, res_base2k, res_offset, res_col, a: VecZnx, a_base2k, a_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_normalize(self.handle, _p(res.data), *_sz(res.cols, res.size, res_base2k), c_int64(res_offset),
                                               c_size_t(res_col), _p(a.data), *_sz(a.cols, a.size, a_base2k, a_col)))

    def vec_znx_normalize_assign(self, base2k, res: VecZnx, res_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_normalize_assign(self.handle, c_size_t(base2k), _p(res.data), *_sz(res.cols, res.size, res_col)))

    def vec_znx_lsh(self, base2k, k, res: VecZnx, res_col, a: VecZnx, a_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_lsh(self.handle, *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                         *_sz(a.cols, a.size, a_col)))

    def vec_znx_rsh(self, base2k, k, res: VecZnx, res_col, a: VecZnx, a_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_rsh(self.handle, *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                         *_sz(a.cols, a.size, a_col)))

    def vec_znx_lsh_assign(self, base2k, k, res: VecZnx, res_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_lsh_assign(self.handle, *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col)))

    # -- X -> X^p on i64 containers (hal_impl.rs:236-243, :517-524) ----------------------------
    def vec_znx_automorphism(self, p: int, res: VecZnx, res_col, a: VecZnx, a_col):
        self._ck(self.lib.pz_vec_znx_automorphism(self.handle, c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                  *_sz(a.cols, a.size, a_col)))

    def vec_znx_automorphism_assign(self, p: int, res: VecZnx, res_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_automorphism_assign(self.handle, c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col)))

    def vec_znx_automorphism_assign_tmp_bytes(self) -> int:
        return self.lib.pz_vec_znx_automorphism_assign_tmp_bytes(self.handle)

    def vec_znx_big_automorphism(self, p: int, res: VecZnxBig, res_col, a: VecZnxBig, a_col):
        self._ck(self.lib.pz_vec_znx_big_automorphism(self.handle, c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col),
                                                      _p(a.data), *_sz(a.cols, a.size, a_col)))

    def vec_znx_big_automorphism_assign(self, p: int, res: VecZnxBig, res_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_big_automorphism_assign(self.handle, c_int64(p), _p(res.data), *_sz(res.cols, res.size, res_col)))

    def vec_znx_big_automorphism_assign_tmp_bytes(self) -> int:
        return self.lib.pz_vec_znx_big_automorphism_assign_tmp_bytes(self.handle)

    def vec_znx_rotate(self, k: int, res: VecZnx, res_col, a: VecZnx, a_col):
        """hal_impl.rs:225: res = X^k * a."""
        self._ck(self.lib.pz_vec_znx_rotate(self.handle, c_int64(k), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                            *_sz(a.cols, a.size, a_col)))

    def vec_znx_rotate_assign(self, k: int, res: VecZnx, res_col, scratch=None):
        self._ck(self.lib.pz_vec_znx_rotate_assign(self.handle, c_int64(k), _p(res.data), *_sz(res.cols, res.size, res_col)))

    def vec_znx_rsh_assign(self, base2k: int, k: int, res: VecZnx, res_col, scratch=None):
        """hal_impl.rs:217 (reference/vec_znx/shift.rs:186-243)."""
        self._ck(self.lib.pz_vec_znx_rsh_assign(self.handle, *_sz(base2k, k), _p(res.data), *_sz(res.cols, res.size, res_col)))

    def glwe_trace_batched(self, res: c_void_p, gals, key_ptrs, params: GlweOpParams, batch: int):
        """poulpy-core glwe_trace.rs:129-176 on device-resident ciphertexts: gals[s] / key_ptrs[s] (device pointers) per step."""
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ptrs = (c_void_p * ns)(*[p.value if isinstance(p, c_void_p) else int(p) for p in key_ptrs])
        self._ck(self.lib.pz_glwe_trace_batched(self.handle, res, c_size_t(ns), g, ptrs, C.byref(params), c_size_t(batch)))

    # -- batched device-resident GLWE ops (CoreImpl overrides) ---------------------------
    AUTO_MODES = {"automorphism": 0, "add": 1, "sub": 2, "sub_negate": 3}

    def glwe_automorphism_batched(self, res: c_void_p, a: c_void_p, key_pmat: c_void_p, params: GlweOpParams, gal: int, mode, batch: int):
        """poulpy-core automorphism/glwe_ct.rs:51-275; mode: "automorphism" | "add" | "sub" | "sub_negate"."""
        mode = self.AUTO_MODES[mode] if isinstance(mode, str) else int(mode)
        self._ck(self.lib.pz_glwe_automorphism_batched(self.handle, res, a, key_pmat, C.byref(params), c_int64(gal), c_int(mode),
                                                       c_size_t(batch)))

    def ggsw_external_product(self, res: c_void_p, a: c_void_p, a_dnum: int, ggsw_pmat: c_void_p, params: GlweOpParams):
        """poulpy-core external_product/ggsw.rs:54-58 on a device-resident GGSW (MatZnx layout)."""
        self._ck(self.lib.pz_ggsw_external_product(self.handle, res, a, c_size_t(a_dnum), ggsw_pmat, C.byref(params)))

    def ggsw_from_gglwe_batched(self, ggsw: c_void_p, a: c_void_p, a_cols_in: int, dnum: int, tsk_pmats, params: GlweOpParams, count: int = 1):
        """conversion/gglwe_to_ggsw.rs:32-61 on `count` contiguous device GGLWEs -> GGSWs."""
        arr = (c_void_p * len(tsk_pmats))(*[k.value if isinstance(k, c_void_p) else int(k) for k in tsk_pmats])
        self._ck(self.lib.pz_ggsw_from_gglwe_batched(self.handle, ggsw, a, *_sz(a_cols_in, dnum), arr, C.byref(params), c_size_t(count)))

    def ggsw_expand_row_batched(self, ggsw: c_void_p, dnum: int, tsk_pmats, params: GlweOpParams, count: int = 1):
        """conversion/gglwe_to_ggsw.rs:116-268 on `count` contiguous device GGSWs, in place; tsk_pmats: rank device pointers."""
        arr = (c_void_p * len(tsk_pmats))(*[k.value if isinstance(k, c_void_p) else int(k) for k in tsk_pmats])
        self._ck(self.lib.pz_ggsw_expand_row_batched(self.handle, ggsw, c_size_t(dnum), arr, C.byref(params), c_size_t(count)))

    def glwe_external_product_batched(self, res: c_void_p, a: c_void_p, ggsw_pmat: c_void_p, params: GlweOpParams, batch: int):
        self._ck(self.lib.pz_glwe_external_product_batched(self.handle, res, a, ggsw_pmat, C.byref(params), c_size_t(batch)))

    def glwe_keyswitch_batched(self, res: c_void_p, a: c_void_p, key_pmat: c_void_p, params: GlweOpParams, batch: int):
        self._ck(self.lib.pz_glwe_keyswitch_batched(self.handle, res, a, key_pmat, C.byref(params), c_size_t(batch)))

    def blind_rotation_execute_batched(self, res: c_void_p, lwe_2n: c_void_p, lut: c_void_p, brk: c_void_p, params: BlindRotationParams,
                                       batch: int):
        """poulpy-bin-fhe blind_rotation/algorithms/cggi/algorithm.rs:76-118,265-440 on a batch of mod-switched LWE ciphertexts."""
        self._ck(self.lib.pz_blind_rotation_execute_batched(self.handle, res, lwe_2n, lut, brk, C.byref(params), c_size_t(batch)))

    def glwe_pack_tmp_bytes(self, params: GlweOpParams, batch: int) -> int:
        return self.lib.pz_glwe_pack_tmp_bytes(self.handle, C.byref(params), c_size_t(batch))

    def glwe_pack_batched(self, res: c_void_p, indices, ct_ptrs, log_gap_out: int, gals, key_ptrs, params: GlweOpParams, tmp: c_void_p,
                          tmp_bytes: int, batch: int):
        """poulpy-core glwe_packing.rs:122-176 on `batch` problems: ct_ptrs[s] -> batch contiguous device GLWEs of index indices[s]."""
        ns = len(indices)
        idx = (c_uint64 * ns)(*[int(i) for i in indices])
        cp = (c_void_p * ns)(*[p.value if isinstance(p, c_void_p) else int(p) for p in ct_ptrs])
        ng = len(gals)
        g = (c_int64 * ng)(*[int(x) for x in gals])
        kp = (c_void_p * ng)(*[p.value if isinstance(p, c_void_p) else int(p) for p in key_ptrs])
        self._ck(self.lib.pz_glwe_pack_batched(self.handle, res, c_size_t(ns), idx, cp, c_size_t(log_gap_out), g, kp, C.byref(params), tmp,
                                               c_size_t(tmp_bytes), c_size_t(batch)))

    def glwe_pack_bases_tmp_bytes(self, params: GlweOpParams, trace_size: int, batch: int) -> int:
        return self.lib.pz_glwe_pack_bases_tmp_bytes(self.handle, C.byref(params), c_size_t(trace_size), c_size_t(batch))

    def glwe_pack_bases_batched(self, res: c_void_p, indices, ct_ptrs, log_gap_out: int, gals, key_ptrs, params: GlweOpParams,
                                trace_size: int, tmp: c_void_p, tmp_bytes: int, batch: int):
        """glwe_pack with the automorphism keys in their own base (poulpy-core test_suite/glwe_packing.rs:40-42)."""
        ns = len(indices)
        idx = (c_uint64 * ns)(*[int(i) for i in indices])
        cp = (c_void_p * ns)(*[p.value if isinstance(p, c_void_p) else int(p) for p in ct_ptrs])
        ng = len(gals)
        g = (c_int64 * ng)(*[int(x) for x in gals])
        kp = (c_void_p * ng)(*[p.value if isinstance(p, c_void_p) else int(p) for p in key_ptrs])
        self._ck(self.lib.pz_glwe_pack_bases_batched(self.handle, res, c_size_t(ns), idx, cp, c_size_t(log_gap_out), g, kp, C.byref(params),
                                                     c_size_t(trace_size), tmp, c_size_t(tmp_bytes), c_size_t(batch)))

    def set_graphs(self, enable: bool):
        """HIP-graph replay of the launch-bound composite calls (blind rotation, trace, circuit bootstrapping); on by default."""
        self._ck(self.lib.pz_module_set_graphs(self.handle, C.c_int(1 if enable else 0)))

    def graph_launches(self) -> int:
        return int(self.lib.pz_module_graph_launches(self.handle))

    def circuit_bootstrapping_tmp_bytes(self, params: CircuitBootstrappingParams, batch: int) -> int:
        return self.lib.pz_circuit_bootstrapping_tmp_bytes(self.handle, C.byref(params), c_size_t(batch))

    def circuit_bootstrapping_execute_to_constant_batched(self, ggsw: c_void_p, lwe_2n: c_void_p, lut: c_void_p, brk: c_void_p, gals,
                                                          atk_ptrs, tsk_ptrs, params: CircuitBootstrappingParams, tmp: c_void_p,
                                                          tmp_bytes: int, batch: int):
        """poulpy-bin-fhe circuit_bootstrapping/circuit.rs:177-195 (core :219-370, constant mode, one base2k) on a batch of LWEs."""
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ap = (c_void_p * ns)(*[k.value if isinstance(k, c_void_p) else int(k) for k in atk_ptrs])
        tp = (c_void_p * len(tsk_ptrs))(*[k.value if isinstance(k, c_void_p) else int(k) for k in tsk_ptrs])
        self._ck(self.lib.pz_circuit_bootstrapping_execute_to_constant_batched(self.handle, ggsw, lwe_2n, lut, brk, c_size_t(ns), g, ap, tp,
                                                                               C.byref(params), tmp, c_size_t(tmp_bytes), c_size_t(batch)))

    def circuit_bootstrapping_to_exponent_tmp_bytes(self, params: CircuitBootstrappingParams, log_domain: int, batch: int) -> int:
        return self.lib.pz_circuit_bootstrapping_to_exponent_tmp_bytes(self.handle, C.byref(params), *_sz(log_domain, batch))

    def circuit_bootstrapping_execute_to_exponent_batched(self, ggsw: c_void_p, lwe_2n: c_void_p, lut: c_void_p, brk: c_void_p, gals,
                                                          atk_ptrs, tsk_ptrs, params: CircuitBootstrappingParams, log_gap_in: int,
                                                          log_gap_out: int, log_domain: int, tmp: c_void_p, tmp_bytes: int, batch: int):
        """circuit.rs:197-216 + post_process :373-421 (one base2k); gals / atk_ptrs: all log2(n) trace steps."""
        ns = len(gals)
        g = (c_int64 * ns)(*[int(x) for x in gals])
        ap = (c_void_p * ns)(*[k.value if isinstance(k, c_void_p) else int(k) for k in atk_ptrs])
        tp = (c_void_p * len(tsk_ptrs))(*[k.value if isinstance(k, c_void_p) else int(k) for k in tsk_ptrs])
        self._ck(self.lib.pz_circuit_bootstrapping_execute_to_exponent_batched(
            self.handle, ggsw, lwe_2n, lut, brk, g, ap, tp, C.byref(params), *_sz(log_gap_in, log_gap_out, log_domain), tmp,
            c_size_t(tmp_bytes), c_size_t(batch)))

    def blind_rotation_extended_tmp_bytes(self, params: BlindRotationParams, ext: int, batch: int) -> int:
        return self.lib.pz_blind_rotation_extended_tmp_bytes(self.handle, C.byref(params), *_sz(ext, batch))

    def blind_rotation_execute_extended_batched(self, res: c_void_p, lwe_2n: c_void_p, lut: c_void_p, brk: c_void_p,
                                                params: BlindRotationParams, ext: int, tmp: c_void_p, tmp_bytes: int, batch: int):
        """algorithm.rs:121-273 (extension_factor > 1): lut = ext contiguous VecZnx(1, lut_size), lwe_2n switched to 2*n*ext."""
        self._ck(self.lib.pz_blind_rotation_execute_extended_batched(self.handle, res, lwe_2n, lut, brk, C.byref(params), c_size_t(ext), tmp,
                                                                     c_size_t(tmp_bytes), c_size_t(batch)))

    def blind_rotation_workspace_bytes(self, params: BlindRotationParams, batch: int) -> int:
        return self.lib.pz_blind_rotation_workspace_bytes(self.handle, C.byref(params), c_size_t(batch))

    # -- batched primitives on device pointers (object b at ptr + b * len(object)) ----------------
    def vec_znx_dft_apply_batched(self, batch, step, offset, res: c_void_p, res_cols, res_size, res_col, a: c_void_p, a_cols, a_size, a_col):
        self._ck(self.lib.pz_vec_znx_dft_apply_batched(self.handle, *_sz(batch, step, offset), res, *_sz(res_cols, res_size, res_col), a,
                                                       *_sz(a_cols, a_size, a_col)))

    def vec_znx_idft_apply_consume_batched(self, batch, data: c_void_p, cols, size):
        self._ck(self.lib.pz_vec_znx_idft_apply_consume_batched(self.handle, c_size_t(batch), data, *_sz(cols, size)))

    def vmp_apply_dft_to_dft_batched(self, batch, res: c_void_p, res_cols, res_size, a: c_void_p, a_cols, a_size, pmat: c_void_p, rows, cols_in,
                                     cols_out, size, limb_offset=0):
        self._ck(self.lib.pz_vmp_apply_dft_to_dft_batched(self.handle, c_size_t(batch), res, *_sz(res_cols, res_size), a, *_sz(a_cols, a_size),
                                                          pmat, *_sz(rows, cols_in, cols_out, size, limb_offset)))

    def vec_znx_big_normalize_batched(self, batch, res: c_void_p, res_cols, res_size, res_base2k, res_offset, res_col, a: c_void_p, a_cols,
                                      a_size, a_base2k, a_col):
        self._ck(self.lib.pz_vec_znx_big_normalize_batched(self.handle, c_size_t(batch), res, *_sz(res_cols, res_size, res_base2k),
                                                           c_int64(res_offset), c_size_t(res_col), a, *_sz(a_cols, a_size, a_base2k, a_col)))

    # -- convolution family (api/convolution.rs; hal_impl.rs:670-754) --------------------------------------
    def cnv_pvec_left_alloc(self, cols, size) -> CnvPVecL:
        return CnvPVecL(self._n, cols, size)

    def cnv_pvec_right_alloc(self, cols, size) -> CnvPVecR:
        return CnvPVecR(self._n, cols, size)

    def cnv_prepare_left_tmp_bytes(self, res_size, a_size) -> int:
        return self.lib.pz_cnv_prepare_left_tmp_bytes(self.handle, *_sz(res_size, a_size))

    def cnv_prepare_right_tmp_bytes(self, res_size, a_size) -> int:
        return self.lib.pz_cnv_prepare_right_tmp_bytes(self.handle, *_sz(res_size, a_size))

    def cnv_prepare_self_tmp_bytes(self, res_size, a_size) -> int:
        return self.lib.pz_cnv_prepare_self_tmp_bytes(self.handle, *_sz(res_size, a_size))

    def cnv_apply_dft_tmp_bytes(self, cnv_offset, res_size, a_size, b_size) -> int:
        return self.lib.pz_cnv_apply_dft_tmp_bytes(self.handle, *_sz(cnv_offset, res_size, a_size, b_size))

    def cnv_pairwise_apply_dft_tmp_bytes(self, cnv_offset, res_size, a_size, b_size) -> int:
        return self.lib.pz_cnv_pairwise_apply_dft_tmp_bytes(self.handle, *_sz(cnv_offset, res_size, a_size, b_size))

    def cnv_by_const_apply_tmp_bytes(self, cnv_offset, res_size, a_size, b_size) -> int:
        return self.lib.pz_cnv_by_const_apply_tmp_bytes(self.handle, *_sz(cnv_offset, res_size, a_size, b_size))

    def cnv_prepare_left(self, res: CnvPVecL, a: VecZnx, mask: int = -1, scratch=None):
        self._ck(self.lib.pz_cnv_prepare_left(self.handle, _p(res.data), *_sz(res.cols, res.size), _p(a.data), *_sz(a.cols, a.size), c_int64(mask)))

    def cnv_prepare_right(self, res: CnvPVecR, a: VecZnx, mask: int = -1, scratch=None):
        self._ck(self.lib.pz_cnv_prepare_right(self.handle, _p(res.data), *_sz(res.cols, res.size), _p(a.data), *_sz(a.cols, a.size), c_int64(mask)))

    def cnv_prepare_self(self, left: CnvPVecL, right: CnvPVecR, a: VecZnx, mask: int = -1, scratch=None):
        assert (left.cols, left.size) == (right.cols, right.size)
        self._ck(self.lib.pz_cnv_prepare_self(self.handle, _p(left.data), _p(right.data), *_sz(left.cols, left.size), _p(a.data),
                                              *_sz(a.cols, a.size), c_int64(mask)))

    def cnv_apply_dft(self, cnv_offset, res: VecZnxDft, res_col, a: CnvPVecL, a_col, b: CnvPVecR, b_col, scratch=None):
        self._ck(self.lib.pz_cnv_apply_dft(self.handle, c_size_t(cnv_offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                           *_sz(a.cols, a.size, a_col), _p(b.data), *_sz(b.cols, b.size, b_col)))

    def cnv_pairwise_apply_dft(self, cnv_offset, res: VecZnxDft, res_col, a: CnvPVecL, b: CnvPVecR, i, j, scratch=None):
        self._ck(self.lib.pz_cnv_pairwise_apply_dft(self.handle, c_size_t(cnv_offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                    *_sz(a.cols, a.size), _p(b.data), *_sz(b.cols, b.size, i, j)))

    def cnv_by_const_apply(self, cnv_offset, res: VecZnxBig, res_col, a: VecZnx, a_col, b, scratch=None):
        b = np.ascontiguousarray(b, dtype=np.int64)
        self._ck(self.lib.pz_cnv_by_const_apply(self.handle, c_size_t(cnv_offset), _p(res.data), *_sz(res.cols, res.size, res_col), _p(a.data),
                                                *_sz(a.cols, a.size, a_col), _p(b), c_size_t(b.size)))

    TENSOR_MODES = {"apply": 0, "add_assign": 1, "square": 2}

    def glwe_tensor_apply_workspace_bytes(self, params: GlweTensorParams, mode, batch: int) -> int:
        mode = self.TENSOR_MODES[mode] if isinstance(mode, str) else int(mode)
        return self.lib.pz_glwe_tensor_apply_workspace_bytes(self.handle, C.byref(params), c_int(mode), c_size_t(batch))

    def glwe_tensor_apply_batched(self, res: c_void_p, a: c_void_p, b, params: GlweTensorParams, mode, batch: int):
        """poulpy-core operations/glwe.rs:609-913 on device-resident ciphertexts; mode: "apply" | "add_assign" | "square"."""
        mode = self.TENSOR_MODES[mode] if isinstance(mode, str) else int(mode)
        self._ck(self.lib.pz_glwe_tensor_apply_batched(self.handle, res, a, b if b is not None else a, C.byref(params), c_int(mode), c_size_t(batch)))

    def glwe_tensor_relinearize_batched(self, res: c_void_p, a: c_void_p, tsk_pmat: c_void_p, params: GlweOpParams, batch: int):
        """poulpy-core operations/glwe.rs:541-607 on device-resident GLWETensors sharing one prepared tensor key."""
        self._ck(self.lib.pz_glwe_tensor_relinearize_batched(self.handle, res, a, tsk_pmat, C.byref(params), c_size_t(batch)))

    def glwe_tensor_mul_relinearize_batched(self, res: c_void_p, a: c_void_p, b, tsk_pmat: c_void_p, tparams: GlweTensorParams, rparams: GlweOpParams,
                                            mode, batch: int):
        """glwe_tensor_apply / _square_apply + glwe_tensor_relinearize with the tensor in scratch (poulpy-ckks leveled/default/mul.rs:49-85, :131-170);
        mode: "apply" | "square"."""
        mode = self.TENSOR_MODES[mode] if isinstance(mode, str) else int(mode)
        self._ck(self.lib.pz_glwe_tensor_mul_relinearize_batched(self.handle, res, a, b if b is not None else a, tsk_pmat, C.byref(tparams),
                                                                 C.byref(rparams), c_int(mode), c_size_t(batch)))

    # -- LWE glue of the gate bootstrap (device-resident batches; an LWE = VecZnx(n_lwe + 1, 1, size)) -------------
    def lwe_mod_switch_2n_batched(self, res: c_void_p, lwe: c_void_p, n_lwe: int, lwe_size: int, base2k: int, n2: int, negate: bool, batch: int):
        """poulpy-bin-fhe blind_rotation/algorithms/mod.rs:136-176."""
        self._ck(self.lib.pz_lwe_mod_switch_2n_batched(self.handle, res, lwe, c_size_t(n_lwe), c_size_t(lwe_size), c_size_t(base2k), c_size_t(n2),
                                                       c_int(1 if negate else 0), c_size_t(batch)))

    def lwe_sample_extract_batched(self, res: c_void_p, res_n_lwe: int, res_size: int, a: c_void_p, a_cols: int, a_size: int, batch: int):
        """poulpy-core api/conversion.rs:15-40."""
        self._ck(self.lib.pz_lwe_sample_extract_batched(self.handle, res, c_size_t(res_n_lwe), c_size_t(res_size), a, c_size_t(a_cols),
                                                        c_size_t(a_size), c_size_t(batch)))

    def lwe_keyswitch_batched(self, res: c_void_p, res_n_lwe: int, a: c_void_p, a_n_lwe: int, ksk_pmat: c_void_p, params: GlweOpParams, batch: int):
        """poulpy-core keyswitching/lwe.rs:49-94."""
        self._ck(self.lib.pz_lwe_keyswitch_batched(self.handle, res, c_size_t(res_n_lwe), a, c_size_t(a_n_lwe), ksk_pmat, C.byref(params), c_size_t(batch)))

    def glwe_from_lwe_batched(self, res: c_void_p, lwe: c_void_p, n_lwe: int, lwe_size: int, lwe_base2k: int, ksk_pmat: c_void_p,
                              params: GlweOpParams, batch: int):
        """poulpy-core conversion/lwe_to_glwe.rs:46-121."""
        self._ck(self.lib.pz_glwe_from_lwe_batched(self.handle, res, lwe, c_size_t(n_lwe), c_size_t(lwe_size), c_size_t(lwe_base2k), ksk_pmat,
                                                   C.byref(params), c_size_t(batch)))

    def lwe_from_glwe_batched(self, res: c_void_p, res_n_lwe: int, a: c_void_p, a_idx: int, ksk_pmat: c_void_p, params: GlweOpParams, batch: int):
        """poulpy-core conversion/glwe_to_lwe.rs:42-90."""
        self._ck(self.lib.pz_lwe_from_glwe_batched(self.handle, res, c_size_t(res_n_lwe), a, c_size_t(a_idx), ksk_pmat, C.byref(params), c_size_t(batch)))

    # -- multi-GPU: RCCL broadcast of prepared keys on the module stream ---------------------
    def comm_available(self):
        """Raises unless RCCL can be loaded in this process (pz_comm_available: dlopen + symbols; draws no id, opens no socket)."""
        self._ck(self.lib.pz_comm_available())

    def comm_unique_id(self) -> bytes:
        """ncclGetUniqueId (call on ONE rank, ship the bytes to the others out of band)."""
        n = self.lib.pz_comm_unique_id_bytes()
        buf = C.create_string_buffer(n)
        self._ck(self.lib.pz_comm_unique_id(buf))
        return buf.raw

    def comm_init_rank(self, world_size: int, rank: int, unique_id: bytes):
        self._ck(self.lib.pz_comm_init_rank(self.handle, c_int(world_size), c_int(rank), C.c_char_p(unique_id)))

    def comm_destroy(self):
        self._ck(self.lib.pz_comm_destroy(self.handle))

    def bcast_key(self, dev_ptr: c_void_p, nbytes: int, root: int = 0):
        """In-place ncclBroadcast of a device buffer (a prepared key) from `root`, asynchronous on the module stream."""
        self._ck(self.lib.pz_bcast_key(self.handle, dev_ptr, c_size_t(nbytes), c_int(root)))

    def pin_key(self, pmat: c_void_p, rows: int, cols_in: int, cols_out: int, size: int):
        """Declare a prepared device key immutable: the fused pipeline keeps its row-sliced copy instead of rebuilding it per call."""
        self._ck(self.lib.pz_module_pin_key(self.handle, pmat, *_sz(rows, cols_in, cols_out, size)))

    def unpin_key(self, pmat: c_void_p):
        self._ck(self.lib.pz_module_unpin_key(self.handle, pmat))

    def glwe_op_workspace_bytes(self, params: GlweOpParams, batch: int, keyswitch: bool) -> int:
        return self.lib.pz_glwe_op_workspace_bytes(self.handle, C.byref(params), c_size_t(batch), c_int(int(keyswitch)))

    # events on the module stream
    def event_create(self) -> c_void_p:
        ev = c_void_p()
        self._ck(self.lib.pz_event_create(C.byref(ev)))
        return ev

    def event_record(self, ev):
        self._ck(self.lib.pz_event_record(self.handle, ev))

    def event_elapsed_ms(self, e0, e1) -> float:
        ms = C.c_float()
        self._ck(self.lib.pz_event_elapsed_ms(e0, e1, C.byref(ms)))
        return ms.value
