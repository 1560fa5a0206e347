"""CPU: the C-ABI library loads and exports every symbol include/poulpy_hip.h declares; host-side
logic (layouts, byte sizes, sharding) and loud failure without a device.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "poulpy_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pz_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from poulpy_amd.hal import load_library
    return load_library()


def test_header_declares_hot_path_entry_points():
    syms = _declared_symbols()
    for required in ("pz_module_new", "pz_vec_znx_dft_apply", "pz_vec_znx_idft_apply", "pz_vec_znx_idft_apply_tmpa",
                     "pz_vec_znx_idft_apply_consume", "pz_svp_prepare", "pz_svp_apply_dft", "pz_svp_apply_dft_to_dft",
                     "pz_svp_apply_dft_to_dft_assign", "pz_vmp_prepare", "pz_vmp_apply_dft", "pz_vmp_apply_dft_to_dft",
                     "pz_vmp_zero", "pz_vec_znx_big_normalize", "pz_vec_znx_big_add_small_assign", "pz_vec_znx_dft_add_assign",
                     "pz_vec_znx_dft_sub_assign", "pz_vec_znx_dft_copy", "pz_vec_znx_dft_zero", "pz_glwe_external_product_batched",
                     "pz_glwe_keyswitch_batched"):
        assert required in syms


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/poulpy_hip.h but not exported: {missing}"
    from poulpy_amd.hal import PZ_ABI_VERSION
    assert lib.pz_abi_version() == PZ_ABI_VERSION


def test_byte_sizes_match_reference_formulas(lib):
    # poulpy-hal/src/layouts/module.rs:51-65
    n = 1 << 12
    assert lib.pz_bytes_of_vec_znx_dft(C.c_uint64(n), C.c_size_t(2), C.c_size_t(4)) == n * 2 * 4 * 8
    assert lib.pz_bytes_of_vec_znx_big(C.c_uint64(n), C.c_size_t(2), C.c_size_t(4)) == n * 2 * 4 * 8
    assert lib.pz_bytes_of_svp_ppol(C.c_uint64(n), C.c_size_t(3)) == n * 3 * 8
    assert lib.pz_bytes_of_vmp_pmat(C.c_uint64(n), C.c_size_t(4), C.c_size_t(2), C.c_size_t(2), C.c_size_t(4)) == n * 4 * 2 * 2 * 4 * 8
    # scratch sizes the callers assume; module pointer may be NULL for the shape-only ones
    lib.pz_vmp_apply_dft_to_dft_tmp_bytes.argtypes = [C.c_void_p] + [C.c_size_t] * 6
    assert lib.pz_vmp_apply_dft_to_dft_tmp_bytes(None, 4, 3, 5, 2, 2, 4) == (16 + 8 * 3 * 2) * 8


def test_module_new_rejects_bad_n_before_touching_the_device(lib):
    h = C.c_void_p()
    assert lib.pz_module_new(C.c_uint64(12345), C.byref(h)) == -1  # PZ_ERR_INVALID: not a power of two
    assert b"power of two" in lib.pz_last_error()
    assert lib.pz_module_new(C.c_uint64(4), C.byref(h)) == -2      # PZ_ERR_UNSUPPORTED: n < 8 (vmp.rs:67 asserts n >= 8)
    assert lib.pz_module_new(C.c_uint64(1 << 18), C.byref(h)) == -2  # beyond the largest plan


def test_no_cpu_fallback_without_a_device():
    import torch
    from poulpy_amd.hal import Module, PoulpyHipError
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(PoulpyHipError):
        Module(1 << 10)


def test_missing_library_fails_loudly(tmp_path):
    from poulpy_amd import hal
    with pytest.raises(hal.PoulpyHipError):
        hal.load_library(str(tmp_path / "nope.so"))


def test_layouts_follow_reference_offsets():
    from poulpy_amd.layouts import MatZnx, VecZnx, VecZnxDft
    n, cols, size = 16, 3, 4
    v = VecZnx(n, cols, size)
    flat = v.data.reshape(-1)
    flat[:] = np.arange(flat.size)
    # limb j of column i starts at n*(j*cols + i)  (znx_base.rs:71-76)
    for i in range(cols):
        for j in range(size):
            assert v.at(i, j)[0] == n * (j * cols + i)
    m = MatZnx(n, 2, 3, 2, 4)
    m.data.reshape(-1)[:] = np.arange(m.data.size)
    # entry (row, col_in) is a VecZnx(cols_out, size) at n*cols_out*size*(cols_in*row + col_in)  (mat_znx.rs:161-181)
    assert m.at(1, 2).at(0, 0)[0] == n * 2 * 4 * (3 * 1 + 2)
    d = VecZnxDft(n, 2, 3)
    big = d.into_big()
    assert big.data.dtype == np.int64 and np.shares_memory(big.data, d.data)
    assert v.view(2).data.shape == (2, cols, n) and np.shares_memory(v.view(2).data, v.data)


def test_shard_range_partitions_exactly():
    from poulpy_amd.dist import shard_range
    for total in (0, 1, 7, 128, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_cpp_header_mirror_compiles_and_reports_errors(lib, tmp_path):
    """include/poulpy_hip.hpp (C++ mirror of the operator interface) builds with g++ against the C ABI and
    turns a status code into an exception the way the reference panics."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text('#include "poulpy_hip.hpp"\n'
                   'int main() { try { pz::Module m(12345); } catch (const pz::Error& e) { return e.status == PZ_ERR_INVALID ? 0 : 1; } return 2; }\n')
    exe = tmp_path / "t"
    libdir = os.path.join(ROOT, "poulpy_amd")
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lpoulpy_hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
