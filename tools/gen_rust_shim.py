#!/usr/bin/env python3
"""Writes rust/poulpy-hip-mi355x/src/hal_impl.rs: `unsafe impl HalImpl<FFT64Hip> for FFT64Hip` with EVERY required method of the
trait (poulpy-hal/src/oep/hal_impl.rs:25-755; none is elided).

The method SIGNATURES are the trait's own — an impl has to repeat them verbatim — and are read from the reference checkout at
generation time (this script runs in the build container only; its output is committed).  The BODIES are this repository's:

  * every method that reads or writes `ScalarPrep` bytes (VecZnxDft / SvpPPol / VmpPMat / CnvPVecL / CnvPVecR: backend-private
    "device order") or that SURVEY.md §8 puts on the hot path forwards to the C ABI function of the same name (FORWARD below);
  * the pure-i64 families (scratch, vec_znx_*, the rest of vec_znx_big_*) delegate to poulpy-cpu-ref's portable defaults on the
    host-addressable pinned buffers, exactly as poulpy-cpu-avx does (poulpy-cpu-avx/src/hal_impl/*.rs).

    python tools/gen_rust_shim.py [--reference /root/reference]

tests/test_rust_shim.py checks the committed file against tests/golden/hal_impl_fns.txt (the trait's required fn names).
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust", "poulpy-hip-mi355x", "src", "hal_impl.rs")
NAMES = os.path.join(ROOT, "tests", "golden", "hal_impl_fns.txt")

# (res / a / b ...) -> the expressions a forward passes for one container argument
V = "{0}.cols(), {0}.size()"

FORWARD = {
    "new": """        crate::check_abi();   // before the first call that allocates anything in the library
        let mut raw: *mut ffi::pz_module = std::ptr::null_mut();
        check(unsafe { ffi::pz_module_new(n, &mut raw) }, "Module::new");
        let handle: Box<FFT64HipHandle> = Box::new(FFT64HipHandle::new(raw));
        unsafe { Module::from_nonnull(NonNull::from(Box::leak(handle)), n) }""",
    # ---- VecZnxBig: the two hot-path ops ----
    "vec_znx_big_normalize_tmp_bytes": "        unsafe { ffi::pz_vec_znx_big_normalize_tmp_bytes(raw(module)) }",
    "vec_znx_big_normalize": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_vec_znx_big_normalize(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_base2k, res_offset, res_col,
            a.as_ptr(), a.cols(), a.size(), a_base2k, a_col) }, "vec_znx_big_normalize");""",
    "vec_znx_big_add_small_assign": """        let mut res = res.to_mut();
        let a = a.to_ref();
        check(unsafe { ffi::pz_vec_znx_big_add_small_assign(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(),
            a.cols(), a.size(), a_col) }, "vec_znx_big_add_small_assign");""",
    # ---- VecZnxDft ----
    "vec_znx_dft_apply": """        let mut res = res.to_mut();
        let a = a.to_ref();
        check(unsafe { ffi::pz_vec_znx_dft_apply(raw(module), step, offset, res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(),
            a.cols(), a.size(), a_col) }, "vec_znx_dft_apply");""",
    "vec_znx_idft_apply_tmp_bytes": "        unsafe { ffi::pz_vec_znx_idft_apply_tmp_bytes(raw(module)) }",
    "vec_znx_idft_apply": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_vec_znx_idft_apply(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(),
            a.size(), a_col) }, "vec_znx_idft_apply");""",
    "vec_znx_idft_apply_tmpa": """        let mut res = res.to_mut();
        let mut a = a.to_mut();
        check(unsafe { ffi::pz_vec_znx_idft_apply_tmpa(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_mut_ptr(),
            a.cols(), a.size(), a_col) }, "vec_znx_idft_apply_tmpa");""",
    "vec_znx_idft_apply_consume": """        let mut a = a;
        {
            let mut v = a.to_mut();
            let (cols, size) = (v.cols(), v.size());
            check(unsafe { ffi::pz_vec_znx_idft_apply_consume(raw(module), v.as_mut_ptr() as *mut std::ffi::c_void, cols, size) },
                "vec_znx_idft_apply_consume");
        }
        a.into_big()""",
    "vec_znx_dft_zero": """        let mut res = res.to_mut();
        check(unsafe { ffi::pz_vec_znx_dft_zero(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col) }, "vec_znx_dft_zero");""",
    # ---- SVP ----
    "svp_prepare": """        let mut res = res.to_mut();
        let a = a.to_ref();
        check(unsafe { ffi::pz_svp_prepare(raw(module), res.as_mut_ptr(), res.cols(), res_col, a.as_ptr(), a.cols(), a_col) }, "svp_prepare");""",
    "svp_apply_dft": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let b = b.to_ref();
        check(unsafe { ffi::pz_svp_apply_dft(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(), a_col,
            b.as_ptr(), b.cols(), b.size(), b_col) }, "svp_apply_dft");""",
    "svp_apply_dft_to_dft": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let b = b.to_ref();
        check(unsafe { ffi::pz_svp_apply_dft_to_dft(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(), a_col,
            b.as_ptr(), b.cols(), b.size(), b_col) }, "svp_apply_dft_to_dft");""",
    "svp_apply_dft_to_dft_assign": """        let mut res = res.to_mut();
        let a = a.to_ref();
        check(unsafe { ffi::pz_svp_apply_dft_to_dft_assign(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(),
            a_col) }, "svp_apply_dft_to_dft_assign");""",
    # ---- VMP ----
    "vmp_prepare_tmp_bytes": "        unsafe { ffi::pz_vmp_prepare_tmp_bytes(raw(module), rows, cols_in, cols_out, size) }",
    "vmp_prepare": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let _ = scratch;
        assert_eq!((res.rows(), res.cols_in(), res.cols_out(), res.size()), (a.rows(), a.cols_in(), a.cols_out(), a.size()));
        check(unsafe { ffi::pz_vmp_prepare(raw(module), res.as_mut_ptr(), a.as_ptr(), a.rows(), a.cols_in(), a.cols_out(), a.size()) },
            "vmp_prepare");""",
    "vmp_apply_dft_tmp_bytes": """        unsafe { ffi::pz_vmp_apply_dft_tmp_bytes(raw(module), res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size) }""",
    "vmp_apply_dft": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let b = b.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_vmp_apply_dft(raw(module), res.as_mut_ptr(), res.cols(), res.size(), a.as_ptr(), a.cols(), a.size(), b.as_ptr(),
            b.rows(), b.cols_in(), b.cols_out(), b.size()) }, "vmp_apply_dft");""",
    "vmp_apply_dft_to_dft_tmp_bytes": """        unsafe { ffi::pz_vmp_apply_dft_to_dft_tmp_bytes(raw(module), res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size) }""",
    "vmp_apply_dft_to_dft": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let b = b.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_vmp_apply_dft_to_dft(raw(module), res.as_mut_ptr(), res.cols(), res.size(), a.as_ptr(), a.cols(), a.size(),
            b.as_ptr(), b.rows(), b.cols_in(), b.cols_out(), b.size(), limb_offset) }, "vmp_apply_dft_to_dft");""",
    "vmp_zero": """        let mut res = res.to_mut();
        let (rows, cols_in, cols_out, size) = (res.rows(), res.cols_in(), res.cols_out(), res.size());
        check(unsafe { ffi::pz_vmp_zero(raw(module), res.as_mut_ptr(), rows, cols_in, cols_out, size) }, "vmp_zero");""",
    # ---- convolution ----
    "cnv_prepare_left_tmp_bytes": "        unsafe { ffi::pz_cnv_prepare_left_tmp_bytes(raw(module), res_size, a_size) }",
    "cnv_prepare_right_tmp_bytes": "        unsafe { ffi::pz_cnv_prepare_right_tmp_bytes(raw(module), res_size, a_size) }",
    "cnv_prepare_self_tmp_bytes": "        unsafe { ffi::pz_cnv_prepare_self_tmp_bytes(raw(module), res_size, a_size) }",
    "cnv_apply_dft_tmp_bytes": "        unsafe { ffi::pz_cnv_apply_dft_tmp_bytes(raw(module), cnv_offset, res_size, a_size, b_size) }",
    "cnv_by_const_apply_tmp_bytes": "        unsafe { ffi::pz_cnv_by_const_apply_tmp_bytes(raw(module), cnv_offset, res_size, a_size, b_size) }",
    "cnv_pairwise_apply_dft_tmp_bytes": "        unsafe { ffi::pz_cnv_pairwise_apply_dft_tmp_bytes(raw(module), cnv_offset, res_size, a_size, b_size) }",
    "cnv_prepare_left": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_cnv_prepare_left(raw(module), res.as_mut_ptr(), res.cols(), res.size(), a.as_ptr(), a.cols(), a.size(), mask) },
            "cnv_prepare_left");""",
    "cnv_prepare_right": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_cnv_prepare_right(raw(module), res.as_mut_ptr(), res.cols(), res.size(), a.as_ptr(), a.cols(), a.size(), mask) },
            "cnv_prepare_right");""",
    "cnv_prepare_self": """        let mut left = left.to_mut();
        let mut right = right.to_mut();
        let a = a.to_ref();
        let _ = scratch;
        assert_eq!((left.cols(), left.size()), (right.cols(), right.size()));
        check(unsafe { ffi::pz_cnv_prepare_self(raw(module), left.as_mut_ptr(), right.as_mut_ptr(), left.cols(), left.size(), a.as_ptr(),
            a.cols(), a.size(), mask) }, "cnv_prepare_self");""",
    "cnv_by_const_apply": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_cnv_by_const_apply(raw(module), cnv_offset, res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(),
            a.cols(), a.size(), a_col, b.as_ptr(), b.len()) }, "cnv_by_const_apply");""",
    "cnv_apply_dft": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let b = b.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_cnv_apply_dft(raw(module), cnv_offset, res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(),
            a.size(), a_col, b.as_ptr(), b.cols(), b.size(), b_col) }, "cnv_apply_dft");""",
    "cnv_pairwise_apply_dft": """        let mut res = res.to_mut();
        let a = a.to_ref();
        let b = b.to_ref();
        let _ = scratch;
        check(unsafe { ffi::pz_cnv_pairwise_apply_dft(raw(module), cnv_offset, res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(),
            a.cols(), a.size(), b.as_ptr(), b.cols(), b.size(), i, j) }, "cnv_pairwise_apply_dft");""",
}

# VecZnxDft limb-wise ops: res (op) a [, b]  -> pz function of the same name with (ptr, cols, size, col) per container
for name, nargs, extra in (("vec_znx_dft_add_into", 3, ""), ("vec_znx_dft_sub", 3, ""), ("vec_znx_dft_add_assign", 2, ""),
                           ("vec_znx_dft_sub_assign", 2, ""), ("vec_znx_dft_sub_negate_assign", 2, ""),
                           ("vec_znx_dft_add_scaled_assign", 2, ", a_scale")):
    lines = ["        let mut res = res.to_mut();", "        let a = a.to_ref();"]
    call = "res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(), a.size(), a_col"
    if nargs == 3:
        lines.append("        let b = b.to_ref();")
        call += ", b.as_ptr(), b.cols(), b.size(), b_col"
    lines.append(f"        check(unsafe {{ ffi::pz_{name}(raw(module), {call}{extra}) }}, \"{name}\");")
    FORWARD[name] = "\n".join(lines)
FORWARD["vec_znx_dft_copy"] = """        let mut res = res.to_mut();
        let a = a.to_ref();
        check(unsafe { ffi::pz_vec_znx_dft_copy(raw(module), step, offset, res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(),
            a.size(), a_col) }, "vec_znx_dft_copy");"""

PROVIDED = {"vec_znx_big_normalize_assign_fallback", "vec_znx_big_normalize_add_assign", "vec_znx_big_normalize_sub_assign"}


def defaults_trait(name: str) -> str:
    if name.startswith("scratch_") or name == "take_slice":
        return "HalScratchDefaults"
    if name.startswith("vec_znx_big_"):
        return "FFT64VecZnxBigDefaults"
    if name.startswith("vec_znx_") and not name.startswith(("vec_znx_dft", "vec_znx_idft")):
        return "HalVecZnxDefaults"
    raise KeyError(name)


def parse_trait(path: str):
    """-> [(name, header text up to and including the where clause, [argument names])] for every method WITHOUT a body."""
    src = open(path).read()
    body = src[src.index("pub unsafe trait HalImpl"):]
    body = body[body.index("{") + 1:]
    out = []
    i = 0
    for m in re.finditer(r"\n    fn (\w+)", body):
        name = m.group(1)
        start = m.start() + 1
        # find the terminating ';' or '{' at depth 0 (outside <>, ())
        depth_par = depth_ang = 0
        j = m.end()
        while True:
            c = body[j]
            if c == "(":
                depth_par += 1
            elif c == ")":
                depth_par -= 1
            elif c in ";{" and depth_par == 0:
                break
            j += 1
        if body[j] == "{":
            continue          # provided method
        header = body[start:j].rstrip()
        # argument names: inside the first top-level parentheses
        p0 = header.index("(")
        depth = 0
        k = p0
        while True:
            if header[k] == "(":
                depth += 1
            elif header[k] == ")":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        args = []
        cur, d = "", 0
        for ch in header[p0 + 1:k]:
            if ch in "<([":
                d += 1
            elif ch in ">)]":
                d -= 1
            if ch == "," and d == 0:
                args.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur)
        names = [a.split(":")[0].strip() for a in args if a.strip()]
        out.append((name, header, names))
    return out


def main():
    ref = "/root/reference"
    if "--reference" in sys.argv:
        ref = sys.argv[sys.argv.index("--reference") + 1]
    methods = parse_trait(os.path.join(ref, "poulpy-hal", "src", "oep", "hal_impl.rs"))
    names = [m[0] for m in methods]
    assert not (set(names) & PROVIDED)
    os.makedirs(os.path.dirname(NAMES), exist_ok=True)
    with open(NAMES, "w") as f:
        f.write("# required (body-less) fns of `unsafe trait HalImpl` (poulpy-hal/src/oep/hal_impl.rs:25-755), written by tools/gen_rust_shim.py\n")
        f.write("\n".join(names) + "\n")
    out = ['''//! `unsafe impl HalImpl<FFT64Hip> for FFT64Hip`: all %d required methods of the trait (poulpy-hal/src/oep/hal_impl.rs:25-755).
//! GENERATED by tools/gen_rust_shim.py — signatures are the trait's own (an impl repeats them verbatim), bodies are either
//!   * a forward to the C ABI function of the same name (include/poulpy_hip.h): every method that touches `ScalarPrep` bytes
//!     (VecZnxDft / SvpPPol / VmpPMat / CnvPVec*, backend-private device order) and the VecZnxBig hot-path ops; `scratch` is
//!     ignored there (the device path owns its workspace; the `*_tmp_bytes` still return the reference's sizes, SURVEY.md A.5); or
//!   * a delegation to poulpy-cpu-ref's portable defaults (hal_defaults) for the pure-i64 families on the pinned host buffers,
//!     as poulpy-cpu-avx does (poulpy-cpu-avx/src/hal_impl/{scratch,vec_znx,vec_znx_big_fft64}.rs).
#![allow(clippy::too_many_arguments)]
use std::ptr::NonNull;

use poulpy_cpu_ref::hal_defaults::{FFT64VecZnxBigDefaults, HalScratchDefaults, HalVecZnxDefaults};
use poulpy_hal::{
    layouts::{
        Backend, CnvPVecLToMut, CnvPVecLToRef, CnvPVecRToMut, CnvPVecRToRef, Data, MatZnxToRef, Module, NoiseInfos, ScalarZnxToRef, Scratch,
        ScratchOwned, SvpPPolToMut, SvpPPolToRef, VecZnxBig, VecZnxBigToMut, VecZnxBigToRef, VecZnxDft, VecZnxDftToMut, VecZnxDftToRef,
        VecZnxToMut, VecZnxToRef, VmpPMatToMut, VmpPMatToRef, ZnxInfos, ZnxView, ZnxViewMut,
    },
    oep::HalImpl,
    source::Source,
};

use crate::{FFT64Hip, FFT64HipHandle, ffi, ffi::check};

/// The C module a call on `module` from THIS thread uses: the handle's own module on the thread that created it, a sibling
/// (`pz_module_clone`: shared device tables, own stream / workspaces / lock) on any other thread, so that the scoped threads poulpy
/// runs over one `&Module` (poulpy-bin-fhe bdd_arithmetic/eval.rs:210-221) overlap on the device instead of queueing on one lock.
#[inline]
pub(crate) fn raw(module: &Module<FFT64Hip>) -> *mut ffi::pz_module {
    unsafe { (*module.ptr()).for_this_thread() }
}

unsafe impl HalImpl<FFT64Hip> for FFT64Hip {''' % len(methods)]
    for name, header, args in methods:
        h = header.replace("crate::layouts::", "").replace("crate::", "poulpy_hal::")
        h = re.sub(r"\bBE\b", "Self", h)
        h = h.replace("Self: Backend", "BE: Backend")   # (no such bound in required methods; keep generic text intact otherwise)
        h = re.sub(r"#\[allow\([^\]]*\)\]\s*", "", h)
        h = re.sub(r"#\[doc\(hidden\)\]\s*", "", h)
        out.append("    " + h.strip() + (" {" if "where" not in h else "\n    {"))
        if name in FORWARD:
            out.append(FORWARD[name])
        else:
            tr = defaults_trait(name)
            call_args = ", ".join(args)
            out.append(f"        <Self as {tr}<Self>>::{name}_default({call_args})")
        out.append("    }\n")
    out.append("}")
    with open(OUT, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"wrote {OUT}: {len(methods)} methods ({sum(1 for m in methods if m[0] in FORWARD)} forwarded to the C ABI)")
    missing = [k for k in FORWARD if k not in names]
    assert not missing, missing


# ------------------------------------------------------------------------------------------------------------------------
# core_impl.rs (feature `core-fused`): `unsafe impl CoreImpl<FFT64Hip>` — poulpy-core/src/oep/core_impl.rs:34.
# Families without an override use the reference's own `impl_core_*_default_methods!` macros (poulpy-core/src/oep/mod.rs:31-42);
# the keyswitch / external-product / automorphism families are written out: the GLWE-level ops forward to the fused device
# pipeline (one call = pass 1 | row pass + VMP + inverse row pass | tail), everything else delegates to Core*Defaults.
# ------------------------------------------------------------------------------------------------------------------------
CORE_OUT = os.path.join(ROOT, "rust", "poulpy-hip-mi355x", "src", "core_impl.rs")
CORE_NAMES = os.path.join(ROOT, "tests", "golden", "core_impl_fns.txt")

EP_BODY = """        assert_eq!(ggsw.rank(), {a}.rank());
        assert_eq!(ggsw.rank(), res.rank());
        assert_eq!(ggsw.n(), res.n());
        let _ = scratch;
        let p = op_params(res.rank().as_usize(), res.rank().as_usize(), ggsw.dnum().as_usize(), ggsw.dsize().as_usize(), ggsw.size(),
            ggsw.base2k().as_usize(), {a}.size(), {a}.base2k().as_usize(), res.size(), res.base2k().as_usize());
        let g = ggsw.to_ref();
{take}
        check(unsafe {{ ffi::pz_glwe_external_product_batched(raw(module), rp, ap, g.data().as_ptr(), &p, 1) }}, "glwe_external_product");"""
KS_BODY = """        assert_eq!({a}.rank(), key.rank_in());
        assert_eq!(res.rank(), key.rank_out());
        let _ = scratch;
        let p = op_params({a}.rank().as_usize(), res.rank().as_usize(), key.dnum().as_usize(), key.dsize().as_usize(), key.size(),
            key.base2k().as_usize(), {a}.size(), {a}.base2k().as_usize(), res.size(), res.base2k().as_usize());
        let k = key.to_ref();
{take}
        check(unsafe {{ {call} }}, "{what}");"""
TAKE_2 = """        let a_ref = a.to_ref();
        let mut r = res.to_mut();
        let (rp, ap) = (r.data_mut().as_mut_ptr(), a_ref.data().as_ptr());"""
TAKE_1 = """        let mut r = res.to_mut();
        let rp = r.data_mut().as_mut_ptr();
        let ap = rp as *const i64;   // *_assign: res is also the input (allowed: every ciphertext is consumed before its result is written)"""

CORE_FORWARD = {
    "glwe_external_product": EP_BODY.format(a="a", take=TAKE_2),
    "glwe_external_product_assign": EP_BODY.format(a="res", take=TAKE_1),
    "glwe_keyswitch": KS_BODY.format(a="a", take=TAKE_2, what="glwe_keyswitch",
                                     call="ffi::pz_glwe_keyswitch_batched(raw(module), rp, ap, k.data().as_ptr(), &p, 1)"),
    "glwe_keyswitch_assign": KS_BODY.format(a="res", take=TAKE_1, what="glwe_keyswitch_assign",
                                            call="ffi::pz_glwe_keyswitch_batched(raw(module), rp, ap, k.data().as_ptr(), &p, 1)"),
}
for base, mode in (("glwe_automorphism", "PZ_AUTO"), ("glwe_automorphism_add", "PZ_AUTO_ADD"), ("glwe_automorphism_sub", "PZ_AUTO_SUB"),
                   ("glwe_automorphism_sub_negate", "PZ_AUTO_SUB_NEGATE")):
    call = f"ffi::pz_glwe_automorphism_batched(raw(module), rp, ap, k.data().as_ptr(), &p, key.p(), ffi::{mode}, 1)"
    CORE_FORWARD[base] = KS_BODY.format(a="a", take=TAKE_2, what=base, call=call)
    CORE_FORWARD[base + "_assign"] = KS_BODY.format(a="res", take=TAKE_1, what=base + "_assign", call=call)

# matrix-level forms (keyswitching/gglwe.rs:29-75, external_product/gglwe.rs:26-77, external_product/ggsw.rs:29-65): the reference loops
# glwe_keyswitch / glwe_external_product over the (row, column) entries; the entries of a GGLWE / GGSW are contiguous GLWEs, so the loop
# is ONE batched call (staged once, `batch` = rows x columns) instead of rows x columns host round trips
MAT_KS_BODY = """        let _ = scratch;
        let res = &mut res.to_mut();
{a_view}
        assert_eq!(res.rank_in(), {a}.rank_in());
        assert_eq!({a}.rank_out(), key.rank_in());
        assert_eq!(res.rank_out(), key.rank_out());
        assert!(res.dnum() <= {a}.dnum());
        assert_eq!(res.base2k(), {a}.base2k());
        let batch = res.dnum().as_usize() * res.rank_in().as_usize();
        if batch == 0 {{
            return;
        }}
        let p = op_params({a}.rank_out().as_usize(), res.rank_out().as_usize(), key.dnum().as_usize(), key.dsize().as_usize(), key.size(),
            key.base2k().as_usize(), {a}.size(), {a}.base2k().as_usize(), res.size(), res.base2k().as_usize());
        let k = key.to_ref();
{ptrs}
        check(unsafe {{ ffi::pz_glwe_keyswitch_batched(raw(module), rp, ap, k.data().as_ptr(), &p, batch) }}, "{what}");"""
MAT_EP_BODY = """        let _ = scratch;
        let res = &mut res.to_mut();
{a_view}
        let g = {ggsw}.to_ref();
        assert_eq!(res.{rank_in}(), {a}.{rank_in}());
        assert_eq!({a}.{rank_out}(), g.rank());
        assert_eq!(res.{rank_out}(), g.rank());
        assert_eq!(res.base2k(), {a}.base2k());
        let cols_m: usize = {cols_m};
        let rows = res.dnum().as_usize().min({a}.dnum().as_usize());
        let batch = rows * cols_m;
        if batch != 0 {{
            let p = op_params(g.rank().as_usize(), g.rank().as_usize(), g.dnum().as_usize(), g.dsize().as_usize(), g.size(),
                g.base2k().as_usize(), {a}.size(), {a}.base2k().as_usize(), res.size(), res.base2k().as_usize());
{ptrs}
            check(unsafe {{ ffi::pz_glwe_external_product_batched(raw(module), rp, ap, g.data().as_ptr(), &p, batch) }}, "{what}");
        }}
        for row in rows..res.dnum().as_usize() {{
            for col in 0..cols_m {{
                res.at_mut(row, col).data_mut().zero();
            }}
        }}"""
MAT_A_VIEW = "        let a = &a.to_ref();"
MAT_PTRS_2 = """        let rp = res.at_mut(0, 0).data_mut().as_mut_ptr();
        let ap = a.at(0, 0).data().as_ptr();"""
MAT_PTRS_1 = """        let rp = res.at_mut(0, 0).data_mut().as_mut_ptr();
        let ap = rp as *const i64;   // *_assign: every entry is consumed before its result is written"""
def _indent(t):
    return "\n".join("    " + l for l in t.split("\n"))
CORE_FORWARD["gglwe_keyswitch"] = MAT_KS_BODY.format(a="a", a_view=MAT_A_VIEW, ptrs=MAT_PTRS_2, what="gglwe_keyswitch")
CORE_FORWARD["gglwe_keyswitch_assign"] = MAT_KS_BODY.format(a="res", a_view="", ptrs=MAT_PTRS_1, what="gglwe_keyswitch_assign")
CORE_FORWARD["gglwe_external_product"] = MAT_EP_BODY.format(a="a", a_view=MAT_A_VIEW, ggsw="b", rank_in="rank_in", rank_out="rank_out",
                                                            cols_m="res.rank_in().as_usize()", ptrs=_indent(MAT_PTRS_2), what="gglwe_external_product")
CORE_FORWARD["gglwe_external_product_assign"] = MAT_EP_BODY.format(a="res", a_view="", ggsw="a", rank_in="rank_in", rank_out="rank_out",
                                                                   cols_m="res.rank_in().as_usize()", ptrs=_indent(MAT_PTRS_1),
                                                                   what="gglwe_external_product_assign")
CORE_FORWARD["ggsw_external_product"] = MAT_EP_BODY.format(a="a", a_view=MAT_A_VIEW, ggsw="b", rank_in="rank", rank_out="rank",
                                                           cols_m="res.rank().as_usize() + 1", ptrs=_indent(MAT_PTRS_2), what="ggsw_external_product")
CORE_FORWARD["ggsw_external_product_assign"] = MAT_EP_BODY.format(a="res", a_view="", ggsw="a", rank_in="rank", rank_out="rank",
                                                                  cols_m="res.rank().as_usize() + 1", ptrs=_indent(MAT_PTRS_1),
                                                                  what="ggsw_external_product_assign")

CORE_FAMILIES = (("keyswitching", "CoreKeyswitchDefaults"), ("external_product", "CoreExternalProductDefaults"),
                 ("automorphism", "CoreAutomorphismDefaults"))


def core_main(ref: str):
    trait_methods = {m[0]: m for m in parse_trait_named(os.path.join(ref, "poulpy-core", "src", "oep", "core_impl.rs"), "pub unsafe trait CoreImpl")}
    with open(CORE_NAMES, "w") as f:
        f.write("# required fns of `unsafe trait CoreImpl` (poulpy-core/src/oep/core_impl.rs:34), written by tools/gen_rust_shim.py\n")
        f.write("\n".join(trait_methods) + "\n")
    out = ['''//! `unsafe impl CoreImpl<FFT64Hip> for FFT64Hip` (feature `core-fused`; poulpy-core/src/oep/core_impl.rs:34).
//! GENERATED by tools/gen_rust_shim.py.  poulpy-core dispatches every high-level algorithm through this trait
//! (poulpy-core/src/oep/mod.rs:31-42), so overriding a method here is how `Module<FFT64Hip>::glwe_external_product` reaches the fused
//! device pipeline instead of the per-op sequence (external_product/glwe.rs:99-141: dft_apply x cols, vmp_apply_dft_to_dft,
//! idft_apply_consume, big_normalize x cols — five kernels and two `VecZnxDft` round trips; fused: three kernels, none).
//!   * GLWE-level key switch / external product / automorphism family: forwarded to `pz_glwe_*_batched` with batch = 1; the C ABI
//!     takes host containers (staged H2D / D2H) and keeps a device mirror of host-resident prepared keys (include/poulpy_hip.h).
//!   * every other method of these three families: `Core*Defaults` (the reference algorithm on top of this backend's HalImpl);
//!   * the remaining families: the reference's own `impl_core_*_default_methods!` macros, unchanged.
//! Features: `core-fused` (default) compiles against the UNTOUCHED reference: it forwards the forms whose key is a `GGSWPrepared`
//! (`GGSWPrepared::data()` is public, prepared/ggsw.rs:177) - glwe / gglwe / ggsw external products.  `GGLWEPrepared` has no public
//! accessor for its `VmpPMat` (prepared/gglwe.rs:20 is `pub(crate)`), so the key-switch and automorphism forwards are compiled only
//! under the non-default feature `ks-fused`, which needs rust/patches/gglwe_prepared_data.patch applied to poulpy-core; without it
//! those methods run `Core*Defaults` (the reference algorithm over this backend's HalImpl).
#![allow(clippy::too_many_arguments)]
use poulpy_core::{
    ScratchTakeCore,
    layouts::{
        GGLWEInfos, GGLWEPreparedToRef, GGLWEToGGSWKeyPreparedToRef, GGLWEToMut, GGLWEToRef, GGSWInfos, GGSWPreparedToRef, GGSWToMut,
        GGSWToRef, GLWEInfos, GLWEToMut, GLWEToRef, GetGaloisElement, LWEInfos, LWEToMut, LWEToRef, SetGaloisElement,
    },
    oep::{CoreAutomorphismDefaults, CoreExternalProductDefaults, CoreImpl, CoreKeyswitchDefaults},
};
use poulpy_hal::layouts::{Module, Scratch, ZnxView, ZnxViewMut, ZnxZero};

use crate::{FFT64Hip, ffi, ffi::check, hal_impl::raw};

#[inline]
fn op_params(rank: usize, rank_out: usize, dnum: usize, dsize: usize, key_size: usize, key_base2k: usize, a_size: usize, a_base2k: usize,
    res_size: usize, res_base2k: usize) -> ffi::pz_glwe_op_params {
    ffi::pz_glwe_op_params {
        rank: rank as u64, dnum: dnum as u64, dsize: dsize as u64, key_size: key_size as u64, key_base2k: key_base2k as u64,
        a_size: a_size as u64, a_base2k: a_base2k as u64, res_size: res_size as u64, res_base2k: res_base2k as u64, rank_out: rank_out as u64,
    }
}

unsafe impl CoreImpl<FFT64Hip> for FFT64Hip {
    poulpy_core::impl_core_decryption_default_methods!(FFT64Hip);
    poulpy_core::impl_core_conversion_default_methods!(FFT64Hip);
    poulpy_core::impl_core_operations_default_methods!(FFT64Hip);
    poulpy_core::impl_core_encryption_default_methods!(FFT64Hip);
''']
    n_fwd = n_all = 0
    for fam, tr in CORE_FAMILIES:
        src = open(os.path.join(ref, "poulpy-core", "src", "oep", fam + ".rs")).read()
        names = re.findall(r"\n        fn (\w+)", src[src.index("macro_rules!"):])
        out.append(f"    // ---- {fam} ----")
        for name in names:
            _, header, args = trait_methods[name]
            h = header.replace("crate::layouts::", "").replace("crate::", "poulpy_core::")
            h = re.sub(r"\bBE\b", "Self", h)
            h = re.sub(r"#\[allow\([^\]]*\)\]\s*", "", h)
            out.append("    " + h.strip() + (" {" if "where" not in h else "\n    {"))
            if name in CORE_FORWARD and "k.data()" in CORE_FORWARD[name]:
                # the key of these forms is a GGLWEPrepared, whose VmpPMat has no public accessor upstream (prepared/gglwe.rs:20):
                # forwarded only under the non-default feature `ks-fused` (rust/patches/gglwe_prepared_data.patch)
                out.append('        #[cfg(feature = "ks-fused")]\n        {')
                out.append(_indent(CORE_FORWARD[name]))
                out.append('        }\n        #[cfg(not(feature = "ks-fused"))]\n        {')
                out.append(f"            <Self as {tr}<Self>>::{name}_default({', '.join(args)})")
                out.append("        }")
                n_fwd += 1
            elif name in CORE_FORWARD:
                out.append(CORE_FORWARD[name])
                n_fwd += 1
            else:
                out.append(f"        <Self as {tr}<Self>>::{name}_default({', '.join(args)})")
            out.append("    }\n")
            n_all += 1
    out.append("}")
    with open(CORE_OUT, "w") as f:
        f.write("\n".join(out) + "\n")
    assert all(k in trait_methods for k in CORE_FORWARD)
    print(f"wrote {CORE_OUT}: {n_all} methods written out ({n_fwd} forwarded to the fused pipeline), 4 families through the reference's macros")


def parse_trait_named(path: str, marker: str):
    src = open(path).read()
    tmp = "/tmp/_trait_slice.rs"
    body = src[src.index(marker):]
    with open(tmp, "w") as f:
        f.write(body.replace(marker, "pub unsafe trait HalImpl", 1))
    return parse_trait(tmp)


if __name__ == "__main__":
    _ref = "/root/reference"
    if "--reference" in sys.argv:
        _ref = sys.argv[sys.argv.index("--reference") + 1]
    main()
    core_main(_ref)
