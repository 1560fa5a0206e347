//! `unsafe impl HalImpl<FFT64Hip>` — hot-path methods forward to the C ABI; `scratch` is ignored by the
//! device path (the `*_tmp_bytes` functions still return the reference's sizes so that poulpy-core
//! sizes its arena identically).
use poulpy_cpu_ref::hal_defaults::{HalScratchDefaults, HalVecZnxDefaults};
use poulpy_hal::{
    layouts::{
        Data, MatZnxToRef, Module, ScalarZnxToRef, Scratch, SvpPPolToMut, SvpPPolToRef, VecZnxBig, VecZnxBigToMut,
        VecZnxBigToRef, VecZnxDft, VecZnxDftToMut, VecZnxDftToRef, VecZnxToMut, VecZnxToRef, VmpPMatToMut, VmpPMatToRef,
        ZnxInfos, ZnxView, ZnxViewMut,
    },
    oep::HalImpl,
};

use crate::{FFT64Hip, FFT64HipHandle, ffi, ffi::check};

#[inline]
fn raw(module: &Module<FFT64Hip>) -> *mut ffi::pz_module {
    unsafe { (*module.ptr()).raw }
}

unsafe impl HalImpl<FFT64Hip> for FFT64Hip {
    // i64-only families: cpu-ref defaults on the (host-addressable) pinned buffers, as FFT64Avx does
    hal_impl_scratch!();
    hal_impl_vec_znx!();

    // hal_impl.rs:320
    fn new(n: u64) -> Module<Self> {
        let mut m: *mut ffi::pz_module = std::ptr::null_mut();
        check(unsafe { ffi::pz_module_new(n, &mut m) }, "Module::new");
        let handle = Box::new(FFT64HipHandle { raw: m });
        unsafe { Module::from_raw_parts(Box::into_raw(handle), n) }
    }

    // hal_impl.rs:529
    fn vec_znx_dft_apply<R, A>(module: &Module<Self>, step: usize, offset: usize, res: &mut R, res_col: usize, a: &A, a_col: usize)
    where
        R: VecZnxDftToMut<Self>,
        A: VecZnxToRef,
    {
        let mut res = res.to_mut();
        let a = a.to_ref();
        check(
            unsafe {
                ffi::pz_vec_znx_dft_apply(raw(module), step, offset, res.as_mut_ptr(), res.cols(), res.size(), res_col,
                    a.as_ptr(), a.cols(), a.size(), a_col)
            },
            "vec_znx_dft_apply",
        );
    }

    // hal_impl.rs:534
    fn vec_znx_idft_apply_tmp_bytes(_module: &Module<Self>) -> usize {
        0
    }

    // hal_impl.rs:536
    fn vec_znx_idft_apply<R, A>(module: &Module<Self>, res: &mut R, res_col: usize, a: &A, a_col: usize, _scratch: &mut Scratch<Self>)
    where
        R: VecZnxBigToMut<Self>,
        A: VecZnxDftToRef<Self>,
    {
        let mut res = res.to_mut();
        let a = a.to_ref();
        check(
            unsafe {
                ffi::pz_vec_znx_idft_apply(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_col, a.as_ptr(), a.cols(),
                    a.size(), a_col)
            },
            "vec_znx_idft_apply",
        );
    }

    // hal_impl.rs:546 — all cols x limbs in place, then re-type the same bytes (vec_znx_dft.rs:57-59)
    fn vec_znx_idft_apply_consume<D: Data>(module: &Module<Self>, mut a: VecZnxDft<D, Self>) -> VecZnxBig<D, Self>
    where
        VecZnxDft<D, Self>: VecZnxDftToMut<Self>,
    {
        {
            let mut v = a.to_mut();
            let (cols, size) = (v.cols(), v.size());
            check(unsafe { ffi::pz_vec_znx_idft_apply_consume(raw(module), v.as_mut_ptr() as *mut _, cols, size) },
                "vec_znx_idft_apply_consume");
        }
        a.into_big()
    }

    // hal_impl.rs:620
    fn vmp_prepare<R, A>(module: &Module<Self>, res: &mut R, a: &A, _scratch: &mut Scratch<Self>)
    where
        R: VmpPMatToMut<Self>,
        A: MatZnxToRef,
    {
        let mut res = res.to_mut();
        let a = a.to_ref();
        assert_eq!((res.rows(), res.cols_in(), res.cols_out(), res.size()), (a.rows(), a.cols_in(), a.cols_out(), a.size()));
        check(
            unsafe { ffi::pz_vmp_prepare(raw(module), res.as_mut_ptr(), a.as_ptr(), a.rows(), a.cols_in(), a.cols_out(), a.size()) },
            "vmp_prepare",
        );
    }

    // hal_impl.rs:653
    fn vmp_apply_dft_to_dft<R, A, C>(module: &Module<Self>, res: &mut R, a: &A, b: &C, limb_offset: usize, _scratch: &mut Scratch<Self>)
    where
        R: VecZnxDftToMut<Self>,
        A: VecZnxDftToRef<Self>,
        C: VmpPMatToRef<Self>,
    {
        let mut res = res.to_mut();
        let a = a.to_ref();
        let b = b.to_ref();
        check(
            unsafe {
                ffi::pz_vmp_apply_dft_to_dft(raw(module), res.as_mut_ptr(), res.cols(), res.size(), a.as_ptr(), a.cols(), a.size(),
                    b.as_ptr(), b.rows(), b.cols_in(), b.cols_out(), b.size(), limb_offset)
            },
            "vmp_apply_dft_to_dft",
        );
    }

    // hal_impl.rs:431
    fn vec_znx_big_normalize<R, A>(module: &Module<Self>, res: &mut R, res_base2k: usize, res_offset: i64, res_col: usize, a: &A,
        a_base2k: usize, a_col: usize, _scratch: &mut Scratch<Self>)
    where
        R: VecZnxToMut,
        A: VecZnxBigToRef<Self>,
    {
        let mut res = res.to_mut();
        let a = a.to_ref();
        check(
            unsafe {
                ffi::pz_vec_znx_big_normalize(raw(module), res.as_mut_ptr(), res.cols(), res.size(), res_base2k, res_offset, res_col,
                    a.as_ptr(), a.cols(), a.size(), a_base2k, a_col)
            },
            "vec_znx_big_normalize",
        );
    }

    // hal_impl.rs:517 / :524 — used by glwe_automorphism_{add,sub,sub_negate} (poulpy-core automorphism/glwe_ct.rs:96-275)
    fn vec_znx_big_automorphism_assign<A>(module: &Module<Self>, k: i64, a: &mut A, a_col: usize, _s: &mut Scratch<Self>)
    where A: VecZnxBigToMut<Self> {
        let mut a = a.to_mut();
        check(unsafe { ffi::pz_vec_znx_big_automorphism_assign(raw(module), k, a.as_mut_ptr(), a.cols(), a.size(), a_col) },
              "vec_znx_big_automorphism_assign");
    }
    // (vec_znx_automorphism :236 / _assign :243 on the small container may either use pz_vec_znx_automorphism[_assign] or stay
    //  on the inherited cpu-ref implementation: they are pure i64 host operations in the reference.)

    // The remaining DFT-domain methods (vec_znx_idft_apply_tmpa :541, vec_znx_dft_{add_into :553, add_scaled_assign :559,
    // add_assign :564, sub :569, sub_assign :575, sub_negate_assign :580, copy :585, zero :590}, svp_{prepare :595,
    // apply_dft :600, apply_dft_to_dft :606, apply_dft_to_dft_assign :612}, vmp_{apply_dft :636, zero :665},
    // vec_znx_big_add_small_assign :362 and the *_tmp_bytes functions) follow the same three-line pattern with the
    // pz_* function of the same name; see INTEGRATION.md for the full table.  cnv_* : unimplemented!().
}
