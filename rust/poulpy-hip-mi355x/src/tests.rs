//! The reference's own cross-backend HAL suite (poulpy-hal/src/test_suite/mod.rs:65, `cross_backend_test_suite!`) instantiated for
//! `FFT64Hip` against `FFT64Ref`, as poulpy-cpu-avx/src/fft64/tests.rs does for `FFT64Avx` — every cross-backend test of the
//! vec_znx / vec_znx_big / vec_znx_dft / svp / vmp families (each test runs the op on both backends on the same seeded inputs and
//! compares the normalized i64 results; DFT-domain bytes are never compared), plus the convolution tests against the naive
//! bivariate product.  Needs an MI355X: `POULPY_HIP_LIB_DIR=<dir of libpoulpy_hip.so> cargo test -p poulpy-hip-mi355x`.
use poulpy_hal::{
    api::ModuleNew,
    backend_test_suite, cross_backend_test_suite,
    layouts::Module,
    test_suite::convolution::{test_convolution, test_convolution_by_const, test_convolution_pairwise},
};

use crate::FFT64Hip;

/// one `cross_backend_test_suite!` per test_suite module: `hal_suite!(module: test, test, ...)`
macro_rules! hal_suite {
    ($m:ident : $($t:ident),+ $(,)?) => {
        cross_backend_test_suite! {
            mod $m,
            backend_ref = poulpy_cpu_ref::FFT64Ref,
            backend_test = crate::FFT64Hip,
            params = TestParams { size: 1 << 8, base2k: 12 },
            tests = { $($t => poulpy_hal::test_suite::$m::$t),+ }
        }
    };
}

hal_suite!(vec_znx:
    test_vec_znx_add_scalar_into, test_vec_znx_add_scalar_assign, test_vec_znx_add_into, test_vec_znx_add_assign,
    test_vec_znx_automorphism, test_vec_znx_automorphism_assign, test_vec_znx_copy, test_vec_znx_merge_rings,
    test_vec_znx_mul_xp_minus_one, test_vec_znx_mul_xp_minus_one_assign, test_vec_znx_negate, test_vec_znx_negate_assign,
    test_vec_znx_normalize, test_vec_znx_normalize_assign, test_vec_znx_rotate, test_vec_znx_rotate_assign, test_vec_znx_lsh,
    test_vec_znx_lsh_assign, test_vec_znx_rsh, test_vec_znx_rsh_assign, test_vec_znx_split_ring, test_vec_znx_sub_scalar,
    test_vec_znx_sub_scalar_assign, test_vec_znx_sub, test_vec_znx_sub_assign, test_vec_znx_sub_negate_assign,
    test_vec_znx_switch_ring);

hal_suite!(vec_znx_big:
    test_vec_znx_big_add_into, test_vec_znx_big_add_assign, test_vec_znx_big_add_small_into, test_vec_znx_big_add_small_assign,
    test_vec_znx_big_automorphism, test_vec_znx_big_automorphism_assign, test_vec_znx_big_negate, test_vec_znx_big_negate_assign,
    test_vec_znx_big_normalize, test_vec_znx_big_normalize_fused, test_vec_znx_big_sub, test_vec_znx_big_sub_assign,
    test_vec_znx_big_sub_negate_assign, test_vec_znx_big_sub_small_a, test_vec_znx_big_sub_small_b,
    test_vec_znx_big_sub_small_a_assign, test_vec_znx_big_sub_small_b_assign);

hal_suite!(vec_znx_dft:
    test_vec_znx_dft_add_into, test_vec_znx_dft_add_assign, test_vec_znx_copy, test_vec_znx_idft_apply, test_vec_znx_idft_apply_tmpa,
    test_vec_znx_idft_apply_consume, test_vec_znx_dft_sub, test_vec_znx_dft_sub_assign, test_vec_znx_dft_sub_negate_assign);

hal_suite!(svp: test_svp_apply_dft, test_svp_apply_dft_to_dft, test_svp_apply_dft_to_dft_assign);

hal_suite!(vmp: test_vmp_apply_dft, test_vmp_apply_dft_to_dft);

backend_test_suite! {
    mod sampling,
    backend = crate::FFT64Hip,
    params = TestParams { size: 1 << 12, base2k: 12 },
    tests = {
        test_vec_znx_fill_uniform => poulpy_hal::test_suite::vec_znx::test_vec_znx_fill_uniform,
        test_vec_znx_fill_normal => poulpy_hal::test_suite::vec_znx::test_vec_znx_fill_normal,
        test_vec_znx_add_normal => poulpy_hal::test_suite::vec_znx::test_vec_znx_add_normal,
    }
}

/// the reference's own instantiation: `Module::new(8)`, base2k 17 (poulpy-cpu-ref/src/tests.rs:11-27) - the smallest ring the FFT64
/// family supports (vmp.rs:67 asserts n >= 8); since round 4 the device library has plans for N = 8 and 16
#[test]
fn test_convolution_by_const_fft64_hip() {
    let module: Module<FFT64Hip> = Module::<FFT64Hip>::new(8);
    test_convolution_by_const(&module, 17);
}

#[test]
fn test_convolution_fft64_hip() {
    let module: Module<FFT64Hip> = Module::<FFT64Hip>::new(8);
    test_convolution(&module, 17);
}

#[test]
fn test_convolution_pairwise_fft64_hip() {
    let module: Module<FFT64Hip> = Module::<FFT64Hip>::new(8);
    test_convolution_pairwise(&module, 17);
}

/// ... and at the HAL suite's ring (N = 256, base2k 12)
#[test]
fn convolution_against_the_naive_bivariate_product() {
    let module = Module::<FFT64Hip>::new(1 << 8);
    test_convolution(&module, 12);
    test_convolution_by_const(&module, 12);
    test_convolution_pairwise(&module, 12);
}

/// every FFT plan family of the device library (N = 8 .. 65536) creates and destroys cleanly
#[test]
fn module_new_every_ring_degree() {
    for log_n in 3..=16 {
        let module = Module::<FFT64Hip>::new(1u64 << log_n);
        assert_eq!(module.n(), 1usize << log_n);
    }
}

/// The per-op batched wrappers stride by the module's ring degree: a container of another degree (an LWE batch has n = n_lwe + 1) must be
/// refused in safe code, before the C ABI sees it (batched.rs `owns`).
#[test]
#[should_panic(expected = "vec_znx_copy_batched: layouts")]
fn batched_per_op_wrappers_refuse_lwe_containers() {
    use crate::batched::HipBatched;
    let module = Module::<FFT64Hip>::new(1 << 10);
    let a = module.lwe_batch_alloc(4, 636, 2, 17);
    let mut res = module.lwe_batch_alloc(4, 636, 2, 17);
    module.vec_znx_copy_batched(&mut res, 0, &a, 0);
}

/// ... and a container that belongs to a module of another ring degree
#[test]
#[should_panic(expected = "vec_znx_add_assign_batched: layouts")]
fn batched_per_op_wrappers_refuse_containers_of_another_ring() {
    use crate::batched::HipBatched;
    let (small, large) = (Module::<FFT64Hip>::new(1 << 8), Module::<FFT64Hip>::new(1 << 10));
    let a = small.vec_znx_batch_alloc(2, 1, 2, 17);
    let mut res = large.vec_znx_batch_alloc(2, 1, 2, 17);
    large.vec_znx_add_assign_batched(&mut res, 0, &a, 0);
}
