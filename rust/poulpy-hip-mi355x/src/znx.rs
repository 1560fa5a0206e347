//! `Znx*` single-polynomial traits for `FFT64Hip` (poulpy-cpu-ref/src/reference/znx): plain `&[i64]` slices with the one canonical
//! layout shared by every backend.  The portable defaults this backend inherits for the pure-i64 families
//! (`HalVecZnxDefaults`, `FFT64VecZnxBigDefaults`) are generic over these traits; they run on the host-addressable pinned buffers,
//! so each one delegates to the reference kernel, as `FFT64Ref` does (poulpy-cpu-ref/src/fft64/znx.rs).
use poulpy_cpu_ref::reference::znx::*;

use crate::FFT64Hip;

macro_rules! znx_delegate {
    ($($tr:ident :: $f:ident $(<const $c:ident : bool>)? ( $($a:ident : $t:ty),* ) => $r:ident $(::<$rc:ident>)? ;)*) => {
        $(impl $tr for FFT64Hip {
            #[inline(always)]
            fn $f $(<const $c: bool>)? ($($a: $t),*) {
                $r $(::<$rc>)? ($($a),*)
            }
        })*
    };
}

znx_delegate! {
    ZnxAdd::znx_add(res: &mut [i64], a: &[i64], b: &[i64]) => znx_add_ref;
    ZnxAddAssign::znx_add_assign(res: &mut [i64], a: &[i64]) => znx_add_assign_ref;
    ZnxSub::znx_sub(res: &mut [i64], a: &[i64], b: &[i64]) => znx_sub_ref;
    ZnxSubAssign::znx_sub_assign(res: &mut [i64], a: &[i64]) => znx_sub_assign_ref;
    ZnxSubNegateAssign::znx_sub_negate_assign(res: &mut [i64], a: &[i64]) => znx_sub_negate_assign_ref;
    ZnxMulAddPowerOfTwo::znx_muladd_power_of_two(k: i64, res: &mut [i64], a: &[i64]) => znx_mul_add_power_of_two_ref;
    ZnxMulPowerOfTwo::znx_mul_power_of_two(k: i64, res: &mut [i64], a: &[i64]) => znx_mul_power_of_two_ref;
    ZnxMulPowerOfTwoAssign::znx_mul_power_of_two_assign(k: i64, res: &mut [i64]) => znx_mul_power_of_two_assign_ref;
    ZnxAutomorphism::znx_automorphism(p: i64, res: &mut [i64], a: &[i64]) => znx_automorphism_ref;
    ZnxCopy::znx_copy(res: &mut [i64], a: &[i64]) => znx_copy_ref;
    ZnxNegate::znx_negate(res: &mut [i64], src: &[i64]) => znx_negate_ref;
    ZnxNegateAssign::znx_negate_assign(res: &mut [i64]) => znx_negate_assign_ref;
    ZnxRotate::znx_rotate(p: i64, res: &mut [i64], src: &[i64]) => znx_rotate;
    ZnxZero::znx_zero(res: &mut [i64]) => znx_zero_ref;
    ZnxSwitchRing::znx_switch_ring(res: &mut [i64], a: &[i64]) => znx_switch_ring_ref;
    ZnxNormalizeFirstStep::znx_normalize_first_step<const OVERWRITE: bool>(base2k: usize, lsh: usize, x: &mut [i64], a: &[i64], carry: &mut [i64]) => znx_normalize_first_step_ref::<OVERWRITE>;
    ZnxNormalizeMiddleStep::znx_normalize_middle_step<const OVERWRITE: bool>(base2k: usize, lsh: usize, x: &mut [i64], a: &[i64], carry: &mut [i64]) => znx_normalize_middle_step_ref::<OVERWRITE>;
    ZnxNormalizeFinalStep::znx_normalize_final_step<const OVERWRITE: bool>(base2k: usize, lsh: usize, x: &mut [i64], a: &[i64], carry: &mut [i64]) => znx_normalize_final_step_ref::<OVERWRITE>;
    ZnxNormalizeMiddleStepSub::znx_normalize_middle_step_sub(base2k: usize, lsh: usize, x: &mut [i64], a: &[i64], carry: &mut [i64]) => znx_normalize_middle_step_sub_ref;
    ZnxNormalizeFinalStepSub::znx_normalize_final_step_sub(base2k: usize, lsh: usize, x: &mut [i64], a: &[i64], carry: &mut [i64]) => znx_normalize_final_step_sub_ref;
    ZnxNormalizeFinalStepAssign::znx_normalize_final_step_assign(base2k: usize, lsh: usize, x: &mut [i64], carry: &mut [i64]) => znx_normalize_final_step_assign_ref;
    ZnxNormalizeFirstStepCarryOnly::znx_normalize_first_step_carry_only(base2k: usize, lsh: usize, x: &[i64], carry: &mut [i64]) => znx_normalize_first_step_carry_only_ref;
    ZnxNormalizeFirstStepAssign::znx_normalize_first_step_assign(base2k: usize, lsh: usize, x: &mut [i64], carry: &mut [i64]) => znx_normalize_first_step_assign_ref;
    ZnxNormalizeMiddleStepCarryOnly::znx_normalize_middle_step_carry_only(base2k: usize, lsh: usize, x: &[i64], carry: &mut [i64]) => znx_normalize_middle_step_carry_only_ref;
    ZnxNormalizeMiddleStepAssign::znx_normalize_middle_step_assign(base2k: usize, lsh: usize, x: &mut [i64], carry: &mut [i64]) => znx_normalize_middle_step_assign_ref;
    ZnxExtractDigitAddMul::znx_extract_digit_addmul(base2k: usize, lsh: usize, res: &mut [i64], src: &mut [i64]) => znx_extract_digit_addmul_ref;
    ZnxNormalizeDigit::znx_normalize_digit(base2k: usize, res: &mut [i64], src: &mut [i64]) => znx_normalize_digit_ref;
}
