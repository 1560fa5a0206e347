//! `extern "C"` declarations of include/poulpy_hip.h (hot-path subset used by the shim).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct pz_module {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct pz_glwe_op_params {
    pub rank: u64,
    pub dnum: u64,
    pub dsize: u64,
    pub key_size: u64,
    pub key_base2k: u64,
    pub a_size: u64,
    pub a_base2k: u64,
    pub res_size: u64,
    pub res_base2k: u64,
    pub rank_out: u64,
}

unsafe extern "C" {
    pub fn pz_last_error() -> *const c_char;
    pub fn pz_module_new(n: u64, out: *mut *mut pz_module) -> c_int;
    pub fn pz_module_free(m: *mut pz_module);
    pub fn pz_module_sync(m: *mut pz_module) -> c_int;
    pub fn pz_alloc_bytes(len: usize) -> *mut c_void;
    pub fn pz_free_bytes(p: *mut c_void);

    pub fn pz_vec_znx_dft_apply(m: *mut pz_module, step: usize, offset: usize, res: *mut f64, res_cols: usize, res_size: usize,
        res_col: usize, a: *const i64, a_cols: usize, a_size: usize, a_col: usize) -> c_int;
    pub fn pz_vec_znx_idft_apply(m: *mut pz_module, res: *mut i64, res_cols: usize, res_size: usize, res_col: usize,
        a: *const f64, a_cols: usize, a_size: usize, a_col: usize) -> c_int;
    pub fn pz_vec_znx_idft_apply_tmpa(m: *mut pz_module, res: *mut i64, res_cols: usize, res_size: usize, res_col: usize,
        a: *mut f64, a_cols: usize, a_size: usize, a_col: usize) -> c_int;
    pub fn pz_vec_znx_idft_apply_consume(m: *mut pz_module, data: *mut c_void, cols: usize, size: usize) -> c_int;
    pub fn pz_vec_znx_dft_add_into(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, a: *const f64, ac: usize,
        as_: usize, acol: usize, b: *const f64, bc: usize, bs: usize, bcol: usize) -> c_int;
    pub fn pz_vec_znx_dft_sub(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, a: *const f64, ac: usize,
        as_: usize, acol: usize, b: *const f64, bc: usize, bs: usize, bcol: usize) -> c_int;
    pub fn pz_vec_znx_dft_add_assign(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, a: *const f64, ac: usize,
        as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_dft_add_scaled_assign(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, a: *const f64,
        ac: usize, as_: usize, acol: usize, a_scale: i64) -> c_int;
    pub fn pz_vec_znx_dft_sub_assign(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, a: *const f64, ac: usize,
        as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_dft_sub_negate_assign(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, a: *const f64,
        ac: usize, as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_dft_copy(m: *mut pz_module, step: usize, offset: usize, res: *mut f64, rc: usize, rs: usize, rcol: usize,
        a: *const f64, ac: usize, as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_dft_zero(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize) -> c_int;

    pub fn pz_svp_prepare(m: *mut pz_module, res: *mut f64, res_cols: usize, res_col: usize, a: *const i64, a_cols: usize,
        a_col: usize) -> c_int;
    pub fn pz_svp_apply_dft(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, ppol: *const f64, ac: usize,
        acol: usize, b: *const i64, bc: usize, bs: usize, bcol: usize) -> c_int;
    pub fn pz_svp_apply_dft_to_dft(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, ppol: *const f64,
        ac: usize, acol: usize, b: *const f64, bc: usize, bs: usize, bcol: usize) -> c_int;
    pub fn pz_svp_apply_dft_to_dft_assign(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, rcol: usize, ppol: *const f64,
        ac: usize, acol: usize) -> c_int;

    pub fn pz_vmp_prepare(m: *mut pz_module, pmat: *mut f64, mat: *const i64, rows: usize, cols_in: usize, cols_out: usize,
        size: usize) -> c_int;
    pub fn pz_vmp_apply_dft(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, a: *const i64, ac: usize, as_: usize,
        pmat: *const f64, rows: usize, cols_in: usize, cols_out: usize, size: usize) -> c_int;
    pub fn pz_vmp_apply_dft_to_dft(m: *mut pz_module, res: *mut f64, rc: usize, rs: usize, a: *const f64, ac: usize, as_: usize,
        pmat: *const f64, rows: usize, cols_in: usize, cols_out: usize, size: usize, limb_offset: usize) -> c_int;
    pub fn pz_vmp_zero(m: *mut pz_module, pmat: *mut f64, rows: usize, cols_in: usize, cols_out: usize, size: usize) -> c_int;

    pub fn pz_vec_znx_big_normalize(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, res_base2k: usize, res_offset: i64,
        rcol: usize, a: *const i64, ac: usize, as_: usize, a_base2k: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_big_add_small_assign(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64,
        ac: usize, as_: usize, acol: usize) -> c_int;

    pub fn pz_vec_znx_automorphism(m: *mut pz_module, p: i64, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64,
        ac: usize, as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_automorphism_assign(m: *mut pz_module, p: i64, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    pub fn pz_vec_znx_automorphism_assign_tmp_bytes(m: *const pz_module) -> usize;
    pub fn pz_vec_znx_big_automorphism(m: *mut pz_module, p: i64, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64,
        ac: usize, as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_big_automorphism_assign(m: *mut pz_module, p: i64, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    pub fn pz_vec_znx_big_automorphism_assign_tmp_bytes(m: *const pz_module) -> usize;

    pub fn pz_glwe_external_product_batched(m: *mut pz_module, res: *mut i64, a: *const i64, ggsw: *const f64,
        p: *const pz_glwe_op_params, batch: usize) -> c_int;
    pub fn pz_glwe_keyswitch_batched(m: *mut pz_module, res: *mut i64, a: *const i64, key: *const f64,
        p: *const pz_glwe_op_params, batch: usize) -> c_int;
    /// mode: 0 glwe_automorphism, 1 _add, 2 _sub, 3 _sub_negate (poulpy-core/src/automorphism/glwe_ct.rs:51-275)
    pub fn pz_glwe_automorphism_batched(m: *mut pz_module, res: *mut i64, a: *const i64, key: *const f64,
        p: *const pz_glwe_op_params, gal: i64, mode: c_int, batch: usize) -> c_int;
    /// BlindRotationExecute<CGGI>::blind_rotation_execute on a batch (poulpy-bin-fhe blind_rotation/algorithms/cggi/algorithm.rs)
    pub fn pz_blind_rotation_execute_batched(m: *mut pz_module, res: *mut i64, lwe_2n: *const i64, lut: *const i64, brk: *const f64,
        p: *const pz_blind_rotation_params, batch: usize) -> c_int;
    pub fn pz_blind_rotation_workspace_bytes(m: *const pz_module, p: *const pz_blind_rotation_params, batch: usize) -> usize;
    pub fn pz_module_pin_key(m: *mut pz_module, pmat: *const f64, rows: usize, cols_in: usize, cols_out: usize, size: usize) -> c_int;
    pub fn pz_module_unpin_key(m: *mut pz_module, pmat: *const f64) -> c_int;
    pub fn pz_vec_znx_rotate(m: *mut pz_module, k: i64, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize,
        as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_rotate_assign(m: *mut pz_module, k: i64, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    pub fn pz_vec_znx_rsh_assign(m: *mut pz_module, base2k: usize, k: usize, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    /// glwe_trace_assign (poulpy-core/src/glwe_trace.rs:129-176): host arrays of Galois elements / device key pointers per step
    pub fn pz_glwe_trace_batched(m: *mut pz_module, res: *mut i64, nsteps: usize, gals: *const i64, keys: *const *const f64,
        p: *const pz_glwe_op_params, batch: usize) -> c_int;
    pub fn pz_ggsw_external_product(m: *mut pz_module, res: *mut i64, a: *const i64, a_dnum: usize, ggsw: *const f64,
        p: *const pz_glwe_op_params) -> c_int;
    /// ggsw_expand_row (poulpy-core/src/conversion/gglwe_to_ggsw.rs:116-268), in place on `count` contiguous device GGSWs;
    /// tsk: host array of `rank` device pointers (tsk.at(col - 1))
    /// ggsw_from_gglwe (conversion/gglwe_to_ggsw.rs:32-61): strided copy of a.at(row, 0), then ggsw_expand_row
    pub fn pz_ggsw_from_gglwe_batched(m: *mut pz_module, ggsw: *mut i64, a: *const i64, a_cols_in: usize, dnum: usize,
        tsk: *const *const f64, p: *const pz_glwe_op_params, count: usize) -> c_int;
    pub fn pz_ggsw_expand_row_batched(m: *mut pz_module, ggsw: *mut i64, dnum: usize, tsk: *const *const f64,
        p: *const pz_glwe_op_params, count: usize) -> c_int;
}

#[repr(C)]
pub struct pz_blind_rotation_params {
    pub rank: u64, pub n_lwe: u64, pub block_size: u64, pub dnum: u64, pub brk_size: u64, pub base2k: u64, pub res_size: u64,
    pub lut_size: u64,
}

/// The reference panics (`assert!`) on shape / scratch violations; so does the shim.
#[inline]
pub fn check(status: c_int, what: &str) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(pz_last_error()) }.to_string_lossy().into_owned();
        panic!("{what}: libpoulpy_hip status {status}: {msg}");
    }
}

#[repr(C)]
pub struct pz_circuit_bootstrapping_params {
    pub br: pz_blind_rotation_params,
    pub atk_dnum: u64,
    pub atk_size: u64,
    pub tsk_dnum: u64,
    pub tsk_size: u64,
    pub res_dnum: u64,
    pub res_size: u64,
    pub gap: u64,
    pub extension_factor: u64,
}

extern "C" {
    /// circuit_bootstrapping_execute_to_constant (poulpy-bin-fhe/src/circuit_bootstrapping/circuit.rs:177-195, :219-370), one base2k
    // i64 VecZnx limb-wise family (hal_impl.rs:34-131, :289)
    pub fn pz_vec_znx_add_into(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize, as_: usize,
        acol: usize, b: *const i64, bc: usize, bs: usize, bcol: usize) -> c_int;
    pub fn pz_vec_znx_sub(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize, as_: usize,
        acol: usize, b: *const i64, bc: usize, bs: usize, bcol: usize) -> c_int;
    pub fn pz_vec_znx_add_assign(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize, as_: usize,
        acol: usize) -> c_int;
    pub fn pz_vec_znx_sub_assign(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize, as_: usize,
        acol: usize) -> c_int;
    pub fn pz_vec_znx_sub_negate_assign(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize,
        as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_negate(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize, as_: usize,
        acol: usize) -> c_int;
    pub fn pz_vec_znx_negate_assign(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    pub fn pz_vec_znx_copy(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64, ac: usize, as_: usize,
        acol: usize) -> c_int;
    pub fn pz_vec_znx_zero(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    pub fn pz_vec_znx_normalize_tmp_bytes(m: *const pz_module) -> usize;
    pub fn pz_vec_znx_normalize(m: *mut pz_module, res: *mut i64, rc: usize, rs: usize, res_base2k: usize, res_offset: i64, rcol: usize,
        a: *const i64, ac: usize, as_: usize, a_base2k: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_normalize_assign(m: *mut pz_module, base2k: usize, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    pub fn pz_vec_znx_lsh_tmp_bytes(m: *const pz_module) -> usize;
    pub fn pz_vec_znx_lsh(m: *mut pz_module, base2k: usize, k: usize, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64,
        ac: usize, as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_rsh(m: *mut pz_module, base2k: usize, k: usize, res: *mut i64, rc: usize, rs: usize, rcol: usize, a: *const i64,
        ac: usize, as_: usize, acol: usize) -> c_int;
    pub fn pz_vec_znx_lsh_assign(m: *mut pz_module, base2k: usize, k: usize, res: *mut i64, rc: usize, rs: usize, rcol: usize) -> c_int;
    /// glwe_pack (poulpy-core/src/glwe_packing.rs:122-176): host arrays of indices / device ciphertext pointers / per-step keys
    pub fn pz_glwe_pack_tmp_bytes(m: *const pz_module, p: *const pz_glwe_op_params, batch: usize) -> usize;
    pub fn pz_glwe_pack_batched(m: *mut pz_module, res: *mut i64, nslots: usize, indices: *const u64, cts: *const *mut i64,
        log_gap_out: usize, gals: *const i64, keys: *const *const f64, p: *const pz_glwe_op_params, tmp: *mut c_void, tmp_bytes: usize,
        batch: usize) -> c_int;
    /// execute_block_binary_extended (algorithm.rs:121-273): extension_factor > 1
    pub fn pz_blind_rotation_extended_tmp_bytes(m: *const pz_module, p: *const pz_blind_rotation_params, ext: usize, batch: usize) -> usize;
    pub fn pz_blind_rotation_execute_extended_batched(m: *mut pz_module, res: *mut i64, lwe_2n: *const i64, lut: *const i64,
        brk: *const f64, p: *const pz_blind_rotation_params, ext: usize, tmp: *mut c_void, tmp_bytes: usize, batch: usize) -> c_int;
    /// HIP-graph replay of the launch-bound composite calls (on by default)
    pub fn pz_module_set_graphs(m: *mut pz_module, enable: c_int) -> c_int;
    pub fn pz_module_graph_launches(m: *const pz_module) -> u64;
    pub fn pz_circuit_bootstrapping_to_exponent_tmp_bytes(m: *const pz_module, p: *const pz_circuit_bootstrapping_params, log_domain: usize,
        batch: usize) -> usize;
    pub fn pz_circuit_bootstrapping_execute_to_exponent_batched(m: *mut pz_module, ggsw: *mut i64, lwe_2n: *const i64, lut: *const i64,
        brk: *const f64, gals: *const i64, atk: *const *const f64, tsk: *const *const f64, p: *const pz_circuit_bootstrapping_params,
        log_gap_in: usize, log_gap_out: usize, log_domain: usize, tmp: *mut c_void, tmp_bytes: usize, batch: usize) -> c_int;
    pub fn pz_circuit_bootstrapping_tmp_bytes(m: *const pz_module, p: *const pz_circuit_bootstrapping_params, batch: usize) -> usize;
    pub fn pz_circuit_bootstrapping_execute_to_constant_batched(m: *mut pz_module, ggsw: *mut i64, lwe_2n: *const i64, lut: *const i64,
        brk: *const f64, nsteps: usize, gals: *const i64, atk: *const *const f64, tsk: *const *const f64,
        p: *const pz_circuit_bootstrapping_params, tmp: *mut c_void, tmp_bytes: usize, batch: usize) -> c_int;
}
