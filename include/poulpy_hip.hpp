// poulpy_hip.hpp — header-only C++ mirror of the reference's operator interface for the FFT64 hot
// path, on top of the C ABI (poulpy_hip.h).  Same method names, argument order and error behaviour
// as poulpy-hal's api traits (poulpy-hal/src/api/{vec_znx_dft,svp_ppol,vmp_pmat,vec_znx_big}.rs):
// shape violations and runtime failures throw pz::Error where the reference panics.
// Containers are non-owning views with the reference's layout (znx_base.rs:52-82).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "poulpy_hip.h"

namespace pz {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};
inline void check(int st, const char* what) {
    if (st != 0) throw Error(st, std::string(what) + ": " + pz_last_error());
}

template <typename T>
struct Znx {  // VecZnx (T = int64_t), VecZnxBig (int64_t), VecZnxDft (double)
    T* data;
    size_t n, cols, size;
    T* at(size_t col, size_t limb) const { return data + n * (limb * cols + col); }
};
using VecZnx = Znx<int64_t>;
using VecZnxBig = Znx<int64_t>;
using VecZnxDft = Znx<double>;
struct ScalarZnx { int64_t* data; size_t n, cols; };
struct SvpPPol { double* data; size_t n, cols; };
struct MatZnx { int64_t* data; size_t n, rows, cols_in, cols_out, size; };
struct VmpPMat { double* data; size_t n, rows, cols_in, cols_out, size; };

class Module {  // Module<FFT64Hip>, poulpy-hal/src/layouts/module.rs:97-189
  public:
    explicit Module(uint64_t n) { check_abi(); check(pz_module_new(n, &m_), "Module::new"); }
    Module(uint64_t n, int device) { check_abi(); check(pz_module_new_on_device(n, device, &m_), "Module::new"); }
    // a library built from another revision of poulpy_hip.h would read mismatched struct layouts: refuse it before the first call
    static void check_abi() {
        if (pz_abi_version() != PZ_ABI_VERSION)
            throw Error(PZ_ERR_INVALID, "libpoulpy_hip ABI version " + std::to_string(pz_abi_version()) + ", header expects " + std::to_string(PZ_ABI_VERSION));
    }
    ~Module() { pz_module_free(m_); }
    Module(const Module&) = delete;
    Module& operator=(const Module&) = delete;
    uint64_t n() const { return pz_module_n(m_); }
    pz_module* raw() const { return m_; }
    void sync() { check(pz_module_sync(m_), "sync"); }

    // VecZnxDft
    void vec_znx_dft_apply(size_t step, size_t offset, VecZnxDft& res, size_t res_col, const VecZnx& a, size_t a_col) {
        check(pz_vec_znx_dft_apply(m_, step, offset, res.data, res.cols, res.size, res_col, a.data, a.cols, a.size, a_col), "vec_znx_dft_apply");
    }
    size_t vec_znx_idft_apply_tmp_bytes() const { return pz_vec_znx_idft_apply_tmp_bytes(m_); }
    void vec_znx_idft_apply(VecZnxBig& res, size_t res_col, const VecZnxDft& a, size_t a_col) {
        check(pz_vec_znx_idft_apply(m_, res.data, res.cols, res.size, res_col, a.data, a.cols, a.size, a_col), "vec_znx_idft_apply");
    }
    void vec_znx_idft_apply_tmpa(VecZnxBig& res, size_t res_col, VecZnxDft& a, size_t a_col) {
        check(pz_vec_znx_idft_apply_tmpa(m_, res.data, res.cols, res.size, res_col, a.data, a.cols, a.size, a_col), "vec_znx_idft_apply_tmpa");
    }
    VecZnxBig vec_znx_idft_apply_consume(VecZnxDft a) {
        check(pz_vec_znx_idft_apply_consume(m_, a.data, a.cols, a.size), "vec_znx_idft_apply_consume");
        return VecZnxBig{reinterpret_cast<int64_t*>(a.data), a.n, a.cols, a.size};
    }
    void vec_znx_dft_add_into(VecZnxDft& r, size_t rc, const VecZnxDft& a, size_t ac, const VecZnxDft& b, size_t bc) {
        check(pz_vec_znx_dft_add_into(m_, r.data, r.cols, r.size, rc, a.data, a.cols, a.size, ac, b.data, b.cols, b.size, bc), "vec_znx_dft_add_into");
    }
    void vec_znx_dft_sub(VecZnxDft& r, size_t rc, const VecZnxDft& a, size_t ac, const VecZnxDft& b, size_t bc) {
        check(pz_vec_znx_dft_sub(m_, r.data, r.cols, r.size, rc, a.data, a.cols, a.size, ac, b.data, b.cols, b.size, bc), "vec_znx_dft_sub");
    }
    void vec_znx_dft_add_assign(VecZnxDft& r, size_t rc, const VecZnxDft& a, size_t ac) {
        check(pz_vec_znx_dft_add_assign(m_, r.data, r.cols, r.size, rc, a.data, a.cols, a.size, ac), "vec_znx_dft_add_assign");
    }
    void vec_znx_dft_add_scaled_assign(VecZnxDft& r, size_t rc, const VecZnxDft& a, size_t ac, int64_t a_scale) {
        check(pz_vec_znx_dft_add_scaled_assign(m_, r.data, r.cols, r.size, rc, a.data, a.cols, a.size, ac, a_scale), "vec_znx_dft_add_scaled_assign");
    }
    void vec_znx_dft_sub_assign(VecZnxDft& r, size_t rc, const VecZnxDft& a, size_t ac) {
        check(pz_vec_znx_dft_sub_assign(m_, r.data, r.cols, r.size, rc, a.data, a.cols, a.size, ac), "vec_znx_dft_sub_assign");
    }
    void vec_znx_dft_sub_negate_assign(VecZnxDft& r, size_t rc, const VecZnxDft& a, size_t ac) {
        check(pz_vec_znx_dft_sub_negate_assign(m_, r.data, r.cols, r.size, rc, a.data, a.cols, a.size, ac), "vec_znx_dft_sub_negate_assign");
    }
    void vec_znx_dft_copy(size_t step, size_t offset, VecZnxDft& r, size_t rc, const VecZnxDft& a, size_t ac) {
        check(pz_vec_znx_dft_copy(m_, step, offset, r.data, r.cols, r.size, rc, a.data, a.cols, a.size, ac), "vec_znx_dft_copy");
    }
    void vec_znx_dft_zero(VecZnxDft& r, size_t rc) { check(pz_vec_znx_dft_zero(m_, r.data, r.cols, r.size, rc), "vec_znx_dft_zero"); }

    // SVP
    void svp_prepare(SvpPPol& res, size_t res_col, const ScalarZnx& a, size_t a_col) {
        check(pz_svp_prepare(m_, res.data, res.cols, res_col, a.data, a.cols, a_col), "svp_prepare");
    }
    void svp_apply_dft(VecZnxDft& res, size_t res_col, const SvpPPol& a, size_t a_col, const VecZnx& b, size_t b_col) {
        check(pz_svp_apply_dft(m_, res.data, res.cols, res.size, res_col, a.data, a.cols, a_col, b.data, b.cols, b.size, b_col), "svp_apply_dft");
    }
    void svp_apply_dft_to_dft(VecZnxDft& res, size_t res_col, const SvpPPol& a, size_t a_col, const VecZnxDft& b, size_t b_col) {
        check(pz_svp_apply_dft_to_dft(m_, res.data, res.cols, res.size, res_col, a.data, a.cols, a_col, b.data, b.cols, b.size, b_col), "svp_apply_dft_to_dft");
    }
    void svp_apply_dft_to_dft_assign(VecZnxDft& res, size_t res_col, const SvpPPol& a, size_t a_col) {
        check(pz_svp_apply_dft_to_dft_assign(m_, res.data, res.cols, res.size, res_col, a.data, a.cols, a_col), "svp_apply_dft_to_dft_assign");
    }

    // VMP
    size_t vmp_prepare_tmp_bytes(size_t rows, size_t ci, size_t co, size_t size) const { return pz_vmp_prepare_tmp_bytes(m_, rows, ci, co, size); }
    void vmp_prepare(VmpPMat& res, const MatZnx& a) {
        if (res.rows != a.rows || res.cols_in != a.cols_in || res.cols_out != a.cols_out || res.size != a.size)
            throw Error(PZ_ERR_INVALID, "vmp_prepare: shape mismatch");
        check(pz_vmp_prepare(m_, res.data, a.data, a.rows, a.cols_in, a.cols_out, a.size), "vmp_prepare");
    }
    void vmp_apply_dft(VecZnxDft& res, const VecZnx& a, const VmpPMat& b) {
        check(pz_vmp_apply_dft(m_, res.data, res.cols, res.size, a.data, a.cols, a.size, b.data, b.rows, b.cols_in, b.cols_out, b.size), "vmp_apply_dft");
    }
    void vmp_apply_dft_to_dft(VecZnxDft& res, const VecZnxDft& a, const VmpPMat& b, size_t limb_offset) {
        check(pz_vmp_apply_dft_to_dft(m_, res.data, res.cols, res.size, a.data, a.cols, a.size, b.data, b.rows, b.cols_in, b.cols_out, b.size, limb_offset), "vmp_apply_dft_to_dft");
    }
    void vmp_zero(VmpPMat& r) { check(pz_vmp_zero(m_, r.data, r.rows, r.cols_in, r.cols_out, r.size), "vmp_zero"); }

    // VecZnxBig
    size_t vec_znx_big_normalize_tmp_bytes() const { return pz_vec_znx_big_normalize_tmp_bytes(m_); }
    void vec_znx_big_normalize(VecZnx& res, size_t res_base2k, int64_t res_offset, size_t res_col, const VecZnxBig& a, size_t a_base2k, size_t a_col) {
        check(pz_vec_znx_big_normalize(m_, res.data, res.cols, res.size, res_base2k, res_offset, res_col, a.data, a.cols, a.size, a_base2k, a_col), "vec_znx_big_normalize");
    }
    void vec_znx_big_add_small_assign(VecZnxBig& res, size_t res_col, const VecZnx& a, size_t a_col) {
        check(pz_vec_znx_big_add_small_assign(m_, res.data, res.cols, res.size, res_col, a.data, a.cols, a.size, a_col), "vec_znx_big_add_small_assign");
    }
    // X -> X^p on i64 containers (hal_impl.rs:236-243, :517-524)
    void vec_znx_automorphism(int64_t p, VecZnx& res, size_t res_col, const VecZnx& a, size_t a_col) {
        check(pz_vec_znx_automorphism(m_, p, res.data, res.cols, res.size, res_col, a.data, a.cols, a.size, a_col), "vec_znx_automorphism");
    }
    void vec_znx_automorphism_assign(int64_t p, VecZnx& res, size_t res_col) {
        check(pz_vec_znx_automorphism_assign(m_, p, res.data, res.cols, res.size, res_col), "vec_znx_automorphism_assign");
    }
    void vec_znx_big_automorphism(int64_t p, VecZnxBig& res, size_t res_col, const VecZnxBig& a, size_t a_col) {
        check(pz_vec_znx_big_automorphism(m_, p, res.data, res.cols, res.size, res_col, a.data, a.cols, a.size, a_col), "vec_znx_big_automorphism");
    }
    void vec_znx_big_automorphism_assign(int64_t p, VecZnxBig& res, size_t res_col) {
        check(pz_vec_znx_big_automorphism_assign(m_, p, res.data, res.cols, res.size, res_col), "vec_znx_big_automorphism_assign");
    }

    // batched device-resident GLWE operations (CoreImpl level)
    void glwe_external_product_batched(int64_t* res, const int64_t* a, const double* ggsw, const pz_glwe_op_params& p, size_t batch) {
        check(pz_glwe_external_product_batched(m_, res, a, ggsw, &p, batch), "glwe_external_product_batched");
    }
    void glwe_keyswitch_batched(int64_t* res, const int64_t* a, const double* key, const pz_glwe_op_params& p, size_t batch) {
        check(pz_glwe_keyswitch_batched(m_, res, a, key, &p, batch), "glwe_keyswitch_batched");
    }
    // mode: PZ_AUTO | PZ_AUTO_ADD | PZ_AUTO_SUB | PZ_AUTO_SUB_NEGATE (poulpy-core automorphism/glwe_ct.rs:51-275)
    void glwe_automorphism_batched(int64_t* res, const int64_t* a, const double* key, const pz_glwe_op_params& p, int64_t gal, int mode, size_t batch) {
        check(pz_glwe_automorphism_batched(m_, res, a, key, &p, gal, mode, batch), "glwe_automorphism_batched");
    }
    void blind_rotation_execute_batched(int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk, const pz_blind_rotation_params& p, size_t batch) {
        check(pz_blind_rotation_execute_batched(m_, res, lwe_2n, lut, brk, &p, batch), "blind_rotation_execute_batched");
    }
    void pin_key(const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
        check(pz_module_pin_key(m_, pmat, rows, cols_in, cols_out, size), "module_pin_key");
    }
    void unpin_key(const double* pmat) { check(pz_module_unpin_key(m_, pmat), "module_unpin_key"); }
    void vec_znx_rotate(int64_t k, VecZnx& res, size_t res_col, const VecZnx& a, size_t a_col) {
        check(pz_vec_znx_rotate(m_, k, res.data, res.cols, res.size, res_col, a.data, a.cols, a.size, a_col), "vec_znx_rotate");
    }
    void vec_znx_rotate_assign(int64_t k, VecZnx& res, size_t res_col) {
        check(pz_vec_znx_rotate_assign(m_, k, res.data, res.cols, res.size, res_col), "vec_znx_rotate_assign");
    }
    void vec_znx_rsh_assign(size_t base2k, size_t k, VecZnx& res, size_t res_col) {
        check(pz_vec_znx_rsh_assign(m_, base2k, k, res.data, res.cols, res.size, res_col), "vec_znx_rsh_assign");
    }
    // gals / keys: nsteps host arrays (Galois element and device pointer of the prepared key of every step)
    void glwe_trace_batched(int64_t* res, size_t nsteps, const int64_t* gals, const double* const* keys, const pz_glwe_op_params& p, size_t batch) {
        check(pz_glwe_trace_batched(m_, res, nsteps, gals, keys, &p, batch), "glwe_trace_batched");
    }
    void ggsw_external_product(int64_t* res, const int64_t* a, size_t a_dnum, const double* ggsw, const pz_glwe_op_params& p) {
        check(pz_ggsw_external_product(m_, res, a, a_dnum, ggsw, &p), "ggsw_external_product");
    }
    void circuit_bootstrapping_execute_to_constant_batched(int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                                           size_t nsteps, const int64_t* gals, const double* const* atk,
                                                           const double* const* tsk, const pz_circuit_bootstrapping_params& p, void* tmp,
                                                           size_t tmp_bytes, size_t batch) {
        check(pz_circuit_bootstrapping_execute_to_constant_batched(m_, ggsw, lwe_2n, lut, brk, nsteps, gals, atk, tsk, &p, tmp, tmp_bytes, batch),
              "circuit_bootstrapping_execute_to_constant_batched");
    }
    void circuit_bootstrapping_execute_to_exponent_batched(int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                                           const int64_t* gals, const double* const* atk, const double* const* tsk,
                                                           const pz_circuit_bootstrapping_params& p, size_t log_gap_in, size_t log_gap_out,
                                                           size_t log_domain, void* tmp, size_t tmp_bytes, size_t batch) {
        check(pz_circuit_bootstrapping_execute_to_exponent_batched(m_, ggsw, lwe_2n, lut, brk, gals, atk, tsk, &p, log_gap_in, log_gap_out,
                                                                   log_domain, tmp, tmp_bytes, batch),
              "circuit_bootstrapping_execute_to_exponent_batched");
    }
    size_t circuit_bootstrapping_to_exponent_tmp_bytes(const pz_circuit_bootstrapping_params& p, size_t log_domain, size_t batch) const {
        return pz_circuit_bootstrapping_to_exponent_tmp_bytes(m_, &p, log_domain, batch);
    }
    size_t circuit_bootstrapping_tmp_bytes(const pz_circuit_bootstrapping_params& p, size_t batch) const {
        return pz_circuit_bootstrapping_tmp_bytes(m_, &p, batch);
    }
    // i64 VecZnx limb-wise family (hal_impl.rs:34-131, :289)
    void vec_znx_add_into(int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol, const int64_t* b,
                          size_t bc, size_t bs, size_t bcol) {
        check(pz_vec_znx_add_into(m_, res, rc, rs, rcol, a, ac, as, acol, b, bc, bs, bcol), "vec_znx_add_into");
    }
    void vec_znx_sub(int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol, const int64_t* b,
                     size_t bc, size_t bs, size_t bcol) {
        check(pz_vec_znx_sub(m_, res, rc, rs, rcol, a, ac, as, acol, b, bc, bs, bcol), "vec_znx_sub");
    }
    void vec_znx_add_assign(int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol) {
        check(pz_vec_znx_add_assign(m_, res, rc, rs, rcol, a, ac, as, acol), "vec_znx_add_assign");
    }
    void vec_znx_sub_assign(int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol) {
        check(pz_vec_znx_sub_assign(m_, res, rc, rs, rcol, a, ac, as, acol), "vec_znx_sub_assign");
    }
    void vec_znx_sub_negate_assign(int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol) {
        check(pz_vec_znx_sub_negate_assign(m_, res, rc, rs, rcol, a, ac, as, acol), "vec_znx_sub_negate_assign");
    }
    void vec_znx_negate(int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol) {
        check(pz_vec_znx_negate(m_, res, rc, rs, rcol, a, ac, as, acol), "vec_znx_negate");
    }
    void vec_znx_negate_assign(int64_t* res, size_t rc, size_t rs, size_t rcol) { check(pz_vec_znx_negate_assign(m_, res, rc, rs, rcol), "vec_znx_negate_assign"); }
    void vec_znx_copy(int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol) {
        check(pz_vec_znx_copy(m_, res, rc, rs, rcol, a, ac, as, acol), "vec_znx_copy");
    }
    void vec_znx_zero(int64_t* res, size_t rc, size_t rs, size_t rcol) { check(pz_vec_znx_zero(m_, res, rc, rs, rcol), "vec_znx_zero"); }
    void vec_znx_normalize(int64_t* res, size_t rc, size_t rs, size_t res_base2k, int64_t res_offset, size_t rcol, const int64_t* a, size_t ac,
                           size_t as, size_t a_base2k, size_t acol) {
        check(pz_vec_znx_normalize(m_, res, rc, rs, res_base2k, res_offset, rcol, a, ac, as, a_base2k, acol), "vec_znx_normalize");
    }
    void vec_znx_normalize_assign(size_t base2k, int64_t* res, size_t rc, size_t rs, size_t rcol) {
        check(pz_vec_znx_normalize_assign(m_, base2k, res, rc, rs, rcol), "vec_znx_normalize_assign");
    }
    void vec_znx_lsh(size_t base2k, size_t k, int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol) {
        check(pz_vec_znx_lsh(m_, base2k, k, res, rc, rs, rcol, a, ac, as, acol), "vec_znx_lsh");
    }
    void vec_znx_rsh(size_t base2k, size_t k, int64_t* res, size_t rc, size_t rs, size_t rcol, const int64_t* a, size_t ac, size_t as, size_t acol) {
        check(pz_vec_znx_rsh(m_, base2k, k, res, rc, rs, rcol, a, ac, as, acol), "vec_znx_rsh");
    }
    void vec_znx_lsh_assign(size_t base2k, size_t k, int64_t* res, size_t rc, size_t rs, size_t rcol) {
        check(pz_vec_znx_lsh_assign(m_, base2k, k, res, rc, rs, rcol), "vec_znx_lsh_assign");
    }
    size_t glwe_pack_tmp_bytes(const pz_glwe_op_params& p, size_t batch) const { return pz_glwe_pack_tmp_bytes(m_, &p, batch); }
    void glwe_pack_batched(int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out, const int64_t* gals,
                           const double* const* keys, const pz_glwe_op_params& p, void* tmp, size_t tmp_bytes, size_t batch) {
        check(pz_glwe_pack_batched(m_, res, nslots, indices, cts, log_gap_out, gals, keys, &p, tmp, tmp_bytes, batch), "glwe_pack_batched");
    }
    size_t blind_rotation_extended_tmp_bytes(const pz_blind_rotation_params& p, size_t ext, size_t batch) const {
        return pz_blind_rotation_extended_tmp_bytes(m_, &p, ext, batch);
    }
    void blind_rotation_execute_extended_batched(int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                                 const pz_blind_rotation_params& p, size_t ext, void* tmp, size_t tmp_bytes, size_t batch) {
        check(pz_blind_rotation_execute_extended_batched(m_, res, lwe_2n, lut, brk, &p, ext, tmp, tmp_bytes, batch),
              "blind_rotation_execute_extended_batched");
    }
    void set_graphs(bool enable) { check(pz_module_set_graphs(m_, enable ? 1 : 0), "set_graphs"); }
    uint64_t graph_launches() const { return pz_module_graph_launches(m_); }
    void ggsw_from_gglwe_batched(int64_t* ggsw, const int64_t* a, size_t a_cols_in, size_t dnum, const double* const* tsk,
                                 const pz_glwe_op_params& p, size_t count) {
        check(pz_ggsw_from_gglwe_batched(m_, ggsw, a, a_cols_in, dnum, tsk, &p, count), "ggsw_from_gglwe_batched");
    }
    void ggsw_expand_row_batched(int64_t* ggsw, size_t dnum, const double* const* tsk, const pz_glwe_op_params& p, size_t count) {
        check(pz_ggsw_expand_row_batched(m_, ggsw, dnum, tsk, &p, count), "ggsw_expand_row_batched");
    }

  private:
    pz_module* m_ = nullptr;
};

}  // namespace pz
