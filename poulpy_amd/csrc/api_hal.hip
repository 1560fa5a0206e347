// api_hal.hip — C ABI of the per-op HalImpl methods of the FFT64 hot path (poulpy-hal/src/oep/hal_impl.rs): VecZnxDft, SVP, VMP,
// VecZnxBig and the i64 VecZnx family.  Host or device pointers; host arguments are staged (api_common.hpp) and the call is then
// logically synchronous.
#include "api_common.hpp"
#include "api_glwe.hpp"

using namespace pz;

extern "C" {

// ------------------------------------------------------------------------------
// public: VecZnxDft
// ------------------------------------------------------------------------------


int pz_vec_znx_dft_apply(pz_module* M, size_t step, size_t offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                         const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(step > 0, "vec_znx_dft_apply: step must be > 0");
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_apply(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_apply(a)");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));  // inout: untouched limbs / other columns survive
    cplx* T;
    PZ_TRY(need_T(M, std::min(res_size, a_size), &T));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_dft_apply(M, 1, (int)step, (int)offset, dr, (int)res_col, da, (int)a_col, 1, nullptr, T));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_vec_znx_dft_apply_batched(pz_module* M, size_t batch, size_t step, size_t offset, double* res, size_t res_cols,
                                 size_t res_size, size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(step > 0, "vec_znx_dft_apply: step must be > 0");
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_apply(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_apply(a)");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a), "batched entry points take device pointers");
    cplx* T;
    PZ_TRY(need_T(M, batch * std::min(res_size, a_size), &T));
    DV dr{res, (long long)(M->n * res_cols * res_size), (int)res_cols, (int)res_size};
    DV da{(void*)a, (long long)(M->n * a_cols * a_size), (int)a_cols, (int)a_size};
    return dev_dft_apply(M, (int)batch, (int)step, (int)offset, dr, (int)res_col, da, (int)a_col, 1, nullptr, T);
}

size_t pz_vec_znx_idft_apply_tmp_bytes(const pz_module*) { return 0; }  // hal_defaults/vec_znx_dft.rs:68-73

static int idft_common(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                       size_t a_cols, size_t a_size, size_t a_col) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_idft_apply(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_idft_apply(a)");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));
    const int min_size = (int)std::min(res_size, a_size);
    cplx* T;
    PZ_TRY(need_T(M, min_size, &T));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_idft(M, 1, dr, (int)res_col, da, (int)a_col, 1, min_size, T));
    PZ_TRY(launch_ew(M, EW_ZERO, poly_ptr(M, dr, (int)res_col, min_size), 0, limb_stride(M, dr), nullptr, 0, 0, nullptr, 0, 0,
                     (int)res_size - min_size, 1));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_vec_znx_idft_apply(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                          size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return idft_common(M, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_idft_apply_tmpa(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, double* a,
                               size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);  // `a` may be used as scratch by the reference; this backend leaves it intact
    return idft_common(M, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}

static int consume_common(pz_module* M, size_t batch, void* data, size_t cols, size_t size, bool require_dev) {
    if (require_dev) PZ_REQUIRE(is_device_ptr(data), "batched entry points take device pointers");
    Stage sd;
    if (require_dev) { sd.M = M; sd.dev = data; }
    else PZ_TRY(sd.in(data, vbytes(M, cols, size), true, true, M));
    cplx* T;
    PZ_TRY(need_T(M, batch * cols * size, &T));
    DV d{sd.dev, (long long)(M->n * cols * size), (int)cols, (int)size};
    PZ_TRY(dev_idft(M, (int)batch, d, 0, d, 0, (int)cols, (int)size, T));
    const bool host = sd.owned;
    PZ_TRY(sd.finish());
    return finish_call(M, host);
}
int pz_vec_znx_idft_apply_consume(pz_module* M, void* data, size_t cols, size_t size) {
    PZ_ENTER(M);
    return consume_common(M, 1, data, cols, size, false);
}
int pz_vec_znx_idft_apply_consume_batched(pz_module* M, size_t batch, void* data, size_t cols, size_t size) {
    PZ_ENTER(M);
    return consume_common(M, batch, data, cols, size, true);
}

// generic staged three-operand limb-range op helper
struct Tri {
    Stage sr, sa, sb;
    DV dr, da, db;
    bool host = false;
};
static int tri_in(pz_module* M, Tri& t, double* res, size_t rc, size_t rs, const double* a, size_t ac, size_t as_, const double* b,
                  size_t bc, size_t bs_) {
    PZ_TRY(t.sa.in(a, a ? vbytes(M, ac, as_) : 0, true, false, M));
    if (b) PZ_TRY(t.sb.in(b, vbytes(M, bc, bs_), true, false, M));
    // res aliasing a or b (assign forms pass res as operand): reuse the same staging
    if ((const void*)res == (const void*)a) { t.sr.M = M; t.sr.dev = t.sa.dev; t.sa.out = true; }
    else PZ_TRY(t.sr.in(res, vbytes(M, rc, rs), true, true, M));
    t.dr = DV{t.sr.dev, 0, (int)rc, (int)rs};
    t.da = DV{t.sa.dev, 0, (int)ac, (int)as_};
    t.db = DV{t.sb.dev, 0, (int)bc, (int)bs_};
    t.host = t.sr.owned || t.sa.owned || t.sb.owned;
    return PZ_OK;
}
static int tri_out(pz_module* M, Tri& t) {
    PZ_TRY(t.sr.finish());
    PZ_TRY(t.sa.finish());
    PZ_TRY(t.sb.finish());
    return finish_call(M, t.host);
}
static int ew_limbs(pz_module* M, int op, const DV& r, int rcol, int rl0, const DV* a, int acol, int al0, const DV* b, int bcol,
                    int bl0, int nl) {
    return launch_ew(M, op, poly_ptr(M, r, rcol, rl0), 0, limb_stride(M, r), a ? poly_ptr(M, *a, acol, al0) : nullptr, 0,
                     a ? limb_stride(M, *a) : 0, b ? poly_ptr(M, *b, bcol, bl0) : nullptr, 0, b ? limb_stride(M, *b) : 0, nl, 1);
}

// i64 = true: the same limb-range logic on i64 containers (reference/vec_znx/add.rs:6-65, sub.rs:6-58), wrapping arithmetic
static inline int ew_for(int op, bool i64) {
    if (!i64) return op;
    return op == EW_ADD ? EW_ADD_I64 : op == EW_SUB ? EW_SUB_I64 : op == EW_NEG ? EW_NEG_I64 : op;
}
static int add_sub_into(pz_module* M, bool sub, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                        size_t a_cols, size_t a_size, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col,
                        bool i64 = false) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_add/sub(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_add/sub(a)");
    PZ_CHECK_COL(b_col, b_cols, "vec_znx_dft_add/sub(b)");
    PZ_REQUIRE((const void*)res != (const void*)b, "vec_znx_dft_add/sub: res must not alias b (use the *_assign form)");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, a, a_cols, a_size, b, b_cols, b_size));
    const bool a_le_b = a_size <= b_size;
    const int sum = (int)std::min(a_le_b ? a_size : b_size, res_size);
    const int cpy = (int)std::min(a_le_b ? b_size : a_size, res_size);
    PZ_TRY(ew_limbs(M, ew_for(sub ? EW_SUB : EW_ADD, i64), t.dr, (int)res_col, 0, &t.da, (int)a_col, 0, &t.db, (int)b_col, 0, sum));
    if (a_le_b) PZ_TRY(ew_limbs(M, ew_for(sub ? EW_NEG : EW_COPY, i64), t.dr, (int)res_col, sum, &t.db, (int)b_col, sum, nullptr, 0, 0, cpy - sum));
    else PZ_TRY(ew_limbs(M, EW_COPY, t.dr, (int)res_col, sum, &t.da, (int)a_col, sum, nullptr, 0, 0, cpy - sum));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, cpy, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - cpy));
    return tri_out(M, t);
}
int pz_vec_znx_dft_add_into(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                            size_t a_cols, size_t a_size, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, false, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}
int pz_vec_znx_dft_sub(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a, size_t a_cols,
                       size_t a_size, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, true, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}

// res (op)= a over limb ranges; shifts express add_scaled_assign
static int assign_op(pz_module* M, int op_res_a /*EW_ADD: res+a, EW_SUB: res-a, -EW_SUB: a-res*/, double* res, size_t res_cols,
                     size_t res_size, size_t res_col, const double* a, size_t a_cols, size_t a_size, size_t a_col, int res_shift,
                     int a_shift, int nl, bool negate_tail, bool i64 = false) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_*_assign(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_*_assign(a)");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, a, a_cols, a_size, nullptr, 0, 0));
    if (op_res_a == EW_ADD || op_res_a == EW_SUB)
        PZ_TRY(ew_limbs(M, ew_for(op_res_a, i64), t.dr, (int)res_col, res_shift, &t.dr, (int)res_col, res_shift, &t.da, (int)a_col, a_shift, nl));
    else
        PZ_TRY(ew_limbs(M, ew_for(EW_SUB, i64), t.dr, (int)res_col, res_shift, &t.da, (int)a_col, a_shift, &t.dr, (int)res_col, res_shift, nl));
    if (negate_tail)
        PZ_TRY(ew_limbs(M, ew_for(EW_NEG, i64), t.dr, (int)res_col, nl, &t.dr, (int)res_col, nl, nullptr, 0, 0, (int)res_size - nl));
    return tri_out(M, t);
}
int pz_vec_znx_dft_add_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                              size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_ADD, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, 0, 0, (int)std::min(a_size, res_size), false);
}
int pz_vec_znx_dft_sub_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                              size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_SUB, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, 0, 0, (int)std::min(a_size, res_size), false);
}
int pz_vec_znx_dft_sub_negate_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                                     size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, -EW_SUB, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, 0, 0, (int)std::min(a_size, res_size), true);
}
// ---- i64 VecZnx limb-wise family (hal_impl.rs:59 add_into, :65 add_assign, :90 sub, :96 sub_assign, :101 sub_negate_assign,
//      :126 negate, :131 negate_assign, :289 copy, :34 zero): SURVEY.md 8f rank 3, so that ciphertexts stay on the device
//      between the hot-path operations.  Same limb-range rules as the DFT-domain family above, wrapping i64 arithmetic.
int pz_vec_znx_add_into(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                        size_t a_size, size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, false, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, (const double*)b, b_cols,
                        b_size, b_col, true);
}
int pz_vec_znx_sub(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                   size_t a_size, size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, true, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, (const double*)b, b_cols,
                        b_size, b_col, true);
}
int pz_vec_znx_add_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                          size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_ADD, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, 0, 0,
                     (int)std::min(a_size, res_size), false, true);
}
int pz_vec_znx_sub_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                          size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_SUB, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, 0, 0,
                     (int)std::min(a_size, res_size), false, true);
}
int pz_vec_znx_sub_negate_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                                 size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, -EW_SUB, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, 0, 0,
                     (int)std::min(a_size, res_size), true, true);
}
// res = -a over the common limbs, zero beyond (negate.rs:6-29); copy: res = a, zero beyond (copy.rs)
static int negate_or_copy(pz_module* M, int op, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                          size_t a_cols, size_t a_size, size_t a_col) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_negate/copy(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_negate/copy(a)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    const int mn = (int)std::min(res_size, a_size);
    PZ_TRY(ew_limbs(M, op, t.dr, (int)res_col, 0, &t.da, (int)a_col, 0, nullptr, 0, 0, mn));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, mn, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - mn));
    return tri_out(M, t);
}
int pz_vec_znx_negate(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                      size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return negate_or_copy(M, EW_NEG_I64, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_copy(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                    size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(!((const void*)res == (const void*)a && res_col != a_col), "vec_znx_copy: column-to-column copy inside one container is not supported");
    if ((const void*)res == (const void*)a) return PZ_OK;
    return negate_or_copy(M, EW_COPY, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_negate_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_negate_assign(res)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, nullptr, 0, 0, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_NEG_I64, t.dr, (int)res_col, 0, &t.dr, (int)res_col, 0, nullptr, 0, 0, (int)res_size));
    return tri_out(M, t);
}
int pz_vec_znx_zero(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_zero(res)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, nullptr, 0, 0, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, 0, nullptr, 0, 0, nullptr, 0, 0, (int)res_size));
    return tri_out(M, t);
}
int pz_vec_znx_dft_add_scaled_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                                     size_t a_cols, size_t a_size, size_t a_col, int64_t a_scale) {
    PZ_ENTER(M);
    int rs = 0, as_ = 0, nl;  // vec_znx_dft.rs:93-128
    if (a_scale > 0) {
        size_t shift = std::min<size_t>((size_t)a_scale, a_size);
        size_t mn = std::min(a_size, res_size);
        nl = (int)(mn > shift ? mn - shift : 0);
        as_ = (int)shift;
    } else if (a_scale < 0) {
        size_t shift = std::min<size_t>((size_t)(-a_scale), res_size);
        nl = (int)std::min(a_size, res_size - shift);
        rs = (int)shift;
    } else {
        nl = (int)std::min(a_size, res_size);
    }
    return assign_op(M, EW_ADD, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, rs, as_, nl, false);
}

int pz_vec_znx_dft_copy(pz_module* M, size_t step, size_t offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                        const double* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(step > 0, "vec_znx_dft_copy: step must be > 0");
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_copy(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_copy(a)");
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_dft_copy: res must not alias a");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, a, a_cols, a_size, nullptr, 0, 0));
    const int steps = (int)((a_size + step - 1) / step);
    const int min_steps = std::min((int)res_size, steps);
    int nv = 0;
    if (offset < a_size) nv = std::min(min_steps, (int)((a_size - offset + step - 1) / step));
    // strided source limbs: limb stride of the source is step*cols*n
    PZ_TRY(launch_ew(M, EW_COPY, poly_ptr(M, t.dr, (int)res_col, 0), 0, limb_stride(M, t.dr), poly_ptr(M, t.da, (int)a_col, (int)offset),
                     0, (long long)step * limb_stride(M, t.da), nullptr, 0, 0, nv, 1));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, nv, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - nv));
    return tri_out(M, t);
}
int pz_vec_znx_dft_zero(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_zero(res)");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, nullptr, 0, 0, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, 0, nullptr, 0, 0, nullptr, 0, 0, (int)res_size));
    return tri_out(M, t);
}

// ------------------------------------------------------------------------------
// public: SVP
// ------------------------------------------------------------------------------
int pz_svp_prepare(pz_module* M, double* res, size_t res_cols, size_t res_col, const int64_t* a, size_t a_cols, size_t a_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "svp_prepare(res)");
    PZ_CHECK_COL(a_col, a_cols, "svp_prepare(a)");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, 1), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, 1), true, true, M));
    cplx* T;
    PZ_TRY(need_T(M, 1, &T));
    DV dr{sr.dev, 0, (int)res_cols, 1}, da{sa.dev, 0, (int)a_cols, 1};
    PZ_TRY(dev_dft_apply(M, 1, 1, 0, dr, (int)res_col, da, (int)a_col, 1, nullptr, T));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_svp_apply_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol, size_t a_cols,
                     size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "svp_apply_dft(res)");
    PZ_CHECK_COL(a_col, a_cols, "svp_apply_dft(ppol)");
    PZ_CHECK_COL(b_col, b_cols, "svp_apply_dft(b)");
    Stage sr, sp, sb;
    PZ_TRY(sp.in(ppol, vbytes(M, a_cols, 1), true, false, M));
    PZ_TRY(sb.in(b, vbytes(M, b_cols, b_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));
    const int min_size = (int)std::min(res_size, b_size);
    cplx* T;
    PZ_TRY(need_T(M, min_size, &T));
    // svp.rs:21-54: FFT of limbs < min_size times ppol, the rest zero
    DV dr{sr.dev, 0, (int)res_cols, min_size}, db{sb.dev, 0, (int)b_cols, (int)b_size};
    const cplx* mul = reinterpret_cast<const cplx*>((const double*)sp.dev + (size_t)M->n * a_col);
    PZ_TRY(dev_dft_apply(M, 1, 1, 0, dr, (int)res_col, db, (int)b_col, 1, mul, T));
    DV drf{sr.dev, 0, (int)res_cols, (int)res_size};
    PZ_TRY(ew_limbs(M, EW_ZERO, drf, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));
    const bool host = sr.owned || sp.owned || sb.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sp.finish());
    PZ_TRY(sb.finish());
    return finish_call(M, host);
}

static int svp_dft_to_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol,
                          size_t a_cols, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_CHECK_COL(res_col, res_cols, "svp_apply_dft_to_dft(res)");
    PZ_CHECK_COL(a_col, a_cols, "svp_apply_dft_to_dft(ppol)");
    PZ_CHECK_COL(b_col, b_cols, "svp_apply_dft_to_dft(b)");
    Stage sp;
    PZ_TRY(sp.in(ppol, vbytes(M, a_cols, 1), true, false, M));
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, b, b_cols, b_size, nullptr, 0, 0));
    const int min_size = (int)std::min(res_size, b_size);
    // res[j] = ppol * b[j]: the prepared polynomial is the same for every limb (limb stride 0)
    PZ_TRY(launch_ew(M, EW_CMUL, poly_ptr(M, t.dr, (int)res_col, 0), 0, limb_stride(M, t.dr),
                     (const double*)sp.dev + (size_t)M->n * a_col, 0, 0, poly_ptr(M, t.da, (int)b_col, 0), 0, limb_stride(M, t.da),
                     min_size, 1));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));
    t.host = t.host || sp.owned;
    PZ_TRY(sp.finish());
    return tri_out(M, t);
}
int pz_svp_apply_dft_to_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol,
                            size_t a_cols, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return svp_dft_to_dft(M, res, res_cols, res_size, res_col, ppol, a_cols, a_col, b, b_cols, b_size, b_col);
}
int pz_svp_apply_dft_to_dft_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol,
                                   size_t a_cols, size_t a_col) {
    PZ_ENTER(M);
    return svp_dft_to_dft(M, res, res_cols, res_size, res_col, ppol, a_cols, a_col, res, res_cols, res_size, res_col);
}

// ------------------------------------------------------------------------------
// public: VMP
// ------------------------------------------------------------------------------
size_t pz_vmp_prepare_tmp_bytes(const pz_module* M, size_t, size_t, size_t, size_t) { return M ? (size_t)M->n * 8 : 0; }
size_t pz_vmp_apply_dft_to_dft_tmp_bytes(const pz_module*, size_t, size_t a_size, size_t b_rows, size_t b_cols_in, size_t, size_t) {
    return (16 + 8 * std::min(a_size, b_rows) * b_cols_in) * 8;  // vmp.rs:132-135
}
size_t pz_vmp_apply_dft_tmp_bytes(const pz_module* M, size_t res_size, size_t a_size, size_t b_rows, size_t b_cols_in,
                                  size_t b_cols_out, size_t b_size) {
    // hal_impl/family_common.rs:3-15
    return pz_bytes_of_vec_znx_dft(M ? M->n : 0, b_cols_in, std::min(a_size, b_rows)) +
           pz_vmp_apply_dft_to_dft_tmp_bytes(M, res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size);
}

int pz_vmp_prepare(pz_module* M, double* pmat, const int64_t* mat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    PZ_ENTER(M);
    PZ_TRY(forget_host_key(M, (const void*)pmat));   // (a device mirror of this host buffer would be stale)
    host_key_invalidate(pmat, rows * cols_in * cols_out * size * (size_t)M->n * 8);
    const size_t npolys = rows * cols_in * cols_out * size;
    Stage sp, sm;
    PZ_TRY(sm.in(mat, npolys * M->n * 8, true, false, M));
    PZ_TRY(sp.in(pmat, npolys * M->n * 8, false, true, M));
    // Device VmpPMat = spectra of the MatZnx polynomials in MatZnx order (entry (r, c) at (r*ncols + c)*n):
    // one FFT per matrix entry (vmp.rs:52-93) and no block re-layout.
    const size_t group = 256;
    cplx* T;
    PZ_TRY(need_T(M, std::min(npolys, group), &T));
    const long long n = (long long)M->n;
    for (size_t p0 = 0; p0 < npolys; p0 += group) {
        const int cnt = (int)std::min(group, npolys - p0);
        PolyMap sm_{cnt, 1, 0, n, 0, (long long)p0 * n};
        PolyMap dm_{cnt, 1, 0, n, 0, (long long)p0 * n};
        PZ_TRY(launch_fwd_pass1(M, cnt, (const long long*)sm.dev, sm_, T));
        PZ_TRY(launch_fwd_pass2(M, cnt, T, (double*)sp.dev, dm_, nullptr));
    }
    const bool host = sp.owned || sm.owned;
    PZ_TRY(sp.finish());
    PZ_TRY(sm.finish());
    const int rc = finish_call(M, host);
    // published once more now that the host bytes HAVE changed: a sibling that looked the key up between the first
    // publication and the D2H copy above re-mirrored the old bytes with an epoch >= that publication
    host_key_invalidate(pmat, npolys * (size_t)M->n * 8);
    return rc;
}

int pz_vmp_zero(pz_module* M, double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    PZ_ENTER(M);
    PZ_TRY(forget_host_key(M, (const void*)pmat));
    const size_t bytes = rows * cols_in * cols_out * size * M->n * 8;
    host_key_invalidate(pmat, bytes);
    if (is_device_ptr(pmat)) PZ_HIP(hipMemsetAsync(pmat, 0, bytes, M->stream));
    else { memset(pmat, 0, bytes); host_key_invalidate(pmat, bytes); }   // (again, after the write: see pz_vmp_prepare)
    return PZ_OK;
}

static int vmp_checks(pz_module* M, size_t res_cols, size_t a_cols, size_t cols_in, size_t cols_out) {
    (void)M;
    PZ_REQUIRE(res_cols == cols_out, "vmp_apply: res.cols %zu != pmat.cols_out %zu", res_cols, cols_out);
    PZ_REQUIRE(a_cols == cols_in, "vmp_apply: a.cols %zu != pmat.cols_in %zu", a_cols, cols_in);
    return PZ_OK;
}

int pz_vmp_apply_dft_to_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, const double* a, size_t a_cols, size_t a_size,
                            const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size, size_t limb_offset) {
    PZ_ENTER(M);
    PZ_TRY(vmp_checks(M, res_cols, a_cols, cols_in, cols_out));
    PZ_REQUIRE((const void*)res != (const void*)a, "vmp_apply_dft_to_dft: res must not alias a");
    Stage sr, sa, sp;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sp.in(pmat, rows * cols_in * cols_out * size * M->n * 8, true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), false, true, M));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_vmp(M, 1, dr, da, (const double*)sp.dev, (int)rows, (int)cols_in, (int)cols_out, (int)size, (int)limb_offset));
    const bool host = sr.owned || sa.owned || sp.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    PZ_TRY(sp.finish());
    return finish_call(M, host);
}

int pz_vmp_apply_dft_to_dft_batched(pz_module* M, size_t batch, double* res, size_t res_cols, size_t res_size, const double* a,
                                    size_t a_cols, size_t a_size, const double* pmat, size_t rows, size_t cols_in, size_t cols_out,
                                    size_t size, size_t limb_offset) {
    PZ_ENTER(M);
    PZ_TRY(vmp_checks(M, res_cols, a_cols, cols_in, cols_out));
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a) && is_device_ptr(pmat), "batched entry points take device pointers");
    DV dr{res, (long long)(M->n * res_cols * res_size), (int)res_cols, (int)res_size};
    DV da{(void*)a, (long long)(M->n * a_cols * a_size), (int)a_cols, (int)a_size};
    return dev_vmp(M, (int)batch, dr, da, pmat, (int)rows, (int)cols_in, (int)cols_out, (int)size, (int)limb_offset);
}

int pz_vmp_apply_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, const int64_t* a, size_t a_cols, size_t a_size,
                     const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    PZ_ENTER(M);
    PZ_REQUIRE(res_cols == cols_out, "vmp_apply_dft: res.cols %zu != pmat.cols_out %zu", res_cols, cols_out);
    PZ_REQUIRE(a_cols <= cols_in, "vmp_apply_dft: a.cols %zu > pmat.cols_in %zu", a_cols, cols_in);
    Stage sr, sa, sp;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sp.in(pmat, rows * cols_in * cols_out * size * M->n * 8, true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), false, true, M));
    // family_common.rs:17-54: DFT of a right-aligned into cols_in columns (leading columns zero), then the product
    const size_t sz = std::min(a_size, rows);
    const size_t adft_bytes = vbytes(M, cols_in, sz);
    PZ_TRY(ws_reserve(M, adft_bytes + sz * a_cols * M->m * sizeof(cplx)));
    char* wbase = (char*)M->ws;
    double* adft; cplx* T;
    PZ_TRY(ws_take(M, wbase, adft_bytes, &adft));
    PZ_TRY(ws_take(M, wbase, sz * a_cols * M->m * sizeof(cplx), &T));
    PZ_HIP(hipMemsetAsync(adft, 0, adft_bytes, M->stream));
    DV dad{adft, 0, (int)cols_in, (int)sz}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_dft_apply(M, 1, 1, 0, dad, (int)(cols_in - a_cols), da, 0, (int)a_cols, nullptr, T));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size};
    PZ_TRY(dev_vmp(M, 1, dr, dad, (const double*)sp.dev, (int)rows, (int)cols_in, (int)cols_out, (int)size, 0));
    const bool host = sr.owned || sa.owned || sp.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    PZ_TRY(sp.finish());
    return finish_call(M, host);
}

// ------------------------------------------------------------------------------
// public: VecZnxBig
// ------------------------------------------------------------------------------
size_t pz_vec_znx_big_normalize_tmp_bytes(const pz_module* M) { return M ? 3 * (size_t)M->n * 8 : 0; }  // normalize.rs:13-15

static int normalize_checks(size_t res_col, size_t res_cols, size_t a_col, size_t a_cols, size_t res_base2k, size_t a_base2k) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_big_normalize(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_big_normalize(a)");
    PZ_REQUIRE(res_base2k >= 1 && res_base2k <= 63 && a_base2k >= 1 && a_base2k <= 63, "vec_znx_big_normalize: base2k out of range");
    return PZ_OK;
}

static int normalize_impl(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                          size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    PZ_TRY(normalize_checks(res_col, res_cols, a_col, a_cols, res_base2k, a_base2k));
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_big_normalize: res must not alias a");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_normalize(M, 1, dr, (int)res_base2k, res_offset, (int)res_col, da, (int)a_base2k, (int)a_col));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_vec_znx_big_normalize(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                             size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    PZ_ENTER(M);
    return normalize_impl(M, res, res_cols, res_size, res_base2k, res_offset, res_col, a, a_cols, a_size, a_base2k, a_col);
}
// vec_znx_normalize (hal_impl.rs:41): with ScalarBig = i64 (poulpy-cpu-ref/src/fft64/module.rs:40-43) it is the function
// vec_znx_big_normalize forwards to (reference/fft64/vec_znx_big.rs:241-278 -> vec_znx/normalize.rs:18-48)
size_t pz_vec_znx_normalize_tmp_bytes(const pz_module* M) { return pz_vec_znx_big_normalize_tmp_bytes(M); }
int pz_vec_znx_normalize(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                         size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    PZ_ENTER(M);
    return normalize_impl(M, res, res_cols, res_size, res_base2k, res_offset, res_col, a, a_cols, a_size, a_base2k, a_col);
}
// vec_znx_normalize_assign (hal_impl.rs:55; reference/vec_znx/normalize.rs:403-425): in place, same base == the out-of-place
// same-base normalization of a copy of the column
int pz_vec_znx_normalize_assign(pz_module* M, size_t base2k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_normalize_assign(res)");
    PZ_REQUIRE(base2k >= 1 && base2k <= 63, "vec_znx_normalize_assign: base2k out of range");
    if (size == 0) return PZ_OK;
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
    DV dr{sr.dev, 0, (int)cols, (int)size};
    PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
    DV tv{M->ws, 0, 1, (int)size};
    PZ_TRY(dev_normalize(M, 1, dr, (int)base2k, 0, (int)col, tv, (int)base2k, 0));
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

// vec_znx_lsh (hal_impl.rs:165), vec_znx_rsh (:137), vec_znx_lsh_assign (:221): reference/vec_znx/shift.rs:68-135, :245-342,
// :16-66 walk the limbs exactly as vec_znx_normalize does at equal bases with res_offset = +k / -k (same step functions, same
// ranges; pinned on the literal restatement by tests/test_oracle_exact.py P10), so they run on the normalize kernels.
size_t pz_vec_znx_lsh_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }  // shift.rs:12-14
int pz_vec_znx_lsh(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                   size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(k <= ((size_t)1 << 40), "vec_znx_lsh: shift out of range");
    return normalize_impl(M, res, res_cols, res_size, base2k, (int64_t)k, res_col, a, a_cols, a_size, base2k, a_col);
}
int pz_vec_znx_rsh(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                   size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(k <= ((size_t)1 << 40), "vec_znx_rsh: shift out of range");
    return normalize_impl(M, res, res_cols, res_size, base2k, -(int64_t)k, res_col, a, a_cols, a_size, base2k, a_col);
}
int pz_vec_znx_lsh_assign(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_lsh_assign(res)");
    PZ_REQUIRE(base2k >= 1 && base2k <= 63, "vec_znx_lsh_assign: base2k out of range");
    PZ_REQUIRE(k <= ((size_t)1 << 40), "vec_znx_lsh_assign: shift out of range");
    if (size == 0) return PZ_OK;
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
    DV dr{sr.dev, 0, (int)cols, (int)size};
    PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
    DV tv{M->ws, 0, 1, (int)size};
    PZ_TRY(dev_normalize(M, 1, dr, (int)base2k, (long long)k, (int)col, tv, (int)base2k, 0));
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

int pz_vec_znx_big_normalize_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k,
                                     int64_t res_offset, size_t res_col, const int64_t* a, size_t a_cols, size_t a_size,
                                     size_t a_base2k, size_t a_col) {
    PZ_ENTER(M);
    PZ_TRY(normalize_checks(res_col, res_cols, a_col, a_cols, res_base2k, a_base2k));
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a), "batched entry points take device pointers");
    DV dr{res, (long long)(M->n * res_cols * res_size), (int)res_cols, (int)res_size};
    DV da{(void*)a, (long long)(M->n * a_cols * a_size), (int)a_cols, (int)a_size};
    return dev_normalize(M, (int)batch, dr, (int)res_base2k, res_offset, (int)res_col, da, (int)a_base2k, (int)a_col);
}

int pz_vec_znx_big_add_small_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                                    size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_big_add_small_assign(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_big_add_small_assign(a)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_ADD_I64, t.dr, (int)res_col, 0, &t.dr, (int)res_col, 0, &t.da, (int)a_col, 0, (int)std::min(a_size, res_size)));
    return tri_out(M, t);
}

// vec_znx_automorphism (hal_impl.rs:236) and vec_znx_big_automorphism (:517) are the same operation on i64 containers
// (fft64/vec_znx_big.rs:144-170 re-types the big container and calls the VecZnx function)
static int automorphism_into(pz_module* M, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                             const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_automorphism(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_automorphism(a)");
    PZ_REQUIRE((p & 1) != 0, "vec_znx_automorphism: the Galois element must be odd");
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_automorphism: res must not alias a (use the *_assign form)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    const int min_size = (int)std::min(res_size, a_size);
    const long long n = (long long)M->n;
    PolyMap sm{std::max(min_size, 1), 1, 0, (long long)a_cols * n, 0, n * (long long)a_col};
    PolyMap dm{std::max(min_size, 1), 1, 0, (long long)res_cols * n, 0, n * (long long)res_col};
    PZ_TRY(launch_automorphism(M, min_size, (const long long*)t.da.p, sm, (long long*)t.dr.p, dm, inv_mod_2n(p, n), 1));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));  // automorphism.rs:32-34
    return tri_out(M, t);
}
static int automorphism_assign(pz_module* M, int64_t p, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_CHECK_COL(col, cols, "vec_znx_automorphism_assign(res)");
    PZ_REQUIRE((p & 1) != 0, "vec_znx_automorphism_assign: the Galois element must be odd");
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    if (size > 0) {
        // the reference permutes through one polynomial of scratch (automorphism.rs:37-51); here: the column's limbs are
        // copied to the workspace and gathered back
        PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
        DV dr{sr.dev, 0, (int)cols, (int)size};
        PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
        PolyMap sm{(int)size, 1, 0, n, 0, 0};
        PolyMap dm{(int)size, 1, 0, (long long)cols * n, 0, n * (long long)col};
        PZ_TRY(launch_automorphism(M, (int)size, (const long long*)M->ws, sm, (long long*)sr.dev, dm, inv_mod_2n(p, n), 1));
    }
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}
size_t pz_vec_znx_automorphism_assign_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }      // automorphism.rs:6-8
size_t pz_vec_znx_big_automorphism_assign_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }  // vec_znx_big.rs:140-142
int pz_vec_znx_automorphism(pz_module* M, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                            size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return automorphism_into(M, p, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_automorphism_assign(pz_module* M, int64_t p, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    return automorphism_assign(M, p, res, cols, size, col);
}
int pz_vec_znx_big_automorphism(pz_module* M, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return automorphism_into(M, p, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_big_automorphism_assign(pz_module* M, int64_t p, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    return automorphism_assign(M, p, res, cols, size, col);
}

// vec_znx_rotate (hal_impl.rs:225) / vec_znx_rotate_assign (:232): res = X^k * a (reference/znx/rotate.rs:3-27), limbs of res
// beyond a.size zeroed (vec_znx/rotate.rs:33-35)
size_t pz_vec_znx_rotate_assign_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }
int pz_vec_znx_rotate(pz_module* M, int64_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                      size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_rotate(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_rotate(a)");
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_rotate: res must not alias a (use the *_assign form)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    const int min_size = (int)std::min(res_size, a_size);
    const long long n = (long long)M->n;
    PolyMap sm{std::max(min_size, 1), 1, 0, (long long)a_cols * n, 0, n * (long long)a_col};
    PolyMap dm{std::max(min_size, 1), 1, 0, (long long)res_cols * n, 0, n * (long long)res_col};
    PZ_TRY(launch_rotate(M, min_size, (const long long*)t.da.p, sm, (long long*)t.dr.p, dm, 0, std::max(min_size, 1), nullptr, 0, 0, (long long)k));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));
    return tri_out(M, t);
}
int pz_vec_znx_rotate_assign(pz_module* M, int64_t k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_rotate_assign(res)");
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    if (size > 0) {
        PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
        DV dr{sr.dev, 0, (int)cols, (int)size};
        PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
        PolyMap sm{(int)size, 1, 0, n, 0, 0};
        PolyMap dm{(int)size, 1, 0, (long long)cols * n, 0, n * (long long)col};
        PZ_TRY(launch_rotate(M, (int)size, (const long long*)M->ws, sm, (long long*)sr.dev, dm, 0, (int)size, nullptr, 0, 0, (long long)k));
    }
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

// vec_znx_rsh_assign (hal_impl.rs:217; reference/vec_znx/shift.rs:186-243)
size_t pz_vec_znx_rsh_tmp_bytes(const pz_module* M) { return M ? 2 * (size_t)M->n * 8 : 0; }  // shift.rs: carry + one polynomial
int pz_vec_znx_rsh_assign(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_rsh_assign(res)");
    PZ_REQUIRE(base2k >= 1 && base2k <= 63, "vec_znx_rsh_assign: base2k out of range");
    PZ_REQUIRE(k <= base2k * size, "vec_znx_rsh_assign: shift beyond the precision of res");
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    PZ_TRY(launch_rsh(M, 1, (long long*)sr.dev, 0, (int)cols, (int)size, (int)col, 1, (int)base2k, (int)k));
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

}  // extern "C"
