// api_cnv.hip — C ABI of the convolution family (HalImpl cnv_*, poulpy-hal/src/oep/hal_impl.rs:670-754; reference
// poulpy-cpu-ref/src/reference/fft64/convolution.rs) and of GLWE tensoring on device-resident batches
// (poulpy-core/src/operations/glwe.rs:609-913), SURVEY.md §8f rank 4 / BASELINE configs[4].
#include "api_common.hpp"
#include "api_glwe.hpp"

// CnvPVecL / CnvPVecR bytes in this backend: polynomial (col, limb) = its spectrum in device order at (col*size + limb)*n doubles.
static inline size_t cnv_bytes(const pz_module* M, size_t cols, size_t size) { return (size_t)M->n * cols * size * 8; }

// convolution.rs:35-80: limbs < min_size - 1 plain forward transforms, limb min_size - 1 with `mask` applied to the coefficients,
// limbs >= min_size zero.  `a` batched with stride a_bs (scalars), res with stride res_bs.
static int dev_cnv_prepare(pz_module* M, int batch, double* res, long long res_bs, int cols, int res_size, const int64_t* a, long long a_bs,
                           int a_cols, int a_size, long long mask, cplx* T) {
    const long long n = (long long)M->n;
    const int min_size = std::min(res_size, a_size);
    if (min_size > 1) {
        const int nl = min_size - 1;
        PolyMap sm{nl, cols, a_bs, (long long)a_cols * n, n, 0};
        PolyMap dm{nl, cols, res_bs, n, (long long)res_size * n, 0};
        PZ_TRY(launch_fwd_pass1(M, batch * nl * cols, (const long long*)a, sm, T));
        PZ_TRY(launch_fwd_pass2(M, batch * nl * cols, T, res, dm, nullptr));
    }
    if (min_size > 0) {
        const int last = min_size - 1;
        PolyMap sm{1, cols, a_bs, 0, n, (long long)last * a_cols * n};
        PolyMap dm{1, cols, res_bs, 0, (long long)res_size * n, (long long)last * n};
        PZ_TRY(launch_fwd_pass1(M, batch * cols, (const long long*)a, sm, T, false, mask));
        PZ_TRY(launch_fwd_pass2(M, batch * cols, T, res, dm, nullptr));
    }
    for (int c = 0; c < cols && res_size > min_size; ++c)
        PZ_TRY(launch_ew(M, EW_ZERO, res + ((long long)c * res_size + min_size) * n, res_bs, n, nullptr, 0, 0, nullptr, 0, 0, res_size - min_size, batch));
    return PZ_OK;
}

extern "C" {

// hal_defaults/convolution.rs:40-45, :63-68, and cnv_prepare_self_tmp_bytes: one VecZnxDft(1, min(res_size, a_size))
size_t pz_cnv_prepare_left_tmp_bytes(const pz_module* M, size_t res_size, size_t a_size) {
    return M ? (size_t)M->n * std::min(res_size, a_size) * 8 : 0;
}
size_t pz_cnv_prepare_right_tmp_bytes(const pz_module* M, size_t res_size, size_t a_size) { return pz_cnv_prepare_left_tmp_bytes(M, res_size, a_size); }
size_t pz_cnv_prepare_self_tmp_bytes(const pz_module* M, size_t res_size, size_t a_size) { return pz_cnv_prepare_left_tmp_bytes(M, res_size, a_size); }
// convolution.rs:205-208, :261-263, :142-145
size_t pz_cnv_apply_dft_tmp_bytes(const pz_module*, size_t, size_t res_size, size_t a_size, size_t b_size) {
    return 8 * 8 * std::min(res_size, a_size + b_size - 1);
}
size_t pz_cnv_pairwise_apply_dft_tmp_bytes(const pz_module* M, size_t cnv_offset, size_t res_size, size_t a_size, size_t b_size) {
    return pz_cnv_apply_dft_tmp_bytes(M, cnv_offset, res_size, a_size, b_size) + (a_size + b_size) * 8 * 8;
}
size_t pz_cnv_by_const_apply_tmp_bytes(const pz_module*, size_t, size_t res_size, size_t a_size, size_t b_size) {
    return 8 * (std::min(res_size, a_size + b_size - 1) + a_size) * 8;
}

static int cnv_prepare_one(pz_module* M, double* res, size_t res_cols, size_t res_size, const int64_t* a, size_t a_cols, size_t a_size,
                           int64_t mask, double* copy_to) {
    PZ_REQUIRE(a_cols == res_cols, "cnv_prepare: a.cols %zu != res.cols %zu", a_cols, res_cols);   // convolution.rs:46
    Stage sr, sa, sc;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sr.in(res, cnv_bytes(M, res_cols, res_size), false, true, M));
    if (copy_to) PZ_TRY(sc.in(copy_to, cnv_bytes(M, res_cols, res_size), false, true, M));
    cplx* T;
    PZ_TRY(need_T(M, res_cols * std::max<size_t>(std::min(res_size, a_size), 1), &T));
    PZ_TRY(dev_cnv_prepare(M, 1, (double*)sr.dev, 0, (int)res_cols, (int)res_size, (const int64_t*)sa.dev, 0, (int)a_cols, (int)a_size,
                           (long long)mask, T));
    if (copy_to)   // convolution.rs:134-138: right = left (identical data for FFT64)
        PZ_TRY(launch_ew(M, EW_COPY, sc.dev, 0, (long long)M->n, sr.dev, 0, (long long)M->n, nullptr, 0, 0, (int)(res_cols * res_size), 1));
    const bool host = sr.owned || sa.owned || sc.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    PZ_TRY(sc.finish());
    return finish_call(M, host);
}
int pz_cnv_prepare_left(pz_module* M, double* res, size_t res_cols, size_t res_size, const int64_t* a, size_t a_cols, size_t a_size,
                        int64_t mask) {
    PZ_ENTER(M);
    return cnv_prepare_one(M, res, res_cols, res_size, a, a_cols, a_size, mask, nullptr);
}
int pz_cnv_prepare_right(pz_module* M, double* res, size_t res_cols, size_t res_size, const int64_t* a, size_t a_cols, size_t a_size,
                         int64_t mask) {
    PZ_ENTER(M);
    return cnv_prepare_one(M, res, res_cols, res_size, a, a_cols, a_size, mask, nullptr);
}
int pz_cnv_prepare_self(pz_module* M, double* left, double* right, size_t cols, size_t size, const int64_t* a, size_t a_cols, size_t a_size,
                        int64_t mask) {
    PZ_ENTER(M);
    PZ_REQUIRE((const void*)left != (const void*)right, "cnv_prepare_self: left and right must be distinct buffers");
    return cnv_prepare_one(M, left, cols, size, a, a_cols, a_size, mask, right);
}

static int cnv_apply_common(pz_module* M, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                            size_t a_cols, size_t a_size, const double* b, size_t b_cols, size_t b_size, size_t a_i, size_t b_i, int a_j,
                            int b_j) {
    PZ_CHECK_COL(res_col, res_cols, "cnv_apply_dft(res)");
    PZ_CHECK_COL(a_i, a_cols, "cnv_apply_dft(a)");
    PZ_CHECK_COL(b_i, b_cols, "cnv_apply_dft(b)");
    PZ_REQUIRE(a_size > 0 && b_size > 0, "cnv_apply_dft: empty operand");                                // reim4/mod.rs:47-48
    PZ_REQUIRE((const void*)res != (const void*)a && (const void*)res != (const void*)b, "cnv_apply_dft: res must not alias an operand");
    Stage sr, sa, sb;
    PZ_TRY(sa.in(a, cnv_bytes(M, a_cols, a_size), true, false, M));
    if ((const void*)b == (const void*)a) { sb.M = M; sb.dev = sa.dev; }
    else PZ_TRY(sb.in(b, cnv_bytes(M, b_cols, b_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));   // inout: other columns survive
    const size_t bound = a_size + b_size - 1;
    const int min_size = (int)std::min(res_size, bound), offset = (int)std::min(cnv_offset, bound);
    PZ_TRY(launch_cnv_apply(M, 1, (double*)sr.dev, 0, (int)res_cols, (int)res_col, min_size, offset, (const double*)sa.dev, 0, (int)a_size,
                            (int)a_i, a_j, (const double*)sb.dev, 0, (int)b_size, (int)b_i, b_j));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size};
    PZ_TRY(launch_ew(M, EW_ZERO, poly_ptr(M, dr, (int)res_col, min_size), 0, limb_stride(M, dr), nullptr, 0, 0, nullptr, 0, 0,
                     (int)res_size - min_size, 1));                                                     // convolution.rs:256-258
    const bool host = sr.owned || sa.owned || sb.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    PZ_TRY(sb.finish());
    return finish_call(M, host);
}
int pz_cnv_apply_dft(pz_module* M, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                     size_t a_cols, size_t a_size, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return cnv_apply_common(M, cnv_offset, res, res_cols, res_size, res_col, a, a_cols, a_size, b, b_cols, b_size, a_col, b_col, -1, -1);
}
int pz_cnv_pairwise_apply_dft(pz_module* M, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* a, size_t a_cols, size_t a_size, const double* b, size_t b_cols, size_t b_size, size_t col_i,
                              size_t col_j) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col_j, a_cols, "cnv_pairwise_apply_dft(a)");
    PZ_CHECK_COL(col_j, b_cols, "cnv_pairwise_apply_dft(b)");
    const int j = col_i == col_j ? -1 : (int)col_j;                                                     // convolution.rs:281-284
    return cnv_apply_common(M, cnv_offset, res, res_cols, res_size, res_col, a, a_cols, a_size, b, b_cols, b_size, col_i, col_i, j, j);
}

int pz_cnv_by_const_apply(pz_module* M, size_t cnv_offset, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                          size_t a_cols, size_t a_size, size_t a_col, const int64_t* b, size_t b_len) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "cnv_by_const_apply(res)");
    PZ_CHECK_COL(a_col, a_cols, "cnv_by_const_apply(a)");
    PZ_REQUIRE(a_size > 0 && b_len > 0, "cnv_by_const_apply: empty operand");
    PZ_REQUIRE((const void*)res != (const void*)a, "cnv_by_const_apply: res must not alias a");
    Stage sr, sa, sb;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sb.in(b, b_len * 8, true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));
    const size_t bound = a_size + b_len - 1;
    const int min_size = (int)std::min(res_size, bound), offset = (int)std::min(cnv_offset, bound);
    PZ_TRY(launch_cnv_by_const(M, (long long*)sr.dev, (int)res_cols, (int)res_col, min_size, offset, (const long long*)sa.dev, (int)a_cols,
                               (int)a_size, (int)a_col, (const long long*)sb.dev, (int)b_len));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size};
    PZ_TRY(launch_ew(M, EW_ZERO, poly_ptr(M, dr, (int)res_col, min_size), 0, limb_stride(M, dr), nullptr, 0, 0, nullptr, 0, 0,
                     (int)res_size - min_size, 1));
    const bool host = sr.owned || sa.owned || sb.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    PZ_TRY(sb.finish());
    return finish_call(M, host);
}

// ---------------------------------------------------------------------------------------------------------------------
// GLWE tensoring on `batch` device-resident ciphertext pairs (poulpy-core/src/operations/glwe.rs:700-807 glwe_tensor_apply,
// :809-913 _add_assign, :609-698 glwe_tensor_square_apply).  res: batch GLWETensors = VecZnx((rank+1)(rank+2)/2, res_size);
// column of the pair (i, j >= i) = i cols - i (i + 1) / 2 + j.
// Composition, every step one batched launch: prepare (forward transforms of all columns x limbs, bottom limb masked) | per product
// term: limb convolution in the DFT domain -> inverse transform -> normalize with cnv_offset_lo into a one-column temporary
// -> the reference's copy / negate / add / sub into the tensor columns.
// ---------------------------------------------------------------------------------------------------------------------
static inline long long msb_mask_bottom_limb(size_t base2k, size_t k) {   // operations/glwe.rs:921-926
    const size_t r = k % base2k;
    return r == 0 ? -1ll : (long long)(~0ull << (base2k - r));
}
static inline size_t tensor_dft_size(size_t full, size_t res_size, size_t res_base2k, size_t in_base2k, long long res_offset) {   // :929-957
    long long ob = res_offset % (long long)in_base2k;
    if (res_offset < 0 && ob != 0) ob += (long long)in_base2k;
    return std::min(full, (res_size * res_base2k + (size_t)ob + in_base2k - 1) / in_base2k);
}
struct TensorPlan {
    int cols, tcols, a_size, b_size, res_size, dft_size, hi;
    long long lo;
    size_t prep_a, prep_b, res_dft, tmp, diag, T, d16, per_ct;
};
static int tensor_plan(const pz_module* M, const pz_glwe_tensor_params* p, int mode, TensorPlan& t) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(mode >= PZ_TENSOR_APPLY && mode <= PZ_TENSOR_SQUARE, "glwe_tensor_apply: unknown mode");
    PZ_REQUIRE(p->rank >= 1 && p->a_size >= 1 && p->res_size >= 1 && (mode == PZ_TENSOR_SQUARE || p->b_size >= 1), "glwe_tensor_apply: empty shape");
    PZ_REQUIRE(p->ab_base2k >= 1 && p->ab_base2k <= 63 && p->res_base2k >= 1 && p->res_base2k <= 63, "glwe_tensor_apply: base2k out of range");
    const size_t ab = p->ab_base2k, bsz = mode == PZ_TENSOR_SQUARE ? p->a_size : p->b_size;
    const size_t bk = mode == PZ_TENSOR_SQUARE ? p->a_effective_k : p->b_effective_k;
    PZ_REQUIRE((p->a_effective_k + ab - 1) / ab == p->a_size && (bk + ab - 1) / ab == bsz,
               "glwe_tensor_apply: effective_k.div_ceil(base2k) must equal the size");                  // :721-722
    t.cols = (int)p->rank + 1;
    t.tcols = t.cols * (t.cols + 1) / 2;
    t.a_size = (int)p->a_size; t.b_size = (int)bsz; t.res_size = (int)p->res_size;
    if (p->cnv_offset < ab) { t.hi = 0; t.lo = -(long long)(ab - (p->cnv_offset % ab)); }               // :733-737
    else { const size_t q = p->cnv_offset / ab; t.hi = (int)(q ? q - 1 : 0); t.lo = (long long)(p->cnv_offset % ab); }
    PZ_REQUIRE((size_t)t.hi < p->a_size + bsz, "glwe_tensor_apply: cnv_offset beyond the product");
    t.dft_size = (int)tensor_dft_size(p->a_size + bsz - (size_t)t.hi, p->res_size, p->res_base2k, ab, t.lo);
    const size_t n8 = (size_t)M->n * 8;
    t.prep_a = n8 * t.cols * t.a_size;
    t.prep_b = n8 * t.cols * t.b_size;
    t.res_dft = n8 * std::max(t.dft_size, 1);
    t.tmp = n8 * t.res_size;
    t.diag = mode == PZ_TENSOR_SQUARE ? n8 * t.cols * t.res_size : 0;
    // (k_mid_cnv3 leaves the three result sets of a rank-1 tensoring side by side: 3 x min(dft_size, a_size + b_size - 1) polynomials)
    t.T = (size_t)M->m * sizeof(cplx) * std::max({t.cols * t.a_size, t.cols * t.b_size, t.dft_size, 1, t.cols == 2 ? 3 * std::min(t.dft_size, t.a_size + t.b_size - 1) : 0});
    // 16-bit side copies of the diagonal terms' digits for the pairwise tails (apply / square with base2k <= 16; TailD16)
    t.d16 = (mode != PZ_TENSOR_APPLY_ADD_ASSIGN && p->res_base2k <= 16 && p->res_base2k == p->ab_base2k) ? (size_t)M->n * 2 * t.cols * t.res_size : 0;
    t.per_ct = t.prep_a + t.prep_b + t.res_dft + t.tmp + t.diag + t.T + t.d16;
    return PZ_OK;
}
static size_t tensor_chunk(const pz_module* M, const TensorPlan& t, size_t batch) {
    if (M->chunk) return std::min(M->chunk, batch);
    size_t c = ((size_t)24 << 30) / std::max<size_t>(t.per_ct, 1);   // as the GLWE pipeline: ~24 GiB of the 288 (at 16 GiB the bench's 256 pairs of 16 limbs split 222 + 34)
    return std::min(std::max<size_t>(c, 1), batch);
}
size_t pz_glwe_tensor_apply_workspace_bytes(const pz_module* M, const pz_glwe_tensor_params* p, int mode, size_t batch) {
    TensorPlan t;
    if (!M || tensor_plan(M, p, mode, t) != PZ_OK) return 0;
    return tensor_chunk(M, t, batch) * (t.per_ct + 6 * 256);
}

// One wave of a GLWE tensoring (glwe_tensor_apply / _add_assign / _square, poulpy-core operations/glwe.rs:609-913): `nb` pairs whose operands have
// been prepared, the product terms (i, j) one at a time into the tensor columns.  State shared by the steps below; the entry point owns the loop
// over waves.
struct TensorWave {
    pz_module* M;
    const pz_glwe_tensor_params* p;
    TensorPlan t;
    bool square, add;
    long long n;
    // workspace of a wave
    double *pa = nullptr, *pb = nullptr, *rd = nullptr;
    int64_t *tmp = nullptr, *diag = nullptr;
    short* d16 = nullptr;   // [diagonal term i][pair][res limb][n] (TensorPlan::d16)
    // the fused multiply + relinearize (pz_glwe_tensor_mul_relinearize_batched): the GLWETensor leaves ONLY as 16-bit digits,
    // t16[tensor column][pair][res limb][n] in the tail's tile order; `rb` is not written
    short* t16 = nullptr;
    long long t16_cs = 0;   // int16 elements between tensor columns
    cplx* T = nullptr;
    long long r_ct = 0, pa_bs = 0, pb_bs = 0, rd_bs = 0, tmp_bs = 0, dg_bs = 0, rls = 0, a_mask = 0, b_mask = 0;
    // the wave
    int nb = 0;
    int64_t* rb = nullptr;
    bool fused = false, all3 = false;
    cplx *ta_main = nullptr, *ta_last = nullptr, *tb_main = nullptr, *tb_last = nullptr;

    int take_workspace(size_t chunk) {
        const size_t s_pa = align256(chunk * t.prep_a), s_pb = square ? 0 : align256(chunk * t.prep_b), s_rd = align256(chunk * t.res_dft);
        const size_t s_tmp = align256(chunk * t.tmp), s_dg = align256(chunk * t.diag), s_T = align256(chunk * t.T), s_d16 = align256(chunk * t.d16);
        PZ_TRY(ws_reserve(M, s_pa + s_pb + s_rd + s_tmp + s_dg + s_T + s_d16));
        char* base = (char*)M->ws;
        PZ_TRY(ws_take(M, base, s_pa, &pa));
        PZ_TRY(ws_take(M, base, s_pb, &pb));
        if (square) pb = pa;                                        // convolution.rs:134-138: right = left for FFT64
        PZ_TRY(ws_take(M, base, s_rd, &rd));
        PZ_TRY(ws_take(M, base, s_tmp, &tmp));
        PZ_TRY(ws_take(M, base, s_dg, &diag));
        PZ_TRY(ws_take(M, base, s_T, &T));
        PZ_TRY(ws_take(M, base, s_d16, &d16));
        if (!s_d16) d16 = nullptr;
        r_ct = n * t.tcols * t.res_size;
        pa_bs = n * t.cols * t.a_size; pb_bs = n * t.cols * t.b_size; rd_bs = n * std::max(t.dft_size, 1); tmp_bs = n * t.res_size;
        dg_bs = n * t.cols * t.res_size;
        a_mask = msb_mask_bottom_limb(p->ab_base2k, p->a_effective_k);
        b_mask = square ? a_mask : msb_mask_bottom_limb(p->ab_base2k, p->b_effective_k);
        rls = (long long)t.tcols * n;   // limb stride of the tensor
        return PZ_OK;
    }
    // pass 1 of one operand's limbs in the pipeline's row-major layout: all limbs but the last, then the last one under its mask
    int prep_T(const int64_t* src, long long ct, int size, long long mask, cplx* region, cplx** mainp, cplx** lastp) {
        *mainp = region;
        *lastp = region + (size_t)nb * (size - 1) * t.cols * (size_t)M->m;
        if (size > 1) {
            PolyMap sm{size - 1, t.cols, ct, (long long)t.cols * n, n, 0};
            PZ_TRY(launch_fwd_pass1(M, nb * (size - 1) * t.cols, (const long long*)src, sm, *mainp, true));
        }
        PolyMap sl{1, t.cols, ct, 0, n, (long long)(size - 1) * t.cols * n};
        return launch_fwd_pass1(M, nb * t.cols, (const long long*)src, sl, *lastp, true, mask);
    }
    // Operands of the wave.  Fused row pass (round 3, m = m1 x 128 plans, one base2k, rank <= 2): pass 1 of the operand limbs, then per term ONE
    // kernel for forward row transform + limb convolution + inverse row transform (k_mid_cnv), the inverse column pass with the normalization
    // and the combination in its stores - instead of forward pass 2 of both operands, k_cnv_apply, inverse pass 2 and inverse pass 1
    // (POULPY_DBG_TENSOR_FUSED=0).  Rank 1, 16 / 8 limbs (round 4): the three terms in ONE launch - operand rows loaded and forward-transformed
    // once, the limb convolution with one operand vector in registers (k_mid_cnv3; POULPY_DBG_TENSOR_ALLTERMS=0: k_mid_cnv per term)
    int prepare(const int64_t* ab, const int64_t* bb) {
        const long long a_ct = n * t.cols * t.a_size, b_ct = n * t.cols * t.b_size;
        static const bool fused_env = (rt_knob("POULPY_DBG_TENSOR_FUSED", 1) != 0);
        static const bool combine_ok_env = (rt_knob("POULPY_DBG_TENSOR_COMBINE", 1) != 0);
        const int bound = t.a_size + t.b_size - 1;
        fused = fused_env && combine_ok_env && p->res_base2k == p->ab_base2k && t.cols <= 3 && t.dft_size >= 1 &&
                mid_cnv_supported(M, t.a_size, t.b_size, std::min(t.dft_size, bound));
        if (fused) {
            PZ_TRY(prep_T(ab, a_ct, t.a_size, a_mask, (cplx*)pa, &ta_main, &ta_last));
            if (square) { tb_main = ta_main; tb_last = ta_last; }
            else PZ_TRY(prep_T(bb, b_ct, t.b_size, b_mask, (cplx*)pb, &tb_main, &tb_last));
        } else {
            PZ_TRY(dev_cnv_prepare(M, nb, pa, pa_bs, t.cols, t.a_size, ab, a_ct, t.cols, t.a_size, a_mask, T));
            if (!square) PZ_TRY(dev_cnv_prepare(M, nb, pb, pb_bs, t.cols, t.b_size, bb, b_ct, t.cols, t.b_size, b_mask, T));
        }
        const int ms_all = std::min(t.dft_size, bound), off_all = std::min(t.hi, bound);
        all3 = fused && mid_cnv3_supported(M, t.cols, t.a_size, t.b_size, ms_all);
        if (all3) PZ_TRY(launch_mid_cnv3(M, nb, ta_main, ta_last, tb_main, tb_last, T, t.a_size, ms_all, off_all));
        return PZ_OK;
    }
    // one product term (i, j): convolution -> inverse transform in place -> normalize(res_base2k, cnv_offset_lo) into `dst` column dcol
    int term(int i, int j, int64_t* dst, long long dst_bs, int dst_cols, int dcol, const NzCombine* cb = nullptr, const TailD16* d16s = nullptr) {
        const int bound = t.a_size + t.b_size - 1;
        const int min_size = std::min(t.dft_size, bound), off = std::min(t.hi, bound);
        if (fused) {
            const cplx* Tt = T;
            if (all3) Tt = T + (size_t)(i == j ? i : 2) * nb * min_size * (size_t)M->m;
            else PZ_TRY(launch_mid_cnv(M, nb, ta_main, ta_last, tb_main, tb_last, T, t.cols, t.a_size, t.b_size, i, i == j ? -1 : j, i, i == j ? -1 : j,
                                       min_size, off));
            // the inverse column pass normalizes on its way out (bit offset, combination and all: TailArgs::nz)
            return launch_inv_tail_nz(M, nb, Tt, min_size, (long long*)dst, dst_bs, dst_cols, t.res_size, dcol, (int)p->res_base2k, t.lo, t.dft_size, cb, d16s);
        }
        PZ_TRY(launch_cnv_apply(M, nb, rd, rd_bs, 1, 0, min_size, off, pa, pa_bs, t.a_size, i, i == j ? -1 : j, pb, pb_bs, t.b_size, i, i == j ? -1 : j));
        if (t.dft_size > min_size)
            PZ_TRY(launch_ew(M, EW_ZERO, rd + (long long)min_size * n, rd_bs, n, nullptr, 0, 0, nullptr, 0, 0, t.dft_size - min_size, nb));
        DV dv{rd, rd_bs, 1, t.dft_size};
        PZ_TRY(dev_idft(M, nb, dv, 0, dv, 0, 1, t.dft_size, T));
        DV out{dst, dst_bs, dst_cols, t.res_size};
        return dev_normalize(M, nb, out, (int)p->res_base2k, t.lo, dcol, dv, (int)p->ab_base2k, 0, cb);
    }
    int64_t* col_ptr(int col) const { return rb + (long long)col * n; }
    int ew_res(int op, int col, const int64_t* x, long long x_bs, long long x_ls, const int64_t* y, long long y_bs, long long y_ls) {
        return launch_ew(M, op, col_ptr(col), r_ct, rls, x, x_bs, x_ls, y, y_bs, y_ls, t.res_size, nb);
    }
    int cidx(int i, int j) const { const int lo_ = std::min(i, j), hi_ = std::max(i, j); return lo_ * t.cols - (lo_ * (lo_ + 1) / 2) + hi_; }

    // One base2k, rank <= 2 (round 3): the digits of a term go straight from the normalizing store into every tensor column that takes them.
    // apply / square: the diagonal terms are stored, each pairwise term then reads the two diagonal columns it belongs to and stores
    // pair - d_i - d_j (mode 5: one write per column, no read-modify-write; wrapping i64: the same digits as the reference's order).
    // add_assign: the diagonal term goes into its column (+=) and out of the cross columns (-=), the pairwise term into its cross column (+=)
    int combine_in_stores() {
        if (!add && t16) {
            const long long ts = t16_cs;   // one tensor column
            for (int i = 0; i < t.cols; ++i) {
                NzCombine cb{1, {0, 0}, {0, 0}};
                TailD16 dw;
                dw.w = t16 + cidx(i, i) * ts; dw.only = true;
                PZ_TRY(term(i, i, rb, r_ct, t.tcols, cidx(i, i), &cb, &dw));
            }
            for (int i = 0; i < t.cols; ++i)
                for (int j = i + 1; j < t.cols; ++j) {
                    NzCombine cb{1, {cidx(i, i), cidx(j, j)}, {5, 5}};
                    TailD16 dr;
                    dr.ra = t16 + cidx(i, i) * ts; dr.rb = t16 + cidx(j, j) * ts; dr.w = t16 + cidx(i, j) * ts; dr.only = true;
                    PZ_TRY(term(i, j, rb, r_ct, t.tcols, cidx(i, j), &cb, &dr));
                }
            return PZ_OK;
        }
        if (!add) {
            // the diagonal launches leave 16-bit copies of their digits beside the tensor columns (base2k <= 16, fused tails): the pairwise
            // launches read those - 2 B per coefficient and diagonal column instead of the 8 B a line of the i64 column costs
            static const bool d16_on = (exp_knob("POULPY_DBG_TENSOR_D16", 1) != 0);
            const bool use16 = d16_on && d16 != nullptr && fused;
            const long long d16_ts = (long long)nb * t.res_size * n;   // one diagonal term's copies
            for (int i = 0; i < t.cols; ++i) {
                NzCombine cb{1, {0, 0}, {0, 0}};
                TailD16 dw;
                dw.w = use16 ? d16 + i * d16_ts : nullptr;
                PZ_TRY(term(i, i, rb, r_ct, t.tcols, cidx(i, i), &cb, use16 ? &dw : nullptr));
            }
            for (int i = 0; i < t.cols; ++i)
                for (int j = i + 1; j < t.cols; ++j) {
                    NzCombine cb{1, {cidx(i, i), cidx(j, j)}, {5, 5}};
                    TailD16 dr;
                    dr.ra = use16 ? d16 + i * d16_ts : nullptr; dr.rb = use16 ? d16 + j * d16_ts : nullptr;
                    PZ_TRY(term(i, j, rb, r_ct, t.tcols, cidx(i, j), &cb, use16 ? &dr : nullptr));
                }
            return PZ_OK;
        }
        for (int i = 0; i < t.cols; ++i) {
            NzCombine cb{3, {0, 0}, {0, 0}};
            int u = 0;
            for (int j = 0; j < t.cols; ++j) {
                if (j == i) continue;
                cb.col2[u] = cidx(i, j);
                cb.mode2[u] = 4;
                ++u;
            }
            PZ_TRY(term(i, i, rb, r_ct, t.tcols, cidx(i, i), &cb));
        }
        for (int i = 0; i < t.cols; ++i)
            for (int j = i + 1; j < t.cols; ++j) {
                NzCombine cb{3, {0, 0}, {0, 0}};
                PZ_TRY(term(i, j, rb, r_ct, t.tcols, cidx(i, j), &cb));
            }
        return PZ_OK;
    }
    // glwe_tensor_square_apply in the reference's order (:651-697): diagonal terms kept aside, each pairwise term minus its two diagonals
    int square_reference_order() {
        for (int i = 0; i < t.cols; ++i) {
            PZ_TRY(term(i, i, diag, dg_bs, t.cols, i));
            PZ_TRY(ew_res(EW_COPY, cidx(i, i), diag + (long long)i * n, dg_bs, (long long)t.cols * n, nullptr, 0, 0));
        }
        for (int i = 0; i < t.cols; ++i)
            for (int j = i + 1; j < t.cols; ++j) {
                const int c = cidx(i, j);
                PZ_TRY(term(i, j, rb, r_ct, t.tcols, c));
                PZ_TRY(ew_res(EW_SUB_I64, c, col_ptr(c), r_ct, rls, diag + (long long)i * n, dg_bs, (long long)t.cols * n));
                PZ_TRY(ew_res(EW_SUB_I64, c, col_ptr(c), r_ct, rls, diag + (long long)j * n, dg_bs, (long long)t.cols * n));
            }
        return PZ_OK;
    }
    // glwe_tensor_apply / _add_assign in the reference's order (:762-805 / :870-890): a temporary per term, element-wise passes over the tensor
    int apply_reference_order() {
        for (int i = 0; i < t.cols; ++i) {
            PZ_TRY(term(i, i, tmp, tmp_bs, 1, 0));
            if (add) PZ_TRY(ew_res(EW_ADD_I64, cidx(i, i), col_ptr(cidx(i, i)), r_ct, rls, tmp, tmp_bs, n));
            else PZ_TRY(ew_res(EW_COPY, cidx(i, i), tmp, tmp_bs, n, nullptr, 0, 0));
            for (int j = 0; j < t.cols; ++j) {
                if (j == i) continue;
                const int c = cidx(i, j);
                if (j < i || add) PZ_TRY(ew_res(EW_SUB_I64, c, col_ptr(c), r_ct, rls, tmp, tmp_bs, n));
                else PZ_TRY(ew_res(EW_NEG_I64, c, tmp, tmp_bs, n, nullptr, 0, 0));
            }
        }
        for (int i = 0; i < t.cols; ++i)
            for (int j = i + 1; j < t.cols; ++j) {
                PZ_TRY(term(i, j, tmp, tmp_bs, 1, 0));
                PZ_TRY(ew_res(EW_ADD_I64, cidx(i, j), col_ptr(cidx(i, j)), r_ct, rls, tmp, tmp_bs, n));
            }
        return PZ_OK;
    }
};

static int tensor_apply_nolock(pz_module* M, int64_t* res, const int64_t* a, const int64_t* b, const pz_glwe_tensor_params* p, int mode, size_t batch);
int pz_glwe_tensor_apply_batched(pz_module* M, int64_t* res, const int64_t* a, const int64_t* b, const pz_glwe_tensor_params* p, int mode,
                                 size_t batch) {
    PZ_ENTER(M);
    return tensor_apply_nolock(M, res, a, b, p, mode, batch);
}
static int tensor_apply_nolock(pz_module* M, int64_t* res, const int64_t* a, const int64_t* b, const pz_glwe_tensor_params* p, int mode, size_t batch) {
    TensorPlan t;
    PZ_TRY(tensor_plan(M, p, mode, t));
    const bool square = mode == PZ_TENSOR_SQUARE;
    if (square) b = a;
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a) && is_device_ptr(b), "batched entry points take device pointers");
    PZ_REQUIRE((const void*)res != (const void*)a && (const void*)res != (const void*)b, "glwe_tensor_apply: res must not alias an operand");
    if (batch == 0) return PZ_OK;
    TensorWave w;
    w.M = M; w.p = p; w.t = t; w.square = square; w.add = mode == PZ_TENSOR_APPLY_ADD_ASSIGN; w.n = (long long)M->n;
    const size_t chunk = tensor_chunk(M, t, batch);
    PZ_TRY(w.take_workspace(chunk));
    const long long a_ct = w.n * t.cols * t.a_size, b_ct = w.n * t.cols * t.b_size;
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        w.nb = (int)std::min(chunk, batch - b0);
        w.rb = res + (long long)b0 * w.r_ct;
        PZ_TRY(w.prepare(a + (long long)b0 * a_ct, b + (long long)b0 * b_ct));
        // where the digits of a term can go straight from the normalizing store into every tensor column that takes them (one base2k,
        // rank <= 2; POULPY_DBG_TENSOR_COMBINE=0: the reference's own order of element-wise passes)
        static const bool combine_env = (rt_knob("POULPY_DBG_TENSOR_COMBINE", 1) != 0);
        if (combine_env && p->res_base2k == p->ab_base2k && t.cols <= 3) PZ_TRY(w.combine_in_stores());
        else if (square) PZ_TRY(w.square_reference_order());
        else PZ_TRY(w.apply_reference_order());
    }
    return PZ_OK;
}

// glwe_tensor_apply (or _square_apply) followed by glwe_tensor_relinearize with the GLWETensor in scratch - poulpy-ckks's multiplication
// (poulpy-ckks/src/leveled/default/mul.rs:49-85 ckks_mul_into_default, :131-170 the square: `tmp` is taken from the scratch space, filled by
// glwe_tensor_apply and consumed by glwe_tensor_relinearize; operations/glwe.rs:609-913 and :541-607).  The tensor never reaches the caller, so
// where its digits fit 16 bits (one base2k <= 14 throughout, pipeline plans) it only ever exists as int16 copies in the tails' own tile
// order: the tensoring tails write 2 B per coefficient instead of 8, the relinearization's forward pass and tail read 2 B instead of 8
// (9.6 GB less HBM traffic per 256 multiplications at N = 2^16, 16 limbs).  Everywhere else: the i64 tensor in the workspace, the two calls as they are.
int pz_glwe_tensor_mul_relinearize_batched(pz_module* M, int64_t* res, const int64_t* a, const int64_t* b, const double* tsk_pmat,
                                           const pz_glwe_tensor_params* tp, const pz_glwe_op_params* rp, int mode, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(mode == PZ_TENSOR_APPLY || mode == PZ_TENSOR_SQUARE, "glwe_tensor_mul_relinearize: mode is apply or square");
    TensorPlan t;
    PZ_TRY(tensor_plan(M, tp, mode, t));
    PZ_REQUIRE(rp != nullptr, "null params");
    PZ_REQUIRE(rp->rank == tp->rank && rp->rank_out == rp->rank, "glwe_tensor_mul_relinearize: the tensor key maps rank (rank + 1) / 2 -> rank");
    PZ_REQUIRE(rp->a_size == tp->res_size && rp->a_base2k == tp->res_base2k, "glwe_tensor_mul_relinearize: (a_size, a_base2k) of the relinearization describe the tensor");
    const bool square = mode == PZ_TENSOR_SQUARE;
    if (square) b = a;
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a) && is_device_ptr(b) && is_device_ptr(tsk_pmat), "batched entry points take device pointers");
    if (batch == 0) return PZ_OK;
    TensorWave w;
    w.M = M; w.p = tp; w.t = t; w.square = square; w.add = false; w.n = (long long)M->n;
    static const bool t16_env = (exp_knob("POULPY_DBG_MUL_T16", 1) != 0);
    static const bool combine_env = (rt_knob("POULPY_DBG_TENSOR_COMBINE", 1) != 0), fused_env = (rt_knob("POULPY_DBG_TENSOR_FUSED", 1) != 0);
    const bool compact = t16_env && combine_env && fused_env && tp->res_base2k <= 14 && tp->res_base2k == tp->ab_base2k && t.cols <= 3 && t.dft_size >= 1 &&
                         mid_cnv_supported(M, t.a_size, t.b_size, std::min(t.dft_size, t.a_size + t.b_size - 1)) && glwe_relin_t16_supported(M, rp);
    const size_t chunk = std::min(tensor_chunk(M, t, batch), glwe_relin_chunk(M, rp, batch));
    const long long a_ct = w.n * t.cols * t.a_size, b_ct = w.n * t.cols * t.b_size, res_ct = w.n * t.cols * (long long)rp->res_size;
    const long long tensor_ct = w.n * t.tcols * t.res_size;
    // the tensor of a wave: ws2 (the two halves of the call carve the main workspace one after the other).  16-bit form: the columns of a wave
    // sit a fraction of the 4 MiB channel interleave out of phase - the pairwise tail reads two columns and writes the third at the same offset,
    // the relinearization's tail reads one beside its spectrum and result streams (as T2' against the result, kT2Phase in api_glwe.hip)
    static const long long t16_phase = (long long)exp_knob("POULPY_DBG_T16_PHASE_KIB", 768) * 1024 / 2;
    const long long t16_cs = (long long)chunk * t.res_size * w.n + t16_phase;
    PZ_TRY(ws2_reserve(M, compact ? (size_t)t16_cs * t.tcols * 2 : chunk * (size_t)tensor_ct * 8));
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        int64_t* r0 = res + (long long)b0 * res_ct;
        if (!compact) {
            // (not through the public entry points: the module lock is held)
            PZ_TRY(tensor_apply_nolock(M, (int64_t*)M->ws2, a + (long long)b0 * a_ct, b + (long long)b0 * b_ct, tp, mode, nb));
            PZ_TRY(glwe_op(M, true, r0, (const int64_t*)M->ws2, tsk_pmat, rp, nb, nullptr, nullptr, true));
            continue;
        }
        PZ_TRY(w.take_workspace(nb));
        w.nb = (int)nb; w.rb = nullptr; w.t16 = (short*)M->ws2; w.t16_cs = t16_cs;
        PZ_TRY(w.prepare(a + (long long)b0 * a_ct, b + (long long)b0 * b_ct));
        PZ_REQUIRE(w.fused, "glwe_tensor_mul_relinearize: the fused tensoring path was expected here");
        PZ_TRY(w.combine_in_stores());
        PZ_TRY(glwe_relin_t16(M, r0, (const short*)M->ws2, t16_cs, tsk_pmat, rp, nb));
    }
    return PZ_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Batched i64 VecZnx family: the limb-wise operations poulpy-core runs between the
// hot-path calls (glwe_add / sub / negate / copy / rotate / normalize / lsh / rsh), on `batch` device-resident containers laid
// out back to back — one launch per limb range instead of one call per ciphertext.  Same limb-range rules as the per-container
// entry points (reference/vec_znx/{add,sub,negate,copy,rotate,shift}.rs).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct BV {   // a batched device container
    DV v;
    int col;
};
inline DV dvb(const pz_module* M, const void* p, size_t cols, size_t size) { return DV{(void*)p, (long long)(M->n * cols * size), (int)cols, (int)size}; }
inline int ewb(pz_module* M, int op, int batch, const DV& r, int rcol, int rl0, const DV* a, int acol, int al0, const DV* b, int bcol, int bl0, int nl) {
    return launch_ew(M, op, poly_ptr(M, r, rcol, rl0), r.bs, limb_stride(M, r), a ? poly_ptr(M, *a, acol, al0) : nullptr, a ? a->bs : 0,
                     a ? limb_stride(M, *a) : 0, b ? poly_ptr(M, *b, bcol, bl0) : nullptr, b ? b->bs : 0, b ? limb_stride(M, *b) : 0, nl, batch);
}
int check_batched(const pz_module* M, const void* res, const void* a, const void* b, size_t res_col, size_t res_cols, size_t a_col,
                  size_t a_cols, size_t b_col, size_t b_cols) {
    (void)M;
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_*_batched(res)");
    if (a) PZ_CHECK_COL(a_col, a_cols, "vec_znx_*_batched(a)");
    if (b) PZ_CHECK_COL(b_col, b_cols, "vec_znx_*_batched(b)");
    PZ_REQUIRE(is_device_ptr(res) && (!a || is_device_ptr(a)) && (!b || is_device_ptr(b)), "batched entry points take device pointers");
    return PZ_OK;
}
// res = a +- b  (add.rs:6-65, sub.rs:6-58)
int add_sub_batched(pz_module* M, bool sub, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                    size_t a_cols, size_t a_size, size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_TRY(check_batched(M, res, a, b, res_col, res_cols, a_col, a_cols, b_col, b_cols));
    PZ_REQUIRE((const void*)res != (const void*)b, "vec_znx_add/sub_batched: res must not alias b (use the *_assign form)");
    const DV r = dvb(M, res, res_cols, res_size), av = dvb(M, a, a_cols, a_size), bv = dvb(M, b, b_cols, b_size);
    const bool a_le_b = a_size <= b_size;
    const int sum = (int)std::min(a_le_b ? a_size : b_size, res_size), cpy = (int)std::min(a_le_b ? b_size : a_size, res_size);
    const int B = (int)batch;
    PZ_TRY(ewb(M, sub ? EW_SUB_I64 : EW_ADD_I64, B, r, (int)res_col, 0, &av, (int)a_col, 0, &bv, (int)b_col, 0, sum));
    if (a_le_b) PZ_TRY(ewb(M, sub ? EW_NEG_I64 : EW_COPY, B, r, (int)res_col, sum, &bv, (int)b_col, sum, nullptr, 0, 0, cpy - sum));
    else PZ_TRY(ewb(M, EW_COPY, B, r, (int)res_col, sum, &av, (int)a_col, sum, nullptr, 0, 0, cpy - sum));
    return ewb(M, EW_ZERO, B, r, (int)res_col, cpy, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - cpy);
}
// mode 0: res += a, 1: res -= a, 2: res = a - res (and -res beyond a.size)   (add.rs:88-109, sub.rs:60-112)
int assign_batched(pz_module* M, int mode, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                   size_t a_cols, size_t a_size, size_t a_col) {
    PZ_TRY(check_batched(M, res, a, nullptr, res_col, res_cols, a_col, a_cols, 0, 0));
    const DV r = dvb(M, res, res_cols, res_size), av = dvb(M, a, a_cols, a_size);
    const int nl = (int)std::min(a_size, res_size), B = (int)batch;
    if (mode == 2) {
        PZ_TRY(ewb(M, EW_SUB_I64, B, r, (int)res_col, 0, &av, (int)a_col, 0, &r, (int)res_col, 0, nl));
        return ewb(M, EW_NEG_I64, B, r, (int)res_col, nl, &r, (int)res_col, nl, nullptr, 0, 0, (int)res_size - nl);
    }
    return ewb(M, mode == 0 ? EW_ADD_I64 : EW_SUB_I64, B, r, (int)res_col, 0, &r, (int)res_col, 0, &av, (int)a_col, 0, nl);
}
// op = EW_NEG_I64 / EW_COPY: res = op(a) over the common limbs, zero beyond   (negate.rs:6-29, copy.rs)
int unary_batched(pz_module* M, int op, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                  size_t a_cols, size_t a_size, size_t a_col) {
    PZ_TRY(check_batched(M, res, a, nullptr, res_col, res_cols, a_col, a_cols, 0, 0));
    const DV r = dvb(M, res, res_cols, res_size), av = dvb(M, a, a_cols, a_size);
    const int mn = (int)std::min(res_size, a_size), B = (int)batch;
    PZ_TRY(ewb(M, op, B, r, (int)res_col, 0, &av, (int)a_col, 0, nullptr, 0, 0, mn));
    return ewb(M, EW_ZERO, B, r, (int)res_col, mn, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - mn);
}
}  // namespace

extern "C" {

int pz_vec_znx_add_into_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                                size_t a_cols, size_t a_size, size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_batched(M, false, batch, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}
int pz_vec_znx_sub_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                           size_t a_cols, size_t a_size, size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_batched(M, true, batch, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}
int pz_vec_znx_add_assign_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                                  size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_batched(M, 0, batch, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_sub_assign_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                                  size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_batched(M, 1, batch, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_sub_negate_assign_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                         const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_batched(M, 2, batch, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_negate_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                              size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return unary_batched(M, EW_NEG_I64, batch, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_copy_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                            size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_copy_batched: res must not alias a");
    return unary_batched(M, EW_COPY, batch, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_zero_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    PZ_ENTER(M);
    PZ_TRY(check_batched(M, res, nullptr, nullptr, res_col, res_cols, 0, 0, 0, 0));
    const DV r = dvb(M, res, res_cols, res_size);
    return ewb(M, EW_ZERO, (int)batch, r, (int)res_col, 0, nullptr, 0, 0, nullptr, 0, 0, (int)res_size);
}
// res = X^k * a (rotate.rs:3-27); limbs of res beyond a.size zeroed
int pz_vec_znx_rotate_batched(pz_module* M, size_t batch, int64_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                              const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_TRY(check_batched(M, res, a, nullptr, res_col, res_cols, a_col, a_cols, 0, 0));
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_rotate_batched: res must not alias a");
    const DV r = dvb(M, res, res_cols, res_size), av = dvb(M, a, a_cols, a_size);
    const int mn = (int)std::min(res_size, a_size);
    const long long n = (long long)M->n;
    if (mn > 0 && batch > 0) {
        PolyMap sm{mn, 1, av.bs, (long long)a_cols * n, 0, n * (long long)a_col};
        PolyMap dm{mn, 1, r.bs, (long long)res_cols * n, 0, n * (long long)res_col};
        PZ_TRY(launch_rotate(M, (int)batch * mn, (const long long*)a, sm, (long long*)res, dm, 0, mn, nullptr, 0, 0, (long long)k));
    }
    return ewb(M, EW_ZERO, (int)batch, r, (int)res_col, mn, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - mn);
}
// vec_znx_normalize (same / cross base, any res_offset), vec_znx_lsh / rsh (= the same limb walk with res_offset = +-k at equal bases)
int pz_vec_znx_normalize_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                                 size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    return pz_vec_znx_big_normalize_batched(M, batch, res, res_cols, res_size, res_base2k, res_offset, res_col, a, a_cols, a_size, a_base2k, a_col);
}
int pz_vec_znx_lsh_batched(pz_module* M, size_t batch, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    if (k > ((size_t)1 << 40)) return fail(PZ_ERR_INVALID, "vec_znx_lsh_batched: shift out of range");
    return pz_vec_znx_big_normalize_batched(M, batch, res, res_cols, res_size, base2k, (int64_t)k, res_col, a, a_cols, a_size, base2k, a_col);
}
int pz_vec_znx_rsh_batched(pz_module* M, size_t batch, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    if (k > ((size_t)1 << 40)) return fail(PZ_ERR_INVALID, "vec_znx_rsh_batched: shift out of range");
    return pz_vec_znx_big_normalize_batched(M, batch, res, res_cols, res_size, base2k, -(int64_t)k, res_col, a, a_cols, a_size, base2k, a_col);
}

}  // extern "C"
