// api_br.hip — C ABI of the composite calls built on the batched GLWE product: CGGI blind rotation (block-binary, standard, extended),
// circuit bootstrapping (constant / exponent mode) and GLWE packing.  poulpy-bin-fhe blind_rotation/algorithms/cggi/algorithm.rs,
// circuit_bootstrapping/circuit.rs; poulpy-core glwe_packing.rs.  SURVEY.md §8f rank 2.
#include "api_common.hpp"
#include "api_glwe.hpp"

using namespace pz;

// the tail of a rotation step: acc = normalize(idft(acc_add) + acc), in place on the accumulator (algorithm.rs:342-346) - every column
// of `acc` is both the operand and the destination
static TailCall acc_tail(int batch, const cplx* T, bool rowmajor, int nlimbs, int cols, int64_t* acc, long long acc_bs, int acc_size, int base2k) {
    TailCall c;
    c.batch = batch; c.T = T; c.rowmajor = rowmajor; c.nlimbs = nlimbs; c.ncols = cols;
    c.res = (long long*)acc; c.res_bs = acc_bs; c.res_cols = cols; c.res_size = acc_size; c.base2k = base2k;
    c.small = (const long long*)acc; c.small_bs = acc_bs; c.small_cols = cols; c.small_size = acc_size; c.small_all = true;
    return c;
}

extern "C" {

// ------------------------------------------------------------------------------
// public: CGGI blind rotation on a batch of LWE ciphertexts (device-resident)
// poulpy-bin-fhe/src/blind_rotation/algorithms/cggi/algorithm.rs:76-118 (dispatch), :265-368 (block binary), :370-440 (standard)
// ------------------------------------------------------------------------------
size_t pz_blind_rotation_workspace_bytes(const pz_module* M, const pz_blind_rotation_params* p, size_t batch) {
    if (!M || !p) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->rank + 1;
    const size_t tp = cols * std::max({(size_t)p->dnum, (size_t)p->brk_size, (size_t)p->res_size});
    const size_t T = align256(batch * tp * (size_t)M->m * sizeof(cplx));
    if (p->block_size > 1) {
        // the composed path, or (plans with 128-point rows) the row-sliced keys of one block + T' + T2' of the three-kernel block step
        const size_t composed = align256(batch * n8 * cols * p->dnum) + 2 * align256(batch * n8 * cols * p->brk_size) + T +
                                align256(batch * (size_t)M->n * cols * p->res_size * sizeof(int));   // (+ the 32-bit digits of the accumulator between blocks)
        const size_t mid = align256((size_t)p->block_size * p->dnum * cols * cols * p->brk_size * n8) +
                           align256(batch * n8 * cols * std::min((size_t)p->dnum, (size_t)p->res_size)) + align256(batch * n8 * cols * p->brk_size) +
                           kMidDummyBytes + align256(batch * (size_t)M->n * cols * p->res_size * sizeof(int));
        return std::max(composed, mid);
    }
    pz_glwe_op_params ep;
    ep.rank = p->rank; ep.dnum = p->dnum; ep.dsize = 1; ep.key_size = p->brk_size; ep.key_base2k = p->base2k;
    ep.a_size = p->res_size; ep.a_base2k = p->base2k; ep.res_size = p->res_size; ep.res_base2k = p->base2k; ep.rank_out = p->rank;
    return align256(batch * n8 * cols * p->res_size) + pz_glwe_op_workspace_bytes(M, &ep, batch, 0);
}

// One rotation call: the shape, read once; the paths below take it by reference
struct BrCall {
    pz_module* M;
    int64_t* res;
    const int64_t* lwe_2n;
    const int64_t* lut;
    const double* brk;
    const pz_blind_rotation_params* p;
    size_t batch;
    long long n, lwe_bs, res_ct;
    int cols, dnum, bsz, rsz, B, n_lwe, blk, k;
    size_t pmat_doubles, n8;
};
#define PZ_BR_UNPACK(c)                                                                                                                  \
    pz_module* const M = (c).M; int64_t* const res = (c).res; const int64_t* const lwe_2n = (c).lwe_2n; const double* const brk = (c).brk;   \
    const pz_blind_rotation_params* const p = (c).p; const size_t batch = (c).batch; const long long n = (c).n, lwe_bs = (c).lwe_bs, res_ct = (c).res_ct; \
    const int cols = (c).cols, dnum = (c).dnum, bsz = (c).bsz, rsz = (c).rsz, B = (c).B, n_lwe = (c).n_lwe, blk = (c).blk, k = (c).k;      \
    const size_t pmat_doubles = (c).pmat_doubles, n8 = (c).n8;                                                                              \
    (void)p; (void)batch; (void)n; (void)lwe_bs; (void)res_ct; (void)cols; (void)dnum; (void)bsz; (void)rsz; (void)B; (void)n_lwe; (void)blk; (void)k; \
    (void)pmat_doubles; (void)n8; (void)res; (void)lwe_2n; (void)brk; (void)M;

// acc = X^b * LUT in column 0, zero elsewhere (:298-301 / :413-416)
static int br_init_accumulator(const BrCall& c) {
    PZ_BR_UNPACK(c)
    PZ_TRY(launch_zero_bytes(M, res, (size_t)B * res_ct * 8));
    const int nl = std::min(rsz, (int)p->lut_size);
    PolyMap sm{nl, 1, 0, n, 0, 0};                     // the LUT is shared: batch stride 0, VecZnx(1, lut_size)
    PolyMap dm{nl, 1, res_ct, (long long)cols * n, 0, 0};
    return launch_rotate(M, B * nl, (const long long*)c.lut, sm, (long long*)res, dm, 0, nl, (const long long*)lwe_2n, lwe_bs, 0, 0);
}

// plans with 128-point rows (N >= 4096): the block step on the three-kernel pipeline of the GLWE products - pass 1 of the accumulator limbs |
// k_mid128<.., BR> (row DFT, the block's blk products weighted by DFT(X^a_i - 1), inverse row DFT) | tail (inverse column pass + accumulator +
// carry chain): the spectra never reach HBM and one launch covers the whole block.  *taken = false: the shape is not covered
static int br_pipeline_path(const BrCall& c, bool* taken) {
    PZ_BR_UNPACK(c)
    *taken = false;
    const int npi = cols * std::min(dnum, rsz), npo = cols * bsz, nrows_key = dnum * cols, ncols_key = cols * bsz;
    static const int br_mid = exp_knob("POULPY_DBG_BR_MID", 1);
    if (!(br_mid && M->fuse_mid && M->fuse_tail && tail_supported(M) && M->plan.m2 == 128 && mid_supported(M, npi, npo) && npi == nrows_key && blk <= 16))
        return PZ_OK;
    *taken = true;
    const size_t key_bytes = align256((size_t)blk * nrows_key * ncols_key * n8);
    const size_t t_bytes = align256(batch * npi * (size_t)M->m * sizeof(cplx)), t2_bytes = align256(batch * npo * (size_t)M->m * sizeof(cplx));
    // between two blocks the accumulator holds normalized digits: 32-bit values in the workspace (base2k <= 31) - pass 1 and the tail
    // move them at half the bytes; the caller's `res` is the operand of the first block and the destination of the last
    const int nblocks = n_lwe / blk;
    const bool acc32 = k <= 31 && nblocks >= 2 && tail_acc32_supported(M);
    // round 6: 16-bit values in the tails' own tile order where the digits fit them (base2k <= 15): a quarter of the i64 bytes, whole 128-byte runs for
    // pass 1 (k_fwd_pass1_t16) and the tail (TailCall::acc32 bits 3 / 4); POULPY_DBG_BR_ACC16=0: the 32-bit form
    static const int acc16_knob = exp_knob("POULPY_DBG_BR_ACC16", 1);
    const bool acc16 = acc32 && acc16_knob && k <= 15;
    const size_t d_bytes = acc32 ? align256((size_t)B * res_ct * sizeof(int)) : 0;
    PZ_TRY(ws_reserve(M, key_bytes + t_bytes + t2_bytes + kMidDummyBytes + d_bytes));
    char* base = (char*)M->ws;
    cplx* Pp; cplx* T; cplx* T2; cplx* mid_dummy; int* D = nullptr;
    PZ_TRY(ws_take(M, base, key_bytes, &Pp));
    PZ_TRY(ws_take(M, base, t_bytes, &T));
    PZ_TRY(ws_take(M, base, t2_bytes, &T2));
    PZ_TRY(ws_take(M, base, kMidDummyBytes, &mid_dummy));
    if (acc32) PZ_TRY(ws_take(M, base, d_bytes, &D));
    PolyMap sm{npi / cols, cols, res_ct, (long long)cols * n, n, 0};
    // (round 5, measured and dropped: the two-stream split of the small-ring path below applied here - the persistent middle kernel holds every
    //  CU's LDS, so the other half's pass 1 / tail cannot run beside it: N = 4096 -3.6 %, N = 2^14 +-0; profiles/r05_ab_br_two_streams_pipe.txt)
    for (int b0 = 0; b0 + blk <= n_lwe; b0 += blk) {
        const bool in32 = acc32 && b0 > 0, out32 = acc32 && b0 + 2 * blk <= n_lwe;
        PZ_TRY(launch_permute_pmat(M, brk + (size_t)b0 * pmat_doubles, Pp, blk * nrows_key * ncols_key));
        if (in32 && acc16) PZ_TRY(launch_fwd_pass1_t16(M, B * npi, (const short*)D, sm, T));
        else PZ_TRY(launch_fwd_pass1(M, B * npi, in32 ? (const long long*)D : (const long long*)res, sm, T, true, -1, in32));
        MidBr mb{(const long long*)lwe_2n, lwe_bs, b0, blk};
        PZ_TRY(launch_mid(M, B, T, T2, Pp, npi, npo, nrows_key, ncols_key, mid_dummy, 0, 0, nullptr, &mb));
        TailCall tc = acc_tail(B, T2, true, bsz, cols, res, res_ct, rsz, k);
        if (in32) tc.small = (const long long*)D;
        if (out32) tc.res = (long long*)D;
        tc.acc32 = acc16 ? ((in32 ? 8 : 0) | (out32 ? 16 : 0)) : ((in32 ? 1 : 0) | (out32 ? 2 : 0));
        PZ_TRY(launch_inv_tail(M, tc));
    }
    return PZ_OK;
}

// N = 1024 / 2048 off the one-kernel path (accumulators beyond LDS: N = 2048, rank 2 at N = 1024): the transforms around the block step are the
// two kernels of the small-ring pipeline (device_small.hpp) - the whole forward transform of the accumulator limbs in LDS, written in the
// standard spectrum order | the block step on the standard keys | whole inverse transform + accumulator + carry chain per (ciphertext, column) -
// instead of pass 1 / pass 2 and pass 2 / tail.  *taken = false: the shape is not covered
static int br_small_ring_path(const BrCall& c, bool* taken) {
    PZ_BR_UNPACK(c)
    *taken = false;
    const int npi = cols * std::min(dnum, rsz), nrows_key = dnum * cols, ncols_key = cols * bsz;
    static const int br_small = exp_knob("POULPY_DBG_BR_SMALL", 1);
    if (!(br_small && M->small_path && M->fuse_mid && M->fuse_tail && small_supported(M, npi, bsz) && npi == nrows_key && npi <= 12 && blk <= 64))
        return PZ_OK;
    *taken = true;
    const size_t s_bytes = align256(batch * npi * (size_t)M->m * sizeof(cplx)), a_bytes = align256(batch * ncols_key * (size_t)M->m * sizeof(cplx));
    // between two blocks the accumulator holds normalized digits: kept as 32-bit values in the workspace (base2k <= 31), read and
    // written by the inverse kernel at half the bytes; the caller's `res` receives the i64 limbs from the last block
    const int nblocks = n_lwe / blk;
    const bool acc32 = k <= 31 && nblocks >= 2;
    const size_t d_bytes = acc32 ? align256((size_t)B * res_ct * sizeof(int)) : 0;
    PZ_TRY(ws_reserve(M, s_bytes + a_bytes + d_bytes));
    char* base = (char*)M->ws;
    cplx* S; cplx* A; int* D = nullptr;
    PZ_TRY(ws_take(M, base, s_bytes, &S));
    PZ_TRY(ws_take(M, base, a_bytes, &A));
    if (acc32) PZ_TRY(ws_take(M, base, d_bytes, &D));
    PolyMap sm{npi / cols, cols, res_ct, (long long)cols * n, n, 0};
    // the inverse kernel of a block also runs the forward transform of the new accumulator for the next block (POULPY_DBG_BR_SMALL=2:
    // separate k_small_fwd launches), when its limbs are among the ones the inverse produces
    const int fl = npi / cols;
    const bool chain = br_small != 2 && fl <= bsz && fl <= rsz && M->n < 4096;   // (N = 4096 - block sizes the pipeline path declines - has no chained form)
    // Two halves of the batch on two streams (round 5): the block step is bound by FP64 issue, the inverse / forward kernel around it by
    // HBM and LDS latency - issued back to back on one stream each leaves the other's unit idle; as two independent chains the step of one
    // half overlaps with the transforms of the other (split at a tile boundary of the block step: 8 ciphertexts)
    // (measured, profiles/r05_ab_br_two_streams*.txt: N = 2048 at 1024 / 512 / 256 per call +5.5 % / +10 % / -16 %; rank 2 at N = 1024:
    //  1024 per call +1 %, 512 -4 % - a half must still fill the chip: >= 2^19 coefficients per column)
    const int hA = ((long long)(B / 2) * n >= (1ll << 19) && !M->timing) ? ((B / 2 + 7) / 8) * 8 : B;
    SideStream ss(M);
    if (hA < B) PZ_TRY(ss.fork());
    for (int b0 = 0; b0 + blk <= n_lwe; b0 += blk) {
        const bool more = b0 + 2 * blk <= n_lwe;
        // operand: `res` in the first block, the 32-bit digits afterwards; destination: the 32-bit digits while blocks follow
        // (the separate forward launch of the unchained form reads `res`: i64 throughout there)
        const bool use32 = acc32 && chain;
        const bool in32 = use32 && b0 > 0, out32 = use32 && more;
        // (round 6: 16-bit digits - natural order here, the small-ring kernels read whole rows - where base2k <= 15; POULPY_DBG_BR_ACC16=0: 32-bit)
        static const int acc16_knob_s = exp_knob("POULPY_DBG_BR_ACC16", 1);
        const bool use16 = use32 && acc16_knob_s && k <= 15;
        for (int half = 0; half < (hA < B ? 2 : 1); ++half) {
            const int c0 = half ? hA : 0, nb = half ? B - hA : hA;
            ss.on(half == 1);
            int64_t* res_h = res + (long long)c0 * res_ct;
            int* D_h = D ? D + (long long)c0 * res_ct : nullptr;
            cplx* S_h = S + (size_t)c0 * npi * M->m;
            cplx* A_h = A + (size_t)c0 * ncols_key * M->m;
            if (b0 == 0 || !chain) PZ_TRY(launch_small_fwd(M, nb * npi, (const long long*)res_h, sm, S_h, true));
            bool done = false;
            PZ_TRY(br_block_step(M, (const double*)S_h, (long long)npi * n, (double*)A_h, (long long)ncols_key * n, brk, pmat_doubles, npi, ncols_key,
                                 nb, b0, blk, lwe_2n + (long long)c0 * lwe_bs, lwe_bs, &done));
            if (!done) return fail(PZ_ERR_UNSUPPORTED, "blind_rotation: block step not launched");
            PZ_TRY(launch_small_inv(M, nb, A_h, nullptr, ncols_key, 0, 0, cols, bsz, out32 ? (long long*)D_h : (long long*)res_h, res_ct, cols, rsz,
                                    in32 ? (const long long*)D_h : (const long long*)res_h, res_ct, cols, rsz, k, -1, true,
                                    (chain && more) ? S_h : nullptr, fl, false, 0, 0, false,
                                    use16 ? ((in32 ? 4 : 0) | (out32 ? 8 : 0)) : ((in32 ? 1 : 0) | (out32 ? 2 : 0))));
        }
    }
    return ss.join();
}

// every other block-binary shape: per-op transforms around the fused block step (or, without it, the reference's own op sequence)
static int br_composed_path(const BrCall& c) {
    PZ_BR_UNPACK(c)
    DV rv{res, res_ct, cols, rsz};
    const size_t acc_dft_bytes = align256(batch * n8 * cols * dnum), vr_bytes = align256(batch * n8 * cols * bsz);
    const size_t tp = (size_t)cols * std::max({dnum, bsz, rsz});
    const size_t t_bytes = align256(batch * tp * (size_t)M->m * sizeof(cplx));
    PZ_TRY(ws_reserve(M, acc_dft_bytes + 2 * vr_bytes + t_bytes));
    char* base = (char*)M->ws;
    double* acc_dft; double* vmp_res; double* acc_add; cplx* T;
    PZ_TRY(ws_take(M, base, acc_dft_bytes, &acc_dft));
    PZ_TRY(ws_take(M, base, vr_bytes, &vmp_res));
    PZ_TRY(ws_take(M, base, vr_bytes, &acc_add));
    PZ_TRY(ws_take(M, base, t_bytes, &T));
    DV ad{acc_dft, n * cols * dnum, cols, dnum}, vr{vmp_res, n * cols * bsz, cols, bsz}, aa{acc_add, n * cols * bsz, cols, bsz};
    const bool tail = M->fuse_tail && tail_supported(M);
    for (int b0 = 0; b0 + blk <= n_lwe; b0 += blk) {  // chunks_exact: a trailing partial block is ignored, as in the reference
        PZ_TRY(dev_dft_apply(M, B, 1, 0, ad, 0, rv, 0, cols, nullptr, T));                      // :319-321
        const int row_max = std::min(dnum * cols, cols * std::min(dnum, rsz));
        bool block_done = false;
        if (M->fuse_mid) PZ_TRY(br_block_step(M, acc_dft, ad.bs, acc_add, aa.bs, brk, pmat_doubles, row_max, cols * bsz, B, b0, blk, lwe_2n, lwe_bs, &block_done));
        if (!block_done) {
        PZ_TRY(launch_zero_bytes(M, acc_add, (size_t)B * aa.bs * 8));                     // :321
        for (int i = b0; i < b0 + blk; ++i) {                                                       // :324-337
            PZ_TRY(dev_vmp(M, B, vr, ad, brk + (size_t)i * pmat_doubles, dnum, cols, cols, bsz, 0));
            PZ_TRY(launch_xai_acc(M, acc_add, aa.bs, vmp_res, vr.bs, cols * bsz, B, lwe_2n, lwe_bs, i));
        }
        }
        // acc = normalize(idft(acc_add) + acc)  (:342-346)
        if (tail) {
            PolyMap sm{bsz, cols, aa.bs, (long long)cols * n, n, 0};
            PZ_TRY(launch_inv_pass2(M, B * bsz * cols, acc_add, sm, T));
            PZ_TRY(launch_inv_tail(M, acc_tail(B, T, false, bsz, cols, res, res_ct, rsz, k)));
        } else {
            PZ_TRY(dev_idft(M, B, aa, 0, aa, 0, cols, bsz, T));
            for (int c = 0; c < cols; ++c) {
                PZ_TRY(launch_ew(M, EW_ADD_I64, (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n,
                                 (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n, res + (long long)c * n, res_ct,
                                 (long long)cols * n, std::min(bsz, rsz), B));
                PZ_TRY(dev_normalize(M, B, rv, k, 0, c, aa, k, c));
            }
        }
    }
    return PZ_OK;
}

// execute_standard (block size 1): acc += (X^a_i - 1) * (acc (x) BRK_i) per coefficient, one normalization at the end (:423-437)
static int br_standard_path(const BrCall& c) {
    PZ_BR_UNPACK(c)
    DV rv{res, res_ct, cols, rsz};
    pz_glwe_op_params ep;
    ep.rank = p->rank; ep.dnum = p->dnum; ep.dsize = 1; ep.key_size = p->brk_size; ep.key_base2k = p->base2k;
    ep.a_size = p->res_size; ep.a_base2k = p->base2k; ep.res_size = p->res_size; ep.res_base2k = p->base2k; ep.rank_out = p->rank;
    // acc_tmp lives in the module's second workspace: the external product owns the first one
    PZ_TRY(ws2_reserve(M, (size_t)B * res_ct * 8));
    int64_t* acc_tmp = (int64_t*)M->ws2;
    PolyMap pm{rsz, cols, res_ct, (long long)cols * n, n, 0};
    for (int i = 0; i < n_lwe; ++i) {
        PZ_TRY(glwe_op(M, false, acc_tmp, res, brk + (size_t)i * pmat_doubles, &ep, batch));
        PZ_TRY(launch_rotate(M, B * rsz * cols, (const long long*)acc_tmp, pm, (long long*)res, pm, 2, rsz * cols, (const long long*)lwe_2n,
                             lwe_bs, 1 + i, 0));
    }
    // vec_znx_normalize_assign (normalize.rs:403-425) == out-of-place same-base normalize of a copy
    PZ_TRY(launch_ew(M, EW_COPY, acc_tmp, res_ct, n, res, res_ct, n, nullptr, 0, 0, cols * rsz, B));   // (a kernel node, not a memcpy node: see launch_zero_bytes)
    DV tv{acc_tmp, res_ct, cols, rsz};
    for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, rv, k, 0, c, tv, k, c));
    return PZ_OK;
}

static int blind_rotation(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                          const pz_blind_rotation_params* p, size_t batch) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->n_lwe >= 1 && p->block_size >= 1 && p->dnum >= 1 && p->brk_size >= 1 && p->res_size >= 1 && p->lut_size >= 1,
               "blind_rotation: empty shape");
    PZ_REQUIRE(p->base2k >= 1 && p->base2k <= 63, "blind_rotation: base2k out of range");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(lwe_2n) && is_device_ptr(lut) && is_device_ptr(brk),
               "batched entry points take device pointers");
    if (batch == 0) return PZ_OK;
    BrCall c;
    c.M = M; c.res = res; c.lwe_2n = lwe_2n; c.lut = lut; c.brk = brk; c.p = p; c.batch = batch;
    c.n = (long long)M->n; c.cols = (int)p->rank + 1; c.dnum = (int)p->dnum; c.bsz = (int)p->brk_size; c.rsz = (int)p->res_size;
    c.B = (int)batch; c.n_lwe = (int)p->n_lwe; c.blk = (int)p->block_size; c.k = (int)p->base2k;
    c.lwe_bs = (long long)c.n_lwe + 1;
    c.pmat_doubles = (size_t)c.n * c.dnum * c.cols * c.cols * c.bsz;
    c.res_ct = c.n * c.cols * c.rsz;
    c.n8 = (size_t)M->n * 8;
    PZ_TRY(br_init_accumulator(c));
    PZ_TRY(ensure_w2n(M));
    bool taken = false;
    PZ_TRY(br_try_fused(M, res, lwe_2n, lut, brk, p, batch, &taken));   // the whole rotation in one kernel (device_br.hpp, br_forms.hpp)
    if (taken) return PZ_OK;
    if (c.blk == 1) return br_standard_path(c);
    PZ_TRY(br_pipeline_path(c, &taken));
    if (taken) return PZ_OK;
    PZ_TRY(br_small_ring_path(c, &taken));
    if (taken) return PZ_OK;
    return br_composed_path(c);
}

// execute_block_binary_extended (algorithm.rs:121-273; extension_factor > 1, block_size > 1): the ext accumulators of a
// ciphertext are one more batch dimension ([b][e]); per LWE block: batched forward DFT | per coefficient: batched VMP, then
// k_xai_ext moves the products between the accumulators as the reference does | inverse DFT + acc + carry chain (fused tail).
size_t pz_blind_rotation_extended_tmp_bytes(const pz_module* M, const pz_blind_rotation_params* p, size_t extension_factor, size_t batch) {
    if (!M || !p) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->rank + 1, be = batch * extension_factor;
    return align256(be * n8 * cols * p->res_size) + align256(be * n8 * cols * p->dnum) + 2 * align256(be * n8 * cols * p->brk_size) +
           align256(be * cols * std::max({(size_t)p->dnum, (size_t)p->brk_size, (size_t)p->res_size}) * (size_t)M->m * sizeof(cplx));
}
static int blind_rotation_extended(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                   const pz_blind_rotation_params* p, size_t extension_factor, void* tmp, size_t tmp_bytes, size_t batch) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->n_lwe >= 1 && p->block_size >= 1 && p->dnum >= 1 && p->brk_size >= 1 && p->res_size >= 1 && p->lut_size >= 1,
               "blind_rotation: empty shape");
    PZ_REQUIRE(p->base2k >= 1 && p->base2k <= 63, "blind_rotation: base2k out of range");
    PZ_REQUIRE(extension_factor >= 1 && (extension_factor & (extension_factor - 1)) == 0 && extension_factor <= 64,
               "blind_rotation: extension_factor must be a power of two");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(lwe_2n) && is_device_ptr(lut) && is_device_ptr(brk) && is_device_ptr(tmp),
               "batched entry points take device pointers");
    PZ_REQUIRE(tmp_bytes >= pz_blind_rotation_extended_tmp_bytes(M, p, extension_factor, batch), "blind_rotation: tmp is too small");
    if (batch == 0) return PZ_OK;
    PZ_TRY(ensure_w2n(M));
    int log_ext = 0;
    while (((size_t)1 << log_ext) < extension_factor) ++log_ext;
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1, dnum = (int)p->dnum, bsz = (int)p->brk_size, rsz = (int)p->res_size, k = (int)p->base2k;
    const int B = (int)batch, BE = B * (int)extension_factor, n_lwe = (int)p->n_lwe, blk = (int)p->block_size;
    const long long lwe_bs = (long long)n_lwe + 1, res_ct = n * cols * rsz;
    const size_t pmat_doubles = (size_t)M->n * dnum * cols * cols * bsz;
    const size_t n8 = (size_t)M->n * 8;
    char* base = (char*)tmp;
    int64_t* acc = (int64_t*)base; base += align256((size_t)BE * n8 * cols * rsz);
    double* acc_dft = (double*)base; base += align256((size_t)BE * n8 * cols * dnum);
    double* vmp_res = (double*)base; base += align256((size_t)BE * n8 * cols * bsz);
    double* acc_add = (double*)base; base += align256((size_t)BE * n8 * cols * bsz);
    cplx* T = (cplx*)base;
    // :159-161 zero, :180-190 rotated table
    PZ_TRY(launch_zero_bytes(M, acc, (size_t)BE * res_ct * 8));
    PZ_REQUIRE(BE <= 65535, "blind_rotation: batch * extension_factor exceeds 65535 (split the batch)");
    PZ_TRY(launch_br_ext_init(M, acc, lut, lwe_2n, lwe_bs, log_ext, cols, rsz, (int)p->lut_size, B));
    DV rv{acc, res_ct, cols, rsz};
    DV ad{acc_dft, n * cols * dnum, cols, dnum}, vr{vmp_res, n * cols * bsz, cols, bsz}, aa{acc_add, n * cols * bsz, cols, bsz};
    const bool tail = M->fuse_tail && tail_supported(M);
    // N = 1024 / 2048 / 4096: the transforms of the small-ring pipeline around the per-coefficient steps, as in blind_rotation()
    static const int br_small = exp_knob("POULPY_DBG_BR_SMALL", 1);
    const int npi = cols * std::min(dnum, rsz);
    const bool small_tf = br_small && M->small_path && M->fuse_mid && M->fuse_tail && small_supported(M, npi, bsz) && npi == dnum * cols;
    for (int b0 = 0; b0 + blk <= n_lwe; b0 += blk) {
        if (small_tf) {
            PolyMap sm{npi / cols, cols, res_ct, (long long)cols * n, n, 0};
            PZ_TRY(launch_small_fwd(M, BE * npi, (const long long*)acc, sm, (cplx*)acc_dft, true));
        } else
        PZ_TRY(dev_dft_apply(M, BE, 1, 0, ad, 0, rv, 0, cols, nullptr, T));                           // :195-200
        PZ_TRY(launch_zero_bytes(M, acc_add, (size_t)BE * aa.bs * 8));
        for (int i = b0; i < b0 + blk; ++i) {
            PZ_TRY(dev_vmp(M, BE, vr, ad, brk + (size_t)i * pmat_doubles, dnum, cols, cols, bsz, 0));   // :209-211
            PZ_TRY(launch_xai_ext(M, acc_add, vmp_res, cols * bsz, log_ext, B, lwe_2n, lwe_bs, i));
        }
        if (small_tf) {
            PZ_TRY(launch_small_inv(M, BE, (const cplx*)acc_add, nullptr, cols * bsz, 0, 0, cols, bsz, (long long*)acc, res_ct, cols, rsz,
                                    (const long long*)acc, res_ct, cols, rsz, k, -1, true));
        } else if (tail) {                                                                             // :260-266
            PolyMap sm{bsz, cols, aa.bs, (long long)cols * n, n, 0};
            PZ_TRY(launch_inv_pass2(M, BE * bsz * cols, acc_add, sm, T));
            PZ_TRY(launch_inv_tail(M, acc_tail(BE, T, false, bsz, cols, acc, res_ct, rsz, k)));
        } else {
            PZ_TRY(dev_idft(M, BE, aa, 0, aa, 0, cols, bsz, T));
            for (int c = 0; c < cols; ++c) {
                PZ_TRY(launch_ew(M, EW_ADD_I64, (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n,
                                 (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n, acc + (long long)c * n, res_ct,
                                 (long long)cols * n, std::min(bsz, rsz), BE));
                PZ_TRY(dev_normalize(M, BE, rv, k, 0, c, aa, k, c));
            }
        }
    }
    // :270-272 res = acc[0]
    return launch_ew(M, EW_COPY, res, res_ct, n, acc, (long long)extension_factor * res_ct, n, nullptr, 0, 0, cols * rsz, B);
}
int pz_blind_rotation_execute_extended_batched(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                               const pz_blind_rotation_params* p, size_t extension_factor, void* tmp, size_t tmp_bytes,
                                               size_t batch) {
    PZ_ENTER(M);
    return blind_rotation_extended(M, res, lwe_2n, lut, brk, p, extension_factor, tmp, tmp_bytes, batch);
}

int pz_blind_rotation_execute_batched(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                      const pz_blind_rotation_params* p, size_t batch) {
    PZ_ENTER(M);
    KeyHash k;
    k.add((int)2); k.add(res); k.add(lwe_2n); k.add(lut); k.add(brk); k.add(batch);
    if (p) k.add(*p);
    graph_key_module(M, k);
    return with_graph(M, k.h, [&]() { return blind_rotation(M, res, lwe_2n, lut, brk, p, batch); });
}

// ------------------------------------------------------------------------------
// public: circuit bootstrapping LWE -> GGSW, constant mode, one base2k for every key and the result
// poulpy-bin-fhe/src/circuit_bootstrapping/circuit.rs:219-370 (circuit_bootstrap_core, to_exponent = false):
//   :321-331  acc = blind_rotation(lwe, lut)                                       (copy into the atk layout: same limbs)
//   :344-366  entry (i, 0) of the GGSW = glwe_trace(X^(-i*gap) * acc, skip 0)       (the reference rotates acc in place between rows)
//   :369      ggsw_expand_row
// The dnum_res traces of one LWE are independent, so all batch * dnum_res of them run as one batched trace.
// ------------------------------------------------------------------------------
static int glwe_pack(pz_module* M, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out,
                     const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p, void* tmp, size_t tmp_bytes,
                     size_t batch, size_t trace_size);
struct CbtRepack { size_t log_gap_in, log_gap_out, log_domain; };  // exponent mode with log_gap_in != log_gap_out (post_process)
static inline size_t cbt_atk_size(const pz_circuit_bootstrapping_params* p) { return (size_t)(p->atk_glwe_size ? p->atk_glwe_size : p->br.res_size); }
static inline size_t cbt_tmp_size(const pz_circuit_bootstrapping_params* p) {
    return (size_t)(p->trace_size ? p->trace_size : std::max((uint64_t)cbt_atk_size(p), p->res_size));
}
static inline uint64_t cbt_base(const pz_circuit_bootstrapping_params* p, uint64_t b) { return b ? b : p->br.base2k; }
static inline size_t cbt_ext(const pz_circuit_bootstrapping_params* p) { return p->extension_factor > 1 ? (size_t)p->extension_factor : 1; }
size_t pz_circuit_bootstrapping_tmp_bytes(const pz_module* M, const pz_circuit_bootstrapping_params* p, size_t batch) {
    if (!M || !p) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->br.rank + 1;
    const size_t ext_bytes = cbt_ext(p) > 1 ? align256(pz_blind_rotation_extended_tmp_bytes(M, &p->br, cbt_ext(p), batch)) : 0;
    // acc | (bases differ) acc in the automorphism keys' base | the dnum_res rotated copies / traces | extended rotation scratch
    const size_t conv = cbt_base(p, p->atk_base2k) != p->br.base2k ? align256(batch * n8 * cols * cbt_atk_size(p)) : 0;
    return align256(batch * n8 * cols * p->br.res_size) + align256(batch * p->res_dnum * n8 * cols * cbt_tmp_size(p)) + conv + ext_bytes;
}
size_t pz_circuit_bootstrapping_to_exponent_tmp_bytes(const pz_module* M, const pz_circuit_bootstrapping_params* p, size_t log_domain,
                                                      size_t batch) {
    if (!M || !p || log_domain > 20) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->br.rank + 1;
    const size_t rows_ct = align256(batch * p->res_dnum * n8 * cols * cbt_tmp_size(p));
    // acc | rotated rows | 2^log_domain shifted copies | packed result | glwe_pack scratch (3 ciphertext arrays)
    return pz_circuit_bootstrapping_tmp_bytes(M, p, batch) + (((size_t)1 << log_domain) + 1 + 3) * rows_ct;
}
// circuit bootstrapping, exponent mode - post_process (circuit.rs:373-421) with log_gap_in != log_gap_out: partial trace of the `count` rows in
// `tr`, 2^log_domain shifted copies, glwe_pack.  *row_src = the packed rows (behind `tr` in the caller's scratch)
static int cbt_repack_rows(pz_module* M, int64_t* tr, const int64_t** row_src, size_t nsteps, const int64_t* gals, const double* const* atk_pmats,
                           const pz_glwe_op_params* tp, const CbtRepack* rp, int count, int tsz, int gsz, int cols) {
    const long long n = (long long)M->n, ct_t = n * cols * tsz;
    size_t log_n = 0;
    while (((size_t)1 << log_n) < (size_t)M->n) ++log_n;
    PZ_REQUIRE(nsteps == log_n, "circuit_bootstrapping (exponent mode): gals / atk_pmats must cover all log2(n) trace steps");
    PZ_REQUIRE(tsz == gsz, "circuit_bootstrapping (exponent mode): the GGSW must not be more precise than the GLWE of the rotation");
    PZ_REQUIRE(rp->log_gap_in >= 1 && rp->log_gap_in <= log_n && rp->log_gap_out <= log_n && rp->log_domain <= 20 &&
                   (((size_t)1 << rp->log_domain) - 1) << rp->log_gap_out < (size_t)M->n,
               "circuit_bootstrapping (exponent mode): gaps / domain out of range");
    const size_t skip = log_n - rp->log_gap_in + 1;
    PZ_TRY(glwe_trace(M, tr, log_n - skip, gals + skip, atk_pmats + skip, tp, (size_t)count));
    const size_t steps = (size_t)1 << rp->log_domain;
    const size_t rows_ct = align256((size_t)count * ct_t * 8);
    char* base = (char*)tr + rows_ct;
    std::vector<int64_t*> cts(steps);
    std::vector<uint64_t> idx(steps);
    const PolyMap pm{tsz, cols, ct_t, (long long)cols * n, n, 0};
    for (size_t sidx = 0; sidx < steps; ++sidx) {
        cts[sidx] = (int64_t*)(base + sidx * rows_ct);
        idx[sidx] = (uint64_t)(sidx << rp->log_gap_out);
        PZ_TRY(launch_rotate(M, count * tsz * cols, (const long long*)tr, pm, (long long*)cts[sidx], pm, 0, tsz * cols, nullptr, 0, 0,
                             -(long long)(sidx << rp->log_gap_in)));
    }
    int64_t* packed = (int64_t*)(base + steps * rows_ct);
    void* pack_tmp = (void*)(base + (steps + 1) * rows_ct);
    PZ_TRY(glwe_pack(M, packed, steps, idx.data(), cts.data(), rp->log_gap_out, gals, atk_pmats, tp, pack_tmp, 3 * rows_ct, (size_t)count, 0));
    *row_src = packed;
    return PZ_OK;
}

static int circuit_bootstrapping(pz_module* M, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut, const double* brk, size_t nsteps,
                                 const int64_t* gals, const double* const* atk_pmats, const double* const* tsk_pmats,
                                 const pz_circuit_bootstrapping_params* p, void* tmp, size_t tmp_bytes, size_t batch,
                                 const CbtRepack* rp = nullptr) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->res_dnum >= 1 && p->res_size >= 1 && p->atk_dnum >= 1 && p->atk_size >= 1 && p->tsk_dnum >= 1 && p->tsk_size >= 1,
               "circuit_bootstrapping: empty shape");
    PZ_REQUIRE(is_device_ptr(ggsw) && is_device_ptr(tmp), "batched entry points take device pointers");
    PZ_REQUIRE(tmp_bytes >= (rp ? pz_circuit_bootstrapping_to_exponent_tmp_bytes(M, p, rp->log_domain, batch)
                                : pz_circuit_bootstrapping_tmp_bytes(M, p, batch)),
               "circuit_bootstrapping: tmp is smaller than the *_tmp_bytes of this call");
    if (batch == 0) return PZ_OK;
    const long long n = (long long)M->n;
    const int cols = (int)p->br.rank + 1, bsz_g = (int)p->br.res_size, rsz = (int)p->res_size, tsz = (int)cbt_tmp_size(p);
    const int gsz = (int)cbt_atk_size(p);   // limbs of the rotated GLWE in the automorphism keys' base
    const int rows = (int)p->res_dnum, B = (int)batch;
    const int k_brk = (int)p->br.base2k, k_atk = (int)cbt_base(p, p->atk_base2k), k_tsk = (int)cbt_base(p, p->tsk_base2k),
              k_res = (int)cbt_base(p, p->res_base2k);
    PZ_REQUIRE(k_atk >= 1 && k_atk <= 63 && k_tsk >= 1 && k_tsk <= 63 && k_res >= 1 && k_res <= 63, "circuit_bootstrapping: base2k out of range");
    PZ_REQUIRE(k_atk != k_brk || gsz == bsz_g, "circuit_bootstrapping: atk_glwe_size must be br.res_size when the bases are equal");
    PZ_REQUIRE(tsz >= gsz, "circuit_bootstrapping: trace_size below atk_glwe_size");
    const long long ct_b = n * cols * bsz_g, ct_g = n * cols * gsz, ct_t = n * cols * tsz, ct_r = n * cols * rsz;
    int64_t* acc = (int64_t*)tmp;
    int64_t* acc_brk = acc;
    char* after_acc = (char*)tmp + align256((size_t)B * ct_b * 8);
    if (k_atk != k_brk) { acc = (int64_t*)after_acc; after_acc += align256((size_t)B * ct_g * 8); }
    int64_t* tr = (int64_t*)after_acc;
    if (cbt_ext(p) > 1) {  // key.brk.execute dispatches on lut.extension_factor() (algorithm.rs:76-118); the scratch sits behind ours
        const size_t eb = align256(pz_blind_rotation_extended_tmp_bytes(M, &p->br, cbt_ext(p), batch));
        void* etmp = (char*)tmp + (rp ? pz_circuit_bootstrapping_to_exponent_tmp_bytes(M, p, rp->log_domain, batch)
                                      : pz_circuit_bootstrapping_tmp_bytes(M, p, batch)) - eb;
        PZ_TRY(blind_rotation_extended(M, acc_brk, lwe_2n, lut, brk, &p->br, cbt_ext(p), etmp, eb, batch));
    } else {
        PZ_TRY(blind_rotation(M, acc_brk, lwe_2n, lut, brk, &p->br, batch));
    }
    if (k_atk != k_brk) {   // circuit.rs:326-330 glwe_normalize into the automorphism keys' layout (operations/glwe.rs:1286-1310)
        DV dv{acc, ct_g, cols, gsz}, sv{acc_brk, ct_b, cols, bsz_g};
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, dv, k_atk, 0, c, sv, k_brk, c));
    }
    if (tsz > gsz) PZ_TRY(launch_zero_bytes(M, tr, (size_t)B * rows * ct_t * 8));  // glwe_copy zero-extends (glwe_trace.rs:114)
    for (int i = 0; i < rows; ++i) {
        PolyMap sm{gsz, cols, ct_g, (long long)cols * n, n, 0};
        PolyMap dm{gsz, cols, (long long)rows * ct_t, (long long)cols * n, n, (long long)i * ct_t};
        PZ_TRY(launch_rotate(M, B * gsz * cols, (const long long*)acc, sm, (long long*)tr, dm, 0, gsz * cols, nullptr, 0, 0,
                             -(long long)i * (long long)p->gap));
    }
    pz_glwe_op_params tp;
    tp.rank = p->br.rank; tp.dnum = p->atk_dnum; tp.dsize = 1; tp.key_size = p->atk_size; tp.key_base2k = (uint64_t)k_atk;
    tp.a_size = (uint64_t)tsz; tp.a_base2k = (uint64_t)k_atk; tp.res_size = (uint64_t)tsz; tp.res_base2k = (uint64_t)k_atk; tp.rank_out = p->br.rank;
    const int64_t* row_src = tr;
    if (!rp) {
        PZ_TRY(glwe_trace(M, tr, nsteps, gals, atk_pmats, &tp, (size_t)B * rows));
    } else {
        PZ_TRY(cbt_repack_rows(M, tr, &row_src, nsteps, gals, atk_pmats, &tp, rp, B * rows, tsz, gsz, cols));
    }
    if (k_res == k_atk) {
        // glwe_copy(res.at(i, 0), tmp) (glwe_trace.rs:121-123): the first res_size limbs, into the strided (row, 0) entries
        PZ_REQUIRE(rsz <= tsz, "circuit_bootstrapping: res_size above trace_size");
        PZ_TRY(launch_ew(M, EW_COPY, ggsw, (long long)cols * ct_r, n, row_src, ct_t, n, nullptr, 0, 0, cols * rsz, B * rows));
    } else {
        // glwe_normalize(res.at(i, 0), tmp) (glwe_trace.rs:124-126)
        DV dv{ggsw, (long long)cols * ct_r, cols, rsz}, sv{(int64_t*)row_src, ct_t, cols, tsz};
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B * rows, dv, k_res, 0, c, sv, k_atk, c));
    }
    pz_glwe_op_params ep = tp;
    ep.dnum = p->tsk_dnum; ep.key_size = p->tsk_size; ep.a_size = (uint64_t)rsz; ep.res_size = (uint64_t)rsz;
    ep.key_base2k = (uint64_t)k_tsk; ep.a_base2k = (uint64_t)k_res; ep.res_base2k = (uint64_t)k_res;
    return ggsw_expand_row(M, ggsw, p->res_dnum, tsk_pmats, &ep, batch);
}
int pz_circuit_bootstrapping_execute_to_constant_batched(pz_module* M, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut,
                                                         const double* brk, size_t nsteps, const int64_t* gals,
                                                         const double* const* atk_pmats, const double* const* tsk_pmats,
                                                         const pz_circuit_bootstrapping_params* p, void* tmp, size_t tmp_bytes,
                                                         size_t batch) {
    PZ_ENTER(M);
    KeyHash k;
    k.add((int)3); k.add(ggsw); k.add(lwe_2n); k.add(lut); k.add(brk); k.add(nsteps); k.add(tmp); k.add(tmp_bytes); k.add(batch);
    if (p) {
        k.add(*p);
        for (size_t s = 0; s < nsteps && gals && atk_pmats; ++s) { k.add(gals[s]); k.add(atk_pmats[s]); }
        for (size_t c = 0; c < p->br.rank && tsk_pmats; ++c) k.add(tsk_pmats[c]);
    }
    graph_key_module(M, k);
    return with_graph(M, k.h, [&]() {
        return circuit_bootstrapping(M, ggsw, lwe_2n, lut, brk, nsteps, gals, atk_pmats, tsk_pmats, p, tmp, tmp_bytes, batch);
    });
}

// circuit_bootstrapping_execute_to_exponent (circuit.rs:197-216): equal gaps = the partial trace of post_process (:418-420),
// otherwise the repacking branch (:392-417).  gals / atk_pmats cover all log2(n) trace steps.
int pz_circuit_bootstrapping_execute_to_exponent_batched(pz_module* M, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut,
                                                         const double* brk, const int64_t* gals, const double* const* atk_pmats,
                                                         const double* const* tsk_pmats, const pz_circuit_bootstrapping_params* p,
                                                         size_t log_gap_in, size_t log_gap_out, size_t log_domain, void* tmp,
                                                         size_t tmp_bytes, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(gals != nullptr && atk_pmats != nullptr, "circuit_bootstrapping: null argument");
    size_t log_n = 0;
    while (((size_t)1 << log_n) < (size_t)M->n) ++log_n;
    PZ_REQUIRE(log_gap_in >= 1 && log_gap_in <= log_n, "circuit_bootstrapping (exponent mode): log_gap_in out of range");
    if (log_gap_in == log_gap_out) {
        const size_t skip = log_n - log_gap_in + 1;
        return circuit_bootstrapping(M, ggsw, lwe_2n, lut, brk, log_n - skip, gals + skip, atk_pmats + skip, tsk_pmats, p, tmp, tmp_bytes, batch);
    }
    CbtRepack rp{log_gap_in, log_gap_out, log_domain};
    return circuit_bootstrapping(M, ggsw, lwe_2n, lut, brk, log_n, gals, atk_pmats, tsk_pmats, p, tmp, tmp_bytes, batch, &rp);
}

// ------------------------------------------------------------------------------
// public: GLWEPacking::glwe_pack (poulpy-core/src/glwe_packing.rs:122-176, pack_internal :15-87) on `batch` independent packing
// problems with the same occupancy pattern, one base2k / size for ciphertexts, keys and result.  cts[s] points to the `batch`
// contiguous GLWEs of index indices[s] (the reference's HashMap entry); they are clobbered, as the reference's `&mut` entries.
// The tree is walked on the host exactly as the reference does; every step is a batched launch of the i64 kernels
// (rotate, add / sub, rsh, normalize) or of the fused automorphism pipeline.
// ------------------------------------------------------------------------------
size_t pz_glwe_pack_tmp_bytes(const pz_module* M, const pz_glwe_op_params* p, size_t batch) {
    if (!M || !p) return 0;
    return 3 * align256(batch * (size_t)M->n * (p->rank + 1) * p->res_size * 8);
}
size_t pz_glwe_pack_bases_tmp_bytes(const pz_module* M, const pz_glwe_op_params* p, size_t trace_size, size_t batch) {
    if (!M || !p) return 0;
    return pz_glwe_pack_tmp_bytes(M, p, batch) + align256(batch * (size_t)M->n * (p->rank + 1) * trace_size * 8);
}
// trace_size = 0: ciphertexts, keys and result share one base2k.  Otherwise the keys have their own (test_suite/glwe_packing.rs:40-42):
// pack_internal's arithmetic stays in the ciphertexts' base, the automorphisms convert, and the closing glwe_trace (glwe_trace.rs:91-127)
// runs on a temporary of trace_size limbs in the keys' base (behind the three ciphertext arrays of tmp).
// The limb-wise steps of glwe_pack on `B` ciphertexts at a time (poulpy-core glwe_packing.rs:41-86), and one merge of two slots
struct PackStep {
    pz_module* M;
    const pz_glwe_op_params* p;
    size_t batch;
    long long n, ct;
    int cols, size, k, B;
    int64_t *tmp_b, *t1, *t2;
    PolyMap pm() const { return PolyMap{size, cols, ct, (long long)cols * n, n, 0}; }
    int rotate_to(long long kk, int64_t* dst, const int64_t* src) {
        return launch_rotate(M, B * size * cols, (const long long*)src, pm(), (long long*)dst, pm(), 0, size * cols, nullptr, 0, 0, kk);
    }
    int rotate_assign(long long kk, int64_t* x) {
        PZ_TRY(rotate_to(kk, t1, x));
        return launch_ew(M, EW_COPY, x, ct, n, t1, ct, n, nullptr, 0, 0, cols * size, B);
    }
    int ew3(int op, int64_t* r, const int64_t* x, const int64_t* y) {   // limb-wise over whole ciphertexts (equal sizes)
        return launch_ew(M, op, r, ct, n, x, ct, n, y, ct, n, cols * size, B);
    }
    int rsh1(int64_t* x) { return launch_rsh(M, B, (long long*)x, ct, cols, size, 0, cols, k, 1); }
    int normalize_assign(int64_t* x) {
        PZ_TRY(launch_ew(M, EW_COPY, t2, ct, n, x, ct, n, nullptr, 0, 0, cols * size, B));
        DV xv{x, ct, cols, size}, tv{t2, ct, cols, size};
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, xv, k, 0, c, tv, k, c));
        return PZ_OK;
    }
    // slots j (a) and j + tt (b) of one level -> *out (null when both are empty)
    int merge(int64_t* a, int64_t* b, size_t tt, int64_t gal, const double* key, int64_t** out) {
        *out = nullptr;
        if (a && b) {                                                       // :41-70
            PZ_TRY(rotate_assign(-(long long)tt, a));
            PZ_TRY(ew3(EW_SUB_I64, tmp_b, a, b));
            PZ_TRY(rsh1(tmp_b));
            PZ_TRY(ew3(EW_ADD_I64, a, a, b));
            PZ_TRY(rsh1(a));
            PZ_TRY(normalize_assign(tmp_b));
            AutoSpec au{(long long)gal, 0};
            PZ_TRY(glwe_op(M, true, tmp_b, tmp_b, key, p, batch, &au));
            PZ_TRY(ew3(EW_SUB_I64, a, a, tmp_b));
            PZ_TRY(normalize_assign(a));
            PZ_TRY(rotate_assign((long long)tt, a));
            *out = a;
        } else if (a) {                                                     // :71-75
            PZ_TRY(rsh1(a));
            AutoSpec au{(long long)gal, 1};
            PZ_TRY(glwe_op(M, true, a, a, key, p, batch, &au));
            *out = a;
        } else if (b) {                                                     // :76-86
            PZ_TRY(rotate_to((long long)tt, tmp_b, b));
            PZ_TRY(rsh1(tmp_b));
            AutoSpec au{(long long)gal, 3};
            PZ_TRY(glwe_op(M, true, b, tmp_b, key, p, batch, &au));
            *out = b;
        }
        return PZ_OK;
    }
};
static int glwe_pack(pz_module* M, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out,
                     const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p, void* tmp, size_t tmp_bytes,
                     size_t batch, size_t trace_size) {
    PZ_REQUIRE(p != nullptr && indices != nullptr && cts != nullptr && gals != nullptr && key_pmats != nullptr, "glwe_pack: null argument");
    PZ_REQUIRE(p->a_size == p->res_size && p->a_base2k == p->res_base2k && p->rank_out == p->rank && p->dsize == 1,
               "glwe_pack: ciphertexts and result share base2k and size");
    PZ_REQUIRE(p->res_base2k == p->key_base2k || trace_size >= 1,
               "glwe_pack: keys in another base than the ciphertexts need pz_glwe_pack_bases_batched (trace_size)");
    if (p->res_base2k == p->key_base2k) trace_size = 0;
    size_t log_n = 0;
    while (((size_t)1 << log_n) < (size_t)M->n) ++log_n;
    PZ_REQUIRE(log_gap_out <= log_n && nslots >= 1, "glwe_pack: bad shape");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(tmp), "batched entry points take device pointers");
    PZ_REQUIRE(tmp_bytes >= pz_glwe_pack_bases_tmp_bytes(M, p, trace_size, batch), "glwe_pack: tmp is smaller than pz_glwe_pack[_bases]_tmp_bytes");
    if (batch == 0) return PZ_OK;
    PackStep st;
    st.M = M; st.p = p; st.batch = batch; st.n = (long long)M->n;
    st.cols = (int)p->rank + 1; st.size = (int)p->res_size; st.k = (int)p->res_base2k; st.B = (int)batch;
    st.ct = st.n * st.cols * st.size;
    const long long n = st.n, ct = st.ct;
    const int cols = st.cols, size = st.size, k = st.k, B = st.B;
    const size_t ctb = align256((size_t)B * ct * 8);
    st.tmp_b = (int64_t*)tmp;
    st.t1 = (int64_t*)((char*)tmp + ctb);
    st.t2 = (int64_t*)((char*)tmp + 2 * ctb);
    std::vector<int64_t*> slots((size_t)M->n, nullptr);
    for (size_t s_ = 0; s_ < nslots; ++s_) {
        PZ_REQUIRE(indices[s_] < (uint64_t)M->n, "glwe_pack: index out of range");   // glwe_packing.rs:138
        PZ_REQUIRE(cts[s_] != nullptr && is_device_ptr(cts[s_]) && slots[indices[s_]] == nullptr, "glwe_pack: bad or duplicate entry");
        slots[indices[s_]] = cts[s_];
    }
    for (size_t i = 0; i + log_gap_out < log_n; ++i) {
        const size_t tt = (size_t)1 << (log_n - 1 - i);
        PZ_REQUIRE((gals[i] & 1) != 0 && key_pmats[i] != nullptr, "glwe_pack: bad automorphism key");
        for (size_t j = 0; j < tt; ++j) {
            int64_t* a = slots[j];
            int64_t* b = slots[j + tt];
            slots[j + tt] = nullptr;
            PZ_TRY(st.merge(a, b, tt, gals[i], key_pmats[i], &slots[j]));
        }
    }
    PZ_REQUIRE(slots[0] != nullptr, "glwe_pack: no ciphertext ends at index 0");   // :175 a.get(&0).unwrap()
    const size_t skip = log_n - log_gap_out;
    if (trace_size == 0) {   // glwe_copy both ways
        PZ_TRY(launch_ew(M, EW_COPY, res, ct, n, slots[0], ct, n, nullptr, 0, 0, cols * size, B));
        return glwe_trace(M, res, log_n - skip, gals + skip, key_pmats + skip, p, batch);
    }
    const int kk = (int)p->key_base2k, tsz = (int)trace_size;
    const long long ct_t = n * cols * tsz;
    int64_t* tr = (int64_t*)((char*)tmp + 3 * ctb);
    DV tv{tr, ct_t, cols, tsz}, sv{slots[0], ct, cols, size}, rv{res, ct, cols, size};
    for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, tv, kk, 0, c, sv, k, c));
    pz_glwe_op_params q = *p;
    q.a_size = q.res_size = (uint64_t)tsz; q.a_base2k = q.res_base2k = p->key_base2k;
    PZ_TRY(glwe_trace(M, tr, log_n - skip, gals + skip, key_pmats + skip, &q, batch));
    for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, rv, k, 0, c, tv, kk, c));
    return PZ_OK;
}
int pz_glwe_pack_batched(pz_module* M, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out,
                         const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p, void* tmp, size_t tmp_bytes,
                         size_t batch) {
    PZ_ENTER(M);
    return glwe_pack(M, res, nslots, indices, cts, log_gap_out, gals, key_pmats, p, tmp, tmp_bytes, batch, 0);
}
int pz_glwe_pack_bases_batched(pz_module* M, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out,
                               const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p, size_t trace_size, void* tmp,
                               size_t tmp_bytes, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(trace_size >= 1, "glwe_pack: trace_size = 0");
    return glwe_pack(M, res, nslots, indices, cts, log_gap_out, gals, key_pmats, p, tmp, tmp_bytes, batch, trace_size);
}

}  // extern "C"
