// launch_tail.hip — dispatch of the fused tail (k_inv_tail, device_fft.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "tail_forms.hpp"

namespace pz {

// the two roles of k_inv_tail must be whole waves
bool tail_supported(const pz_module* M) { return (M->plan.f1b * M->plan.cb) % 64 == 0; }
bool tail_acc32_supported(const pz_module* M) {
#define X(A, B, C) if (M->plan.f1a == A && M->plan.f1b == B && M->plan.cb == C) return true;
    PZ_ACC32_CASES(X)
#undef X
    return false;
}
bool tail_d16_only_supported(const pz_module* M) { return tail_rsh_supported(M) && tail_acc32_supported(M); }
bool tail_rsh_supported(const pz_module* M) {
#define X(A, B, C) if (M->plan.f1a == A && M->plan.f1b == B && M->plan.cb == C) return true;
    PZ_RSH_CASES(X)
#undef X
    return false;
}

struct TailNz { int lsh, res_end, res_start, a_end, a_start, zero_from, col, mode, col2[2], mode2[2]; TailD16 d16; };
// columns [col_base, col_base + col_count) of the big value in one launch; raw / nz: the tensoring forms (launch_inv_tail_raw / _nz)
static TailArgs tail_args(const pz_module* M, const TailCall& c, int col_base, int col_count, bool raw, const TailNz* nz) {
    TailArgs g;
    g.T = c.T; g.res = c.res; g.small = c.small; g.res_bs = c.res_bs; g.small_bs = c.small_bs;
    g.nlimbs = c.nlimbs; g.ncols = c.ncols; g.res_cols = c.res_cols; g.res_size = c.res_size;
    g.small_cols = c.small_cols; g.small_size = c.small_size; g.base2k = c.base2k; g.m2 = M->plan.m2;
    g.tw1inv = M->tw1inv; g.wL1 = M->wL1; g.margin = M->probe ? M->margin : nullptr;
    g.small_all = c.small_all ? 1 : 0; g.auto_mul = c.auto_mul; g.auto_neg = c.auto_neg ? 1 : 0;
    g.col_base = col_base; g.col_count = col_count; g.body_col = c.body_col;
    g.gather_mul = c.gather_mul; g.gather_neg = c.gather_neg ? 1 : 0;
    g.pre_body = (c.body_src != nullptr || c.body_gather) ? 1 : 0; g.small_neg = c.small_neg ? 1 : 0;
    g.body_src = c.body_src; g.body_bs = c.body_bs; g.body_ls = c.body_ls;
    g.body_add = c.body_add ? 1 : 0; g.post_neg = c.post_neg ? 1 : 0; g.body_only = c.body_only ? 1 : 0; g.raw = raw ? 1 : 0;
    g.nz = nz ? 1 : 0;
    g.nz_lsh = nz ? nz->lsh : 0; g.nz_res_end = nz ? nz->res_end : 0; g.nz_res_start = nz ? nz->res_start : 0; g.nz_a_end = nz ? nz->a_end : 0;
    g.nz_a_start = nz ? nz->a_start : 0; g.nz_zero_from = nz ? nz->zero_from : 0; g.nz_col = nz ? nz->col : 0; g.nz_mode = nz ? nz->mode : 0;
    for (int u = 0; u < 2; ++u) { g.nz_col2[u] = nz ? nz->col2[u] : 0; g.nz_mode2[u] = nz ? nz->mode2[u] : 0; }
    g.acc32 = c.acc32;
    g.xcd_map = 0;
    g.d16w = nz ? nz->d16.w : nullptr; g.d16a = nz ? nz->d16.ra : nullptr; g.d16b = nz ? nz->d16.rb : nullptr;
    if (c.acc32 & 4) { g.d16a = c.small16; g.body_bs = c.small16_cs; }
    g.body16_wide = nullptr;
    if (c.body16) { g.d16a = c.body16; g.body_bs = (long long)c.body16_limbs * (long long)M->n; g.body16_wide = c.body16_wide; }
    return g;
}
static int launch_inv_tail_cols(pz_module* M, const TailCall& c, int col_base, int col_count, bool raw = false, const TailNz* nz = nullptr) {
    const FftPlan& pl = M->plan;
    int blocks = c.batch * col_count * (pl.m2 / pl.cb);
    if (blocks == 0) return PZ_OK;
    KTimer kt(M, PZ_K_FUSED_TAIL);
    TailArgs g = tail_args(M, c, col_base, col_count, raw, nz);
    // XCD-aware block order (all column blocks of one (ciphertext, column) on one XCD, back to back): the gathers of the automorphism
    // forms need it for L2 locality, and the row-major pipeline streams faster with it (see k_fwd_pass1); the grid is padded to whole
    // groups of 8 (ciphertext, column) pairs
    if (c.gather_mul != 0 || c.rowmajor) {
        const int nbc = c.batch * col_count, ncb = pl.m2 / pl.cb;
        g.xcd_map = nbc;
        blocks = ((nbc + 7) / 8) * 8 * ncb;
    }
    // which form of the kernel: shifted store | sign-only | tensoring (pairwise / diagonal) | plain (with or without the body operand)
    const bool has_small = c.small != nullptr;
    TailForm f;
    f.rowmajor = c.rowmajor; f.has_small = has_small;
    if (c.acc32) {   // 32-bit accumulator digits: the plain every-column-operand form, nothing else
        if (!(tail_acc32_supported(M) && c.rowmajor && has_small && c.small_all && !c.post_rsh && !raw && !nz && c.auto_mul == 0 && c.gather_mul == 0 &&
              c.body_src == nullptr && !c.body_gather && c.base2k <= 31 && (!(c.acc32 & 4) || (c.acc32 == 4 && c.small16 != nullptr))))
            return fail(PZ_ERR_UNSUPPORTED, "fused tail: no 32-bit-accumulator variant for this call");
        f.kind = TailForm::ACC32;
    } else if (c.post_rsh && c.body16 && !has_small) {   // the 16-bit-operand form with the shifted store (glwe_trace's body column)
        if (!(tail_rsh_supported(M) && c.rowmajor && !raw && !nz && c.base2k <= 29)) return fail(PZ_ERR_UNSUPPORTED, "fused tail: no shifted-store 16-bit-operand variant for this call");
        f.kind = TailForm::SGN16R;
        g.small_size = c.small_size;
    } else if (c.post_rsh) {
        if (!(tail_rsh_supported(M) && c.rowmajor && has_small)) return fail(PZ_ERR_UNSUPPORTED, "fused tail: no shifted-store variant for this plan");
        f.kind = TailForm::RSH;
    } else if (!has_small && (c.auto_mul != 0 || c.body16)) {   // signs without an operand (launch_inv_tail: the body-less columns of a plain spectral automorphism)
        if (!(c.rowmajor && !raw && !nz && tail_rsh_supported(M))) return fail(PZ_ERR_UNSUPPORTED, "fused tail: no sign-only variant for this plan");
        f.kind = c.body16 ? TailForm::SGN16 : TailForm::SGN;   // (+ the 16-bit operand of the body column, launch_inv_tail)
        if (c.body16) { g.small_size = c.small_size; if (c.base2k > 31) return fail(PZ_ERR_UNSUPPORTED, "fused tail: the 16-bit-operand form needs digits of at most 31 bits"); }
    } else if (raw || nz) {   // the tensoring forms: their own instantiation of the row-major, operand-free tail
        if (!(c.rowmajor && !has_small && tail_rsh_supported(M))) return fail(PZ_ERR_UNSUPPORTED, "fused tail: the tensoring forms need the row-major layout of a 128-point-row plan");
        // contract of the NZ = 1 instantiation (device_fft.hpp): plain normalized store (mode 1, no second result), shift below one limb;
        // NzCombine mode 5 reads diagonal columns that a mode-1 launch of the same call wrote before it
        if (nz && !(nz->lsh >= 0 && nz->lsh < c.base2k)) return fail(PZ_ERR_INVALID, "fused tail: normalizing store needs 0 <= lsh < base2k");
        // raw values, or any NzCombine mode (pairwise term: with the prefetch of the diagonal digits): NZ2; the diagonal launches: NZ1
        f.kind = (raw || !(nz->mode == 1 && nz->mode2[0] == 0 && nz->mode2[1] == 0)) ? TailForm::NZ2 : TailForm::NZ1;
        // 16-bit side copies (TailD16): the diagonal launch that writes them, the pairwise launch (mode 5) that reads them
        if (nz && f.kind == TailForm::NZ1 && nz->d16.w) f.kind = TailForm::NZ1W;
        if (nz && f.kind == TailForm::NZ2 && nz->d16.ra) {
            if (!(nz->mode2[0] == 5)) return fail(PZ_ERR_INVALID, "fused tail: 16-bit side copies are read by the mode-5 pairwise launch only");
            f.kind = TailForm::NZ2R;
        }
        if (nz && nz->d16.only) {   // the digits leave only as 16-bit copies
            if (f.kind == TailForm::NZ1W) f.kind = TailForm::NZ1O;
            else if (f.kind == TailForm::NZ2R && nz->d16.w) f.kind = TailForm::NZ2O;
            else return fail(PZ_ERR_INVALID, "fused tail: 16-bit-only digits need the side-copy forms (a diagonal launch that writes them, a pairwise launch that reads and writes them)");
        }
    }
    // the rounding-margin instantiation of the same form when the module's probe is on (pz_module_set_margin_probe): the same source with the
    // probe block compiled in, so that the margin is measured on the form the product path dispatches (launch_tail_probe.hip)
    return M->probe ? tail_launch_form_probe(M, g, blocks, f) : tail_launch_form<false>(M, g, blocks, f);
}
// The body operand of a key switch only exists for one column (0; `body_col` for ggsw_expand_row): that column runs the
// variant that prefetches it (more registers, one workgroup less per CU), the other columns the plain one.
int launch_inv_tail(pz_module* M, const TailCall& c) {
    if (c.post_rsh && !(c.small != nullptr && c.small_all)) return fail(PZ_ERR_UNSUPPORTED, "fused tail: shifted store needs an operand per column");
    if (c.small != nullptr && !c.small_all && c.ncols > 1) {
        TailCall body = c;       // the body column: operand, and the signs that go with it
        body.gather_mul = 0; body.gather_neg = false; body.body_src = nullptr; body.body_bs = body.body_ls = 0;
        body.small_neg = body.post_rsh = body.post_neg = body.body_only = body.body_add = false;
        PZ_TRY(launch_inv_tail_cols(M, body, c.body_col, 1));
        TailCall plain = body;   // every other column: no operand, no signs
        plain.small = nullptr; plain.small_bs = 0; plain.auto_mul = 0; plain.auto_neg = false; plain.body_col = 0;
        if (c.body_col > 0) PZ_TRY(launch_inv_tail_cols(M, plain, 0, c.body_col));
        return launch_inv_tail_cols(M, plain, c.body_col + 1, c.ncols - 1 - c.body_col);
    }
    // plain spectral glwe_automorphism (only the body column has an operand): that column on the operand variant, the others on the
    // sign-only variant of the f64 chain (POULPY_DBG_AUTO_SGN=0: every column on the operand variant, as in round 3)
    static const int sgn_knob = exp_knob("POULPY_DBG_AUTO_SGN", 1);
    if (sgn_knob && c.small != nullptr && c.small_all && c.body_only && c.auto_mul != 0 && c.ncols > 1 && !c.post_rsh && c.rowmajor &&
        tail_rsh_supported(M)) {
        // (body16: the body column twice - the operand variant returns at once unless the pre-pass raised the flag, the 16-bit-operand form
        //  of the sign-only tail returns at once if it did; exactly one of them writes the column)
        PZ_TRY(launch_inv_tail_cols(M, c, c.body_col, 1));
        TailCall rest = c;
        rest.small = nullptr; rest.small_bs = 0; rest.small_all = false;
        rest.body_src = nullptr; rest.body_bs = rest.body_ls = 0; rest.body_only = false; rest.body_add = false; rest.body_gather = false; rest.gather_mul = 0;
        if (c.body16) PZ_TRY(launch_inv_tail_cols(M, rest, c.body_col, 1));   // (small_size: the operand's limbs)
        rest.small_size = 0; rest.body16 = nullptr; rest.body16_wide = nullptr; rest.body16_limbs = 0;
        if (c.body_col > 0) PZ_TRY(launch_inv_tail_cols(M, rest, 0, c.body_col));
        return launch_inv_tail_cols(M, rest, c.body_col + 1, c.ncols - 1 - c.body_col);
    }
    // add / sub forms of the spectral automorphism with the body-column operand phi(body) +- a0 as 16-bit copies (TailCall::body16): that column as
    // above - the 16-bit-operand form on the f64 chain, the gathering operand variant beside it for the flag-up case - every other column on the
    // operand variant with its own +-a[col]
    if (c.body16 && c.small != nullptr && c.small_all && !c.body_only && c.rowmajor && tail_rsh_supported(M)) {
        PZ_TRY(launch_inv_tail_cols(M, c, c.body_col, 1));   // (returns at once unless the pre-pass raised the flag)
        TailCall b16 = c;
        b16.small = nullptr; b16.small_bs = 0; b16.small_all = false; b16.small_neg = false;   // (the sign of the operand is the pre-pass's)
        b16.body_src = nullptr; b16.body_bs = b16.body_ls = 0; b16.body_add = false; b16.body_gather = false; b16.gather_mul = 0; b16.gather_neg = false;
        PZ_TRY(launch_inv_tail_cols(M, b16, c.body_col, 1));
        TailCall rest = c;
        if (c.other16 && c.ncols == 2 && c.body_col == 0) {
            // the other column the same way: its +-a[1] was left as 16-bit values by pass 1 - the operand variant only if the flag is up (rest keeps
            // body16_wide), the 16-bit-operand form with the operand's sign (small_neg) otherwise
            PZ_TRY(launch_inv_tail_cols(M, rest, 1, 1));
            TailCall o16 = b16;
            o16.body16 = c.other16; o16.body16_limbs = c.small_size; o16.small_neg = c.small_neg;
            return launch_inv_tail_cols(M, o16, 1, 1);
        }
        rest.body16 = nullptr; rest.body16_wide = nullptr; rest.body16_limbs = 0;
        if (c.body_col > 0) PZ_TRY(launch_inv_tail_cols(M, rest, 0, c.body_col));
        if (c.ncols - 1 - c.body_col > 0) PZ_TRY(launch_inv_tail_cols(M, rest, c.body_col + 1, c.ncols - 1 - c.body_col));
        return PZ_OK;
    }
    return launch_inv_tail_cols(M, c, 0, c.ncols);
}

// the inverse column pass on the row-major T2' with vec_znx_normalize's same-base steps (bit offset res_offset, a.size = a_size limbs of which
// the first nlimbs are transformed and the rest are zero) and NzCombine's stores into `res` (GLWE tensoring: raw inverse pass + normalize
// kernel in one; TailArgs::nz)
int launch_inv_tail_nz(pz_module* M, int batch, const cplx* T, int nlimbs, long long* res, long long res_bs, int res_cols, int res_size, int res_col,
                       int base2k, long long res_offset, int a_size, const NzCombine* cb, const TailD16* d16) {
    const long long k = base2k;
    long long lsh = res_offset % k, lo = res_offset / k;
    if (res_offset < 0 && lsh != 0) { lsh = (lsh + k) % k; lo -= 1; }
    auto cl = [](long long v, long long lo_, long long hi_) { return v < lo_ ? lo_ : (v > hi_ ? hi_ : v); };
    TailNz nz;
    nz.lsh = (int)lsh;
    nz.res_end = (int)cl(-lo, 0, res_size);
    nz.res_start = (int)cl((long long)a_size - lo, 0, res_size);
    nz.a_end = (int)cl(lo, 0, a_size);
    nz.a_start = (int)cl((long long)res_size + lo, 0, a_size);
    nz.zero_from = nz.res_start - std::max(0, nz.a_start - std::max(nlimbs, nz.a_end));
    nz.col = res_col;
    nz.mode = cb ? cb->mode : 1;
    for (int u = 0; u < 2; ++u) { nz.col2[u] = cb ? cb->col2[u] : 0; nz.mode2[u] = cb ? cb->mode2[u] : 0; }
    if (d16) {
        PZ_REQUIRE(base2k <= 16, "fused tail: 16-bit side copies of the digits need base2k <= 16");
        PZ_REQUIRE(!d16->only || base2k <= 14, "fused tail: a pairwise column kept as 16-bit values needs base2k <= 14");
        nz.d16 = *d16;
    }
    TailCall c;
    c.batch = batch; c.T = T; c.rowmajor = true; c.nlimbs = nlimbs; c.ncols = 1;
    c.res = res; c.res_bs = res_bs; c.res_cols = res_cols; c.res_size = res_size; c.base2k = base2k;
    return launch_inv_tail_cols(M, c, 0, 1, false, &nz);
}



}  // namespace pz
