// br_forms.hpp — the instantiations of the one-kernel blind rotation (k_br_fused, device_br.hpp) and the choice among them.  Included by two
// translation units: launch_br.hip instantiates the product forms, launch_br_probe.hip the same forms with the rounding-margin probe compiled
// in (a run-time test in the carry phase cost the reference's bench shape 5 %, profiles/r05_ab_br_probe.txt).
//
// Forms (round 5: the census of what the GPU suite and every bench shape dispatch, tools/dbg/dispatch_census.sh, instead of the full cross
// product of round 4 - 80 instantiations, 54 of them spilling, 14 ever dispatched):  rows x output-column groups (MR, CG) in {(4, 4), (6, 3)};
// PJ = 2 product jobs per thread only where m * ceil(ncols / CG) > 512 can happen (m = 512); two ciphertexts per workgroup (CT = 2, with 64-
// or 32-bit accumulators) for m >= 256; execute_standard (STD, block size 1) with one ciphertext.  A shape outside this list - more than 6
// key rows, 7 - 8 output polynomials with 5 - 6 rows - runs the composed path (block step + transforms), which covers every shape.
#pragma once
#include <algorithm>

#include "internal.hpp"
#include "device_br.hpp"

namespace pz {

struct BrFusedPlan {
    int r0 = 0, ct = 1, pj = 1, mr = 4, cg = 4, nt = 512;
    bool a32 = false, std_variant = false;
    size_t lds = 0, pmat_doubles = 0;
};
constexpr int kBrNT = 512;

// (R0, CT, PJ, MR, CG, A32) of the block-binary forms and (R0, PJ, MR, CG) of the standard forms
#define PZ_BR_FORMS(X)                                                                                                       \
    X(2, 1, 1, 4, 4, false) X(2, 1, 1, 6, 3, false)                                                                          \
    X(4, 1, 1, 4, 4, false) X(4, 1, 1, 6, 3, false) X(4, 2, 1, 4, 4, false) X(4, 2, 1, 6, 3, false) X(4, 2, 1, 4, 4, true) X(4, 2, 1, 6, 3, true) \
    X(8, 1, 1, 4, 4, false) X(8, 1, 1, 6, 3, false) X(8, 1, 2, 4, 4, false) X(8, 1, 2, 6, 3, false)                          \
    X(8, 2, 1, 4, 4, false) X(8, 2, 1, 6, 3, false) X(8, 2, 2, 6, 3, false)                                                  \
    X(8, 2, 1, 4, 4, true) X(8, 2, 1, 6, 3, true) X(8, 2, 2, 6, 3, true)   /* (8, 2, 2, 4, 4, *): 16 work polynomials of 512 points never fit */
// (R0, PJ, MR, CG) of the 256-thread forms: one ciphertext per workgroup, two workgroups per CU whose barriers do not line up
#define PZ_BR_HALF_FORMS(X) X(4, 2, 4, 4, false) X(4, 2, 6, 3, false)
#define PZ_BR_STD_FORMS(X)                                                                                                   \
    X(2, 1, 4, 4) X(2, 1, 6, 3) X(4, 1, 4, 4) X(4, 1, 6, 3) X(8, 1, 4, 4) X(8, 1, 6, 3) X(8, 2, 4, 4) X(8, 2, 6, 3)

inline bool br_form_exists(const BrFusedPlan& pl) {
    if (pl.std_variant) {
#define X(R0_, PJ_, MR_, CG_) if (pl.r0 == R0_ && pl.ct == 1 && pl.pj == PJ_ && pl.mr == MR_ && pl.cg == CG_ && !pl.a32) return true;
        PZ_BR_STD_FORMS(X)
#undef X
        return false;
    }
    if (pl.nt == 256) {
#define X(R0_, PJ_, MR_, CG_, A32_) if (pl.r0 == R0_ && pl.ct == 1 && pl.pj == PJ_ && pl.mr == MR_ && pl.cg == CG_ && pl.a32 == A32_) return true;
        PZ_BR_HALF_FORMS(X)
#undef X
        return false;
    }
#define X(R0_, CT_, PJ_, MR_, CG_, A32_) if (pl.r0 == R0_ && pl.ct == CT_ && pl.pj == PJ_ && pl.mr == MR_ && pl.cg == CG_ && pl.a32 == A32_) return true;
    PZ_BR_FORMS(X)
#undef X
    return false;
}

// does the rotation run as one kernel, and as which form?  false: the composed path
inline bool br_fused_plan(const pz_module* M, const pz_blind_rotation_params* p, size_t batch, BrFusedPlan* out) {
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1, dnum = (int)p->dnum, bsz = (int)p->brk_size, rsz = (int)p->res_size;
    const int B = (int)batch, blk = (int)p->block_size, k = (int)p->base2k;
    BrFusedPlan pl;
    pl.pmat_doubles = (size_t)n * dnum * cols * cols * bsz;
    pl.std_variant = blk == 1;   // execute_standard: one ciphertext per workgroup and a second accumulator-sized array
    const int in_limbs = std::min(dnum, rsz), row_max = cols * in_limbs, ncols = cols * bsz, P = std::max(row_max, ncols);
    const int m = (int)M->m, mp = m + (m >> 4);
    constexpr int NT = kBrNT;
    pl.r0 = m == 128 ? 2 : (m == 256 ? 4 : 8);
    auto lds_for = [&](int ct, bool a32) {
        return ((size_t)m + (size_t)ct * P * mp) * sizeof(cplx) + (size_t)ct * rsz * cols * (size_t)n * (a32 ? 4 : 8) * (pl.std_variant ? 2 : 1);
    };
    auto fits = [&](int ct, bool a32) {
        return lds_for(ct, a32) <= 160 * 1024 && ct * P * (m / 8) <= ((ct == 2 && pl.r0 == 8) ? 2 : 1) * NT && ct * P * (m / pl.r0) <= 2 * NT &&
               (!a32 || k <= 31);
    };
    pl.cg = (ncols % 3 == 0 && ncols % 4 != 0) ? 3 : 4;  // 6 output polynomials: two groups of 3, no idle slot
    pl.pj = m * ((ncols + pl.cg - 1) / pl.cg) <= NT ? 1 : 2;
    if (!(M->fuse_mid && (m == 128 || m == 256 || m == 512) && NT % m == 0 && row_max <= 8 && ncols <= 8 && blk <= 64 &&   // (blk: one lane per rotation amount)
          m * ((ncols + pl.cg - 1) / pl.cg) <= 2 * NT && fits(1, false)))
        return false;
    // two ciphertexts per workgroup share every key value; with 64-bit accumulators when that fits in LDS, else with 32-bit digit
    // accumulators (m = 128 is only built with one ciphertext per workgroup)
    // Ciphertexts per workgroup against the batch (profiles/r05_ab_br_form.txt, N = 512 / 1024): a workgroup with two ciphertexts shares every key
    // value but takes 1.3 - 1.6 x as long as one with one, so a batch that gives every ciphertext its own CU runs one per workgroup (batch 64 -
    // 256: 2.4 instead of 4.0 ms); above that two.  POULPY_DBG_BR_FORM (experiment builds): 1 one ciphertext per workgroup, 2 two, 3 the
    // 256-thread form below.
    static const int form = exp_knob("POULPY_DBG_BR_FORM", 0);
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, M->device);
    if (M->cu_count > 0) ncu = M->cu_count;
    const bool two = form == 2 || (form == 0 && B > ncu);
    if (B >= 2 && m != 128 && !pl.std_variant && two) {
        if (fits(2, false)) pl.ct = 2;
        else if (fits(2, true)) { pl.ct = 2; pl.a32 = true; }
    }
    pl.lds = lds_for(pl.ct, pl.a32);
    pl.mr = row_max <= 4 && pl.cg == 4 ? 4 : (row_max <= 6 && pl.cg == 3 ? 6 : 8);
    if (!br_form_exists(pl)) {
        // two ciphertexts per workgroup without a form: one per workgroup may have one
        BrFusedPlan one = pl;
        one.ct = 1; one.a32 = false; one.lds = lds_for(1, false);
        if (pl.ct == 2 && br_form_exists(one)) pl = one;
        else return false;
    }
    // m = 256, a batch whose last wave of two-ciphertext workgroups would be at most half full (B mod 2 CUs in 1 .. CUs): one ciphertext per
    // 256-thread workgroup, two workgroups per CU, fills it (768: 6.87 instead of 7.94 ms).  At a multiple of 2 x CUs this form ran at the SAME
    // rate as the two-ciphertext workgroup before the latter's per-ciphertext passes were merged (129 700 vs 129 600 rotations/s at 1024: neither
    // the barriers shared by eight waves nor the shared key values are what bounds this kernel); a third workgroup per CU (32-bit accumulators,
    // 168 registers with 150 - 250 B of scratch) gave + 2 % at multiples of 3 x CUs and lost elsewhere: not built.
    const int ragged = B % (2 * ncu);
    if ((form == 3 || (form == 0 && B > 2 * ncu && ragged >= 1 && ragged <= ncu)) && m == 256 && !pl.std_variant) {
        BrFusedPlan h = pl;
        h.nt = 256; h.ct = 1; h.a32 = false; h.lds = lds_for(1, false);
        h.pj = (m * ((ncols + pl.cg - 1) / pl.cg) + 255) / 256;
        if (P * (m / 8) <= 256 && P * (m / pl.r0) <= 2 * 256 && 2 * h.lds <= 160 * 1024 && br_form_exists(h)) pl = h;
    }
    *out = pl;
    return true;
}

template <bool PROBE>
static int br_fused_launch(pz_module* M, const BrFusedArgs& g, const BrFusedPlan& pl) {
    constexpr int NT = kBrNT;
    const int B = g.batch;
    if (pl.std_variant) {
#define X(R0_, PJ_, MR_, CG_)                                                                                                \
    if (pl.r0 == R0_ && pl.pj == PJ_ && pl.mr == MR_ && pl.cg == CG_) {                                                      \
        PZ_TRY(set_lds((k_br_fused<R0_, 1, NT, PJ_, MR_, CG_, false, true, PROBE>), pl.lds));                                \
        hipLaunchKernelGGL((k_br_fused<R0_, 1, NT, PJ_, MR_, CG_, false, true, PROBE>), dim3(B), dim3(NT), pl.lds, M->stream, g); \
        dispatch_note(M, "k_br_fused<R0=%d,CT=1,NT=512,PJ=%d,MR=%d,CG=%d,A32=0,STD=1> lds=%zu", R0_, PJ_, MR_, CG_, pl.lds); \
        return PZ_OK;                                                                                                        \
    }
        PZ_BR_STD_FORMS(X)
#undef X
        return fail(PZ_ERR_UNSUPPORTED, "blind_rotation: no one-kernel standard form for this plan");
    }
    if (pl.nt == 256) {
#define X(R0_, PJ_, MR_, CG_, A32_)                                                                                          \
    if (pl.r0 == R0_ && pl.pj == PJ_ && pl.mr == MR_ && pl.cg == CG_ && pl.a32 == A32_) {                                    \
        PZ_TRY(set_lds((k_br_fused<R0_, 1, 256, PJ_, MR_, CG_, A32_, false, PROBE>), pl.lds));                               \
        hipLaunchKernelGGL((k_br_fused<R0_, 1, 256, PJ_, MR_, CG_, A32_, false, PROBE>), dim3(B), dim3(256), pl.lds, M->stream, g); \
        dispatch_note(M, "k_br_fused<R0=%d,CT=1,NT=256,PJ=%d,MR=%d,CG=%d,A32=%d> lds=%zu", R0_, PJ_, MR_, CG_, (int)(A32_), pl.lds); \
        return PZ_OK;                                                                                                        \
    }
        PZ_BR_HALF_FORMS(X)
#undef X
        return fail(PZ_ERR_UNSUPPORTED, "blind_rotation: no 256-thread one-kernel form for this plan");
    }
#define X(R0_, CT_, PJ_, MR_, CG_, A32_)                                                                                     \
    if (pl.r0 == R0_ && pl.ct == CT_ && pl.pj == PJ_ && pl.mr == MR_ && pl.cg == CG_ && pl.a32 == A32_) {                   \
        PZ_TRY(set_lds((k_br_fused<R0_, CT_, NT, PJ_, MR_, CG_, A32_, false, PROBE>), pl.lds));                              \
        hipLaunchKernelGGL((k_br_fused<R0_, CT_, NT, PJ_, MR_, CG_, A32_, false, PROBE>), dim3((B + CT_ - 1) / CT_), dim3(NT), pl.lds, M->stream, g); \
        dispatch_note(M, "k_br_fused<R0=%d,CT=%d,NT=512,PJ=%d,MR=%d,CG=%d,A32=%d> lds=%zu", R0_, CT_, PJ_, MR_, CG_, (int)(A32_), pl.lds); \
        return PZ_OK;                                                                                                        \
    }
    PZ_BR_FORMS(X)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "blind_rotation: no one-kernel form for this plan");
}
// launch_br_probe.hip
int br_fused_launch_probe(pz_module* M, const BrFusedArgs& g, const BrFusedPlan& pl);

}  // namespace pz
