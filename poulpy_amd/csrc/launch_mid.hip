// launch_mid.hip — dispatch of the fused middle kernels and the key re-slicing (device_mid.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "device_mid.hpp"

namespace pz {

bool mid_supported(const pz_module* M, int npi, int npo) {
    const int np_max = M->plan.m2 == 128 ? 32 : 16;  // 128-point rows: up to 32 polynomial slots (two ciphertexts per tile)
    return (M->plan.m2 == 256 || M->plan.m2 == 128) && (M->plan.m1 % 16) == 0 && npi >= 1 && npi <= np_max && npo >= 1 && npo <= np_max;
}
int launch_permute_pmat(pz_module* M, const double* P, cplx* Pp, int npolys) {
    const FftPlan& pl = M->plan;
    const int blocks = npolys * (pl.m1 / 16) * (pl.m2 / 16);
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_permute_pmat, dim3(blocks), dim3(256), 0, M->stream, reinterpret_cast<const cplx*>(P), Pp, npolys, pl.m1, pl.m2);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}
template <int CT>
static int launch_mid_ct(pz_module* M, MidArgs g, int batch) {
    g.n_ct = (batch + CT - 1) / CT;
    const size_t lds = ((size_t)CT * 16 * 17 * 16 + 512) * sizeof(cplx);
    KTimer kt(M, PZ_K_FUSED_MID);
    PZ_TRY(set_lds(k_mid<CT>, lds));
    // persistent: as many workgroups as fit (LDS-bound: 144 KiB -> 1 per CU at CT = 2, 76 KiB -> 2 per CU at CT = 1)
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, M->device);
        if (M->cu_count > 0) ncu = (M->cu_count / 8) * 8 > 0 ? (M->cu_count / 8) * 8 : M->cu_count;
    const int per_cu = CT == 1 ? 2 : 1;
    const int grid = std::min(ncu * per_cu, g.m1 * g.n_ct);
    hipLaunchKernelGGL((k_mid<CT>), dim3(grid), dim3(CT * 256), lds, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}
// perm_mul != 0: spectrum permutation of X -> X^p folded into the middle kernel (m2 = 128 plans only; see MidArgs)
// m2 = 128 plans: the persistent tile kernels k_mid128 / k_mid128r - which instantiation for this shape (tile geometry by the polynomials in and
// out, the product form: plain / permuted / digit-selected / CGGI block step)
static int launch_mid128(pz_module* M, MidArgs& g, int batch, int npi, int npo, bool perm, bool ds, bool br) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, M->device);
    if (M->cu_count > 0) ncu = (M->cu_count / 8) * 8 > 0 ? (M->cu_count / 8) * 8 : M->cu_count;
    static const bool mid_r = (rt_knob("POULPY_DBG_MID_R", 1) != 0);   // 0: k_mid128 (the kernel of rounds 1-2) instead of k_mid128r (A/B)
    static const bool br_nc3 = (exp_knob("POULPY_DBG_BR_NC3", 1) != 0);   // 0: 4 outputs per thread also for 6-column block steps (A/B)
    KTimer kt(M, PZ_K_FUSED_MID);
#define PZ_MID128_GO(CT_, NP_, PERM_, SKIPW_)                                                                              \
{                                                                                                                      \
    PZ_TRY(set_lds((k_mid128<CT_, NP_, PERM_, false, false, ((SKIPW_) && (NP_ > 8))>), lds));                           \
    hipLaunchKernelGGL((k_mid128<CT_, NP_, PERM_, false, false, ((SKIPW_) && (NP_ > 8))>), grid_, dim3(512), lds, M->stream, g); \
    dispatch_note(M, "k_mid128<CT=%d,NP=%d,PERM=%d,DS=0,BR=0,SKIPW=%d>", CT_, NP_, (int)(PERM_), (int)((SKIPW_) && (NP_ > 8))); \
}
#define PZ_MID128_GOR1(CT_, NP_, PERM_, NR_, HALF_)                                                                        \
{                                                                                                                      \
    PZ_TRY(set_lds((k_mid128r<CT_, NP_, PERM_, NR_, ((HALF_) && (NP_ >= 16))>), lds));                                  \
    hipLaunchKernelGGL((k_mid128r<CT_, NP_, PERM_, NR_, ((HALF_) && (NP_ >= 16))>), grid_, dim3(512), lds, M->stream, g); \
    dispatch_note(M, "k_mid128r<CT=%d,NP=%d,PERM=%d,NR=%d,HALFIN=%d,KR=%d>", CT_, NP_, (int)(PERM_), NR_, (int)((HALF_) && (NP_ >= 16)), NP_ == 32 ? 3 : PZ_MIDR_KR); \
}
/* k_mid128r: product rows = NP (no idle waves) or NP / 2 with the upper half of the slots without input (key switch) */ \
/* or, 8-slot tile, 8 rows                                                                                             */
#define PZ_MID128_GOR(CT_, NP_, PERM_)                                                                                     \
{                                                                                                                      \
    if (g.row_max == NP_) PZ_MID128_GOR1(CT_, NP_, PERM_, NP_, false)                                                  \
    else if (NP_ >= 16 && npi <= NP_ / 2) PZ_MID128_GOR1(CT_, NP_, PERM_, ((NP_ >= 16) ? NP_ / 2 : NP_), true)          \
    else PZ_MID128_GOR1(CT_, NP_, PERM_, ((NP_ >= 16) ? NP_ / 2 : NP_), false)                                         \
}
/* plain product: the interleaved kernel k_mid128r where it applies — 8 or 16 product rows, no idle waves or exactly the upper half */ \
/* of a 16-slot tile without input (key switch) — k_mid128 otherwise                                                            */
#define PZ_MID128_PICK(CT_, NP_, perm_, skipw_, ring_)                                                                     \
if (ring_ && (!(skipw_) || (NP_ >= 16 && npi <= NP_ / 2 && npo > NP_ / 2))) {                                                      \
    if (perm_) PZ_MID128_GOR(CT_, NP_, true) else PZ_MID128_GOR(CT_, NP_, false)                                       \
} else {                                                                                                               \
    if (perm_) { if (skipw_) PZ_MID128_GO(CT_, NP_, true, true) else PZ_MID128_GO(CT_, NP_, true, false) }             \
    else       { if (skipw_) PZ_MID128_GO(CT_, NP_, false, true) else PZ_MID128_GO(CT_, NP_, false, false) }           \
}
#ifdef PZ_EXPERIMENT
    // experiment (POULPY_DBG_MID_CT2=1): the plain 16 x 16 product on 256-thread workgroups - two ciphertexts per tile, two workgroups per
    // CU that are not coupled by barriers (k_mid128r<2,16>; twice the key fetches per ciphertext)
    static const bool mid_ct2 = (exp_knob("POULPY_DBG_MID_CT2", 0) == 1);
    if (mid_ct2 && mid_r && !br && !ds && !perm && npi == 16 && npo == 16 && g.row_max == 16 && g.ncomp == 16) {
        g.n_ct = (batch + 1) / 2;
        const size_t lds = ((size_t)2 * 16 * kMidRS + 384 + 32) * sizeof(cplx);
        const dim3 grid_(std::min({2 * ncu, 512, g.m1 * g.n_ct}));
        PZ_TRY(set_lds((k_mid128r<2, 16, false, 16, false>), lds));
        hipLaunchKernelGGL((k_mid128r<2, 16, false, 16, false>), grid_, dim3(256), lds, M->stream, g);
        dispatch_note(M, "k_mid128r<CT=2,NP=16,NR=16,KR=3> (256 threads, two workgroups per CU)");
        PZ_HIP(hipGetLastError());
        return PZ_OK;
    }
#endif
#define PZ_MID128_LAUNCH(CT_, NP_)                                                                                         \
{                                                                                                                      \
    g.n_ct = (batch + CT_ - 1) / CT_;                                                                                  \
    const size_t lds = ((size_t)CT_ * NP_ * kMidRS + 384 + 32) * sizeof(cplx);   /* tile | wL2 | two twiddle rows | exponents */                                         \
    const dim3 grid_(std::min({ncu, 256, g.m1 * g.n_ct}));   /* <= 256: one scratch tile per workgroup (kMidDummyBytes) */ \
    if (br && NP_ == 8 && (g.br_rm & 1) == 0 && g.ncomp == 6 && npi <= 6 && br_nc3) {   /* six output columns in an 8-slot tile: 2 x 3 per thread */ \
        PZ_TRY(set_lds((k_mid128<CT_, NP_, false, false, true, false, (NP_ == 8 ? 2 : 0), (NP_ == 8 ? 3 : 0)>), lds));  \
        hipLaunchKernelGGL((k_mid128<CT_, NP_, false, false, true, false, (NP_ == 8 ? 2 : 0), (NP_ == 8 ? 3 : 0)>), grid_, dim3(512), lds, M->stream, g); \
        dispatch_note(M, "k_mid128<CT=%d,NP=%d,BR=1,BRNEST=2,NCO=3> (%d ciphertexts per key value)", CT_, NP_, CT_);   \
    } else if (br && NP_ < 32 && (g.br_rm & 1) == 0) {   /* an even number of key rows per coefficient: the nested form of the product loop */ \
        PZ_TRY(set_lds((k_mid128<CT_, NP_, false, false, true, false, (NP_ < 32 ? 2 : 0)>), lds));                     \
        hipLaunchKernelGGL((k_mid128<CT_, NP_, false, false, true, false, (NP_ < 32 ? 2 : 0)>), grid_, dim3(512), lds, M->stream, g); \
        dispatch_note(M, "k_mid128<CT=%d,NP=%d,BR=1,BRNEST=2> (%d ciphertexts per key value)", CT_, NP_, CT_);         \
    } else if (br) {                                                                                                   \
        PZ_TRY(set_lds((k_mid128<CT_, NP_, false, false, true>), lds));                                                \
        hipLaunchKernelGGL((k_mid128<CT_, NP_, false, false, true>), grid_, dim3(512), lds, M->stream, g);             \
        dispatch_note(M, "k_mid128<CT=%d,NP=%d,BR=1> (%d ciphertexts per key value)", CT_, NP_, CT_);                  \
    } else if (ds) {                                                                                                          \
        /* 16-slot tile with 16 terms on 16 inputs (external product) or 8 terms on <= 8 inputs (key switch): k_mid128r */     \
        bool done_ = false;                                                                                            \
        if constexpr (CT_ == 4 && NP_ == 16) {                                                                         \
            if (mid_r && g.ds_n == 16 && npi == 16) {                                                                  \
                PZ_TRY(set_lds((k_mid128r<4, 16, false, 16, false, PZ_MIDR_KR, true>), lds));                          \
                hipLaunchKernelGGL((k_mid128r<4, 16, false, 16, false, PZ_MIDR_KR, true>), grid_, dim3(512), lds, M->stream, g); \
                dispatch_note(M, "k_mid128r<CT=4,NP=16,NR=16,HALFIN=0,KR=%d,DS=1>", PZ_MIDR_KR);                       \
                done_ = true;                                                                                          \
            } else if (mid_r && g.ds_n == 8 && npi <= 8 && npo > 8) {                                                  \
                PZ_TRY(set_lds((k_mid128r<4, 16, false, 8, true, PZ_MIDR_KR, true>), lds));                            \
                hipLaunchKernelGGL((k_mid128r<4, 16, false, 8, true, PZ_MIDR_KR, true>), grid_, dim3(512), lds, M->stream, g); \
                dispatch_note(M, "k_mid128r<CT=4,NP=16,NR=8,HALFIN=1,KR=%d,DS=1>", PZ_MIDR_KR);                        \
                done_ = true;                                                                                          \
            }                                                                                                          \
        }                                                                                                              \
        if (!done_) {                                                                                                  \
            PZ_TRY(set_lds((k_mid128<CT_, NP_, false, true>), lds));                                                   \
            hipLaunchKernelGGL((k_mid128<CT_, NP_, false, true>), grid_, dim3(512), lds, M->stream, g);                \
            dispatch_note(M, "k_mid128<CT=%d,NP=%d,DS=1>", CT_, NP_);                                                  \
        }                                                                                                              \
    } else {                                                                                                           \
        const bool skipw_ = NP_ > 8 && (npi <= NP_ - 8 || npo <= NP_ - 8);   /* shapes with idle waves */              \
        const bool ring_ = mid_r && (g.row_max == NP_ || (NP_ >= 16 && g.row_max == NP_ / 2));   /* product rows k_mid128r is built for */                      \
        PZ_MID128_PICK(CT_, NP_, perm, skipw_, ring_)                                                                  \
    }                                                                                                                  \
}
    // 16 polynomials in, 32 = 16 limbs x 2 columns out, 16 key rows of 32 columns (rank-1 key switch / automorphism / relinearization at 16 limbs:
    // BASELINE configs[4]): 4 ciphertexts per tile and the two output columns as two passes over the same inputs - every key value serves 4
    // ciphertexts instead of the 32-slot tile's 2 (k_mid128r<.., C2>; POULPY_DBG_MID_C2=0 in an experiment build: the 32-slot tile)
    static const bool c2_on = (exp_knob("POULPY_DBG_MID_C2", 1) != 0);
    if (c2_on && mid_r && !br && !ds && npi == 16 && npo == 32 && g.row_max == 16 && g.ncols == 32 && g.ncomp == 32) {
        g.n_ct = (batch + 3) / 4;
        const size_t lds = ((size_t)4 * 16 * kMidRS + 384 + 32) * sizeof(cplx);
        const dim3 grid_(std::min({ncu, 256, g.m1 * g.n_ct}));
        if (perm) {
            PZ_TRY(set_lds((k_mid128r<4, 16, true, 16, false, PZ_MIDR_KR, false, true>), lds));
            hipLaunchKernelGGL((k_mid128r<4, 16, true, 16, false, PZ_MIDR_KR, false, true>), grid_, dim3(512), lds, M->stream, g);
        } else {
            PZ_TRY(set_lds((k_mid128r<4, 16, false, 16, false, PZ_MIDR_KR, false, true>), lds));
            hipLaunchKernelGGL((k_mid128r<4, 16, false, 16, false, PZ_MIDR_KR, false, true>), grid_, dim3(512), lds, M->stream, g);
        }
        dispatch_note(M, "k_mid128r<CT=4,NP=16,PERM=%d,NR=16,HALFIN=0,KR=%d,C2=1> (two output-column passes per tile of 4 ciphertexts)", (int)perm, PZ_MIDR_KR);
        PZ_HIP(hipGetLastError());
        return PZ_OK;
    }
    if (npi <= 8 && npo <= 8) {
        // <= 8 polynomials in and out (e.g. rank 1 with 4 limbs, BASELINE configs[1]): 8 ciphertexts x 8 slots per tile
        PZ_MID128_LAUNCH(8, 8)
    } else if (npi > 16 || npo > 16) {
        // 17..32 polynomials in or out (rank 2-3 with 8 limbs, rank 1 with 16 limbs): 2 ciphertexts x 32 slots per tile
        PZ_MID128_LAUNCH(2, 32)
    } else {
        PZ_MID128_LAUNCH(4, 16)
    }
#undef PZ_MID128_LAUNCH
#undef PZ_MID128_PICK
#undef PZ_MID128_GO
#undef PZ_MID128_GOR
#undef PZ_MID128_GOR1
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}
int launch_mid(pz_module* M, int batch, const cplx* T, cplx* T2, const cplx* Pp, int npi, int npo, int nrows, int ncols, cplx* dummy,
               unsigned perm_mul, unsigned perm_add, const MidDigits* dg, const MidBr* br, bool perm_conj) {
    MidArgs g;
    g.br_lwe = nullptr; g.br_lwe_bs = 0; g.br_i0 = 0; g.br_blk = 0; g.br_rm = 0; g.w2n = M->w2n;
    if (br) {
        // nrows = the key rows of ONE GGSW (= npi here); Pp holds br->blk of them per frequency row
        g.br_lwe = br->lwe; g.br_lwe_bs = br->lwe_bs; g.br_i0 = br->i0; g.br_blk = br->blk; g.br_rm = nrows;
        nrows *= br->blk;
    }
    g.ds_n = 0;
    const bool ds = dg != nullptr && dg->n > 0;
    if (ds) {
        g.ds_n = dg->n;
        for (int t = 0; t < dg->n; ++t) { g.ds_in[t] = dg->in[t]; g.ds_row[t] = dg->row[t]; g.ds_coff[t] = dg->coff[t]; g.ds_cb[t] = dg->cb[t]; }
    }
    g.perm_mul = perm_mul; g.perm_add = perm_add; g.perm_ysign = perm_conj ? -1.0 : 1.0; g.log_m1 = 0;
    static const int mid_skip = exp_knob("POULPY_DBG_MID_SKIP", 0);
    g.dbg = mid_skip;
    while ((1 << g.log_m1) < M->plan.m1) ++g.log_m1;
    const bool perm = perm_mul != 0;
    g.T = T; g.T2 = T2; g.P = Pp; g.npi = npi; g.npo = npo; g.nrows = nrows; g.ncols = ncols;
    g.row_max = br ? nrows : std::min(nrows, npi);
    g.ncomp = std::min(npo, ncols);
    if ((ds ? g.ds_n : g.row_max) < 1)   // k_mid128 assumes at least one product term
        return fail(PZ_ERR_INVALID, "launch_mid: no product term (rows %d, npi %d, ds_n %d)", nrows, npi, g.ds_n);
    g.batch = batch; g.m1 = M->plan.m1; g.n_ct = 0;
    g.wL2 = M->wL2; g.tw12t = M->tw12t; g.dummy = dummy;
    static const int groups = exp_knob("POULPY_DBG_MID_GROUPS", 1);
    g.groups = groups;
    // phase stagger: workgroup w starts (w mod 4) x ~3.4 us late so that the HBM-heavy row passes of some CUs overlap
    // the L2-heavy product phases of others (measured: middle kernel -3 %); off for the m2 = 128 form, where it did not pay
    static const int stg = exp_knob("POULPY_DBG_MID_STAGGER", -1);
    static const int stm = exp_knob("POULPY_DBG_MID_STAGGER_MOD", 4);
    g.stagger = stg >= 0 ? stg : (M->plan.m2 == 128 ? 0 : 1);
    g.stagger_mod = M->plan.m2 == 128 ? (exp_knob("POULPY_DBG_MID_STAGGER_MOD", -1) >= 0 ? stm : 0) : std::max(1, stm);   // m2 = 128: mode bits of k_mid128r's experiments
    if (M->plan.m2 == 128) return launch_mid128(M, g, batch, npi, npo, perm, ds, br != nullptr);
    static const int ct = exp_knob("POULPY_DBG_MID_CT", 2);  // diagnostic knob
    if (ct == 1) return launch_mid_ct<1>(M, g, batch);
    return launch_mid_ct<2>(M, g, batch);
}


}  // namespace pz
