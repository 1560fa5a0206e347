// launch_fft.hip — dispatch of the four FFT passes (device_fft.hpp) on the module plan.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "internal.hpp"

namespace pz {

#define PZ_P1_CASES(X) X(2, 1, 2) X(2, 1, 4) X(4, 1, 4) X(8, 1, 4) X(8, 1, 16) X(16, 1, 16) X(8, 4, 16) X(8, 8, 16) X(16, 8, 16) X(16, 16, 16)
#define PZ_P1F_CASES(X) X(2, 1, 2) X(2, 1, 4) X(4, 1, 4) X(8, 1, 4) X(8, 1, 16) X(16, 1, 16) X(4, 4, 16) X(4, 8, 16) X(8, 8, 16) X(8, 16, 16) X(16, 16, 16) X(8, 16, 8)
#define PZ_P2_CASES(X) X(2, 1, 2) X(4, 1, 2) X(4, 1, 4) X(8, 1, 4) X(16, 1, 4) X(16, 1, 16) X(8, 4, 16) X(8, 8, 16) X(16, 8, 16) X(16, 16, 16)


int launch_fwd_pass1(pz_module* M, int npolys, const long long* src, PolyMap smap, cplx* T, bool rowmajor, long long mask, bool src32) {
    const FftPlan& pl = M->plan;
    if (npolys == 0) return PZ_OK;
    KTimer kt(M, PZ_K_FWD_PASS1);
    // (Round 2: 32-column blocks for the row-major output at m1 = 256 — 256-byte read runs / 512-byte write runs, the shape that moves
    //  a pure copy from 5.3 to 6.0 TB/s, profiles/r02_hbm_pass_pattern.txt — were measured SLOWER, 3.55 vs 3.32 ms per 1024 ciphertexts:
    //  the 147 KiB exchange buffer leaves one 512-thread workgroup per CU, whose load / exchange / store phases no longer overlap with
    //  a second workgroup's.  The kernel sits at the ceiling of its 16-column access shape: 5.3 TB/s.)
    const int ncb = pl.m2 / pl.cb;
    const int blocks = npolys * ncb;
    // row-major (pipeline) launches use the XCD-aware block order of k_fwd_pass1: grid padded to whole groups of 8 polynomials
    static const int xcd_order = exp_knob("POULPY_DBG_XCD_ORDER", 1);
    const int npx = (rowmajor && xcd_order) ? npolys : 0;
    const int blocks_rm = npx ? ((npolys + 7) / 8) * 8 * ncb : blocks;
    if (src32) {   // 32-bit source digits: the row-major form of the 128-point-row plans (the blind rotation's pipeline path)
        if (!(rowmajor && mask == -1)) return fail(PZ_ERR_INVALID, "forward pass 1: 32-bit source digits need the row-major form");
#define X(A, B, C)                                                                                              \
    if (pl.f1a == A && pl.f1b == B && pl.cb == C) {                                                             \
        const size_t lds = ((size_t)(A + 1) * C * B + 2 * A * B) * sizeof(cplx);                                \
        PZ_TRY(set_lds((k_fwd_pass1<A, B, C, true, true>), lds));                                               \
        hipLaunchKernelGGL((k_fwd_pass1<A, B, C, true, true>), dim3(blocks_rm), dim3((A > B ? A : B) * C), lds, M->stream, src, smap, \
                           T, pl.m2, M->tw1, M->wL1, M->tw12t, mask, npx);                                      \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
        X(4, 4, 16) X(4, 8, 16) X(8, 8, 16) X(8, 16, 16) X(16, 16, 16)
#undef X
        return fail(PZ_ERR_UNSUPPORTED, "forward pass 1: no 32-bit-source form for this plan");
    }
#define X(A, B, C)                                                                                              \
    if (pl.f1a == A && pl.f1b == B && pl.cb == C) {                                                             \
        const size_t lds = ((size_t)(A + 1) * C * B + 2 * A * B) * sizeof(cplx);                                              \
        if (rowmajor) {                                                                                         \
            PZ_TRY(set_lds(k_fwd_pass1<A, B, C, true>, lds));                                                   \
            hipLaunchKernelGGL((k_fwd_pass1<A, B, C, true>), dim3(blocks_rm), dim3((A > B ? A : B) * C), lds, M->stream, src, smap, \
                               T, pl.m2, M->tw1, M->wL1, M->tw12t, mask, npx);                                        \
        } else {                                                                                                \
            PZ_TRY(set_lds(k_fwd_pass1<A, B, C>, lds));                                                         \
            hipLaunchKernelGGL((k_fwd_pass1<A, B, C>), dim3(blocks), dim3((A > B ? A : B) * C), lds, M->stream, src, smap, T, \
                               pl.m2, M->tw1, M->wL1, M->tw12, mask, 0);                                              \
        }                                                                                                       \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
    PZ_P1F_CASES(X)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "no forward pass-1 kernel for m1=%d", pl.m1);
}

int launch_fwd_pass1_w16(pz_module* M, int npolys, const long long* src, PolyMap smap, cplx* T, short* w16) {
    const FftPlan& pl = M->plan;
    if (npolys == 0) return PZ_OK;
    KTimer kt(M, PZ_K_FWD_PASS1);
    const int ncb = pl.m2 / pl.cb;
    const int blocks_rm = ((npolys + 7) / 8) * 8 * ncb;
#define X(A, B, C)                                                                                              \
    if (pl.f1a == A && pl.f1b == B && pl.cb == C) {                                                             \
        const size_t lds = ((size_t)(A + 1) * C * B + 2 * A * B) * sizeof(cplx);                                \
        PZ_TRY(set_lds((k_fwd_pass1_w16<A, B, C>), lds));                                                       \
        hipLaunchKernelGGL((k_fwd_pass1_w16<A, B, C>), dim3(blocks_rm), dim3((A > B ? A : B) * C), lds, M->stream, src, smap, T, pl.m2, M->tw1, M->wL1, \
                           M->tw12t, npolys, w16, M->wide16());                                                 \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
    X(4, 8, 16) X(8, 8, 16) X(8, 16, 16) X(16, 16, 16)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "forward pass 1: no side-copy form for this plan");
}

int launch_fwd_pass1_t16(pz_module* M, int npolys, const short* src, PolyMap smap, cplx* T) {
    const FftPlan& pl = M->plan;
    if (npolys == 0) return PZ_OK;
    KTimer kt(M, PZ_K_FWD_PASS1);
    const int ncb = pl.m2 / pl.cb;
    const int blocks_rm = ((npolys + 7) / 8) * 8 * ncb;   // (XCD-aware block order, as the row-major form of launch_fwd_pass1)
#define X(A, B, C)                                                                                              \
    if (pl.f1a == A && pl.f1b == B && pl.cb == C) {                                                             \
        const size_t lds = ((size_t)(A + 1) * C * B + 2 * A * B) * sizeof(cplx);                                \
        PZ_TRY(set_lds((k_fwd_pass1_t16<A, B, C>), lds));                                                       \
        hipLaunchKernelGGL((k_fwd_pass1_t16<A, B, C>), dim3(blocks_rm), dim3((A > B ? A : B) * C), lds, M->stream,  \
                           reinterpret_cast<const long long*>(src), smap, T, pl.m2, M->tw1, M->wL1, M->tw12t, npolys);  \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
    X(4, 4, 16) X(4, 8, 16) X(8, 8, 16) X(8, 16, 16) X(16, 16, 16)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "forward pass 1: no 16-bit-source form for this plan");
}

int launch_fwd_pass2(pz_module* M, int npolys, const cplx* T, double* dst, PolyMap dmap, const cplx* mul) {
    const FftPlan& pl = M->plan;
    const int blocks = npolys * (pl.m1 / pl.qb);
    if (blocks == 0) return PZ_OK;
    KTimer kt(M, PZ_K_FWD_PASS2);
#define X(A, B, C)                                                                                              \
    if (pl.r2a == A && pl.r2b == B && pl.qb == C) {                                                             \
        const size_t lds = (size_t)A * B * C * sizeof(cplx);                                                    \
        PZ_TRY(set_lds(k_fwd_pass2<A, B, C>, lds));                                                             \
        hipLaunchKernelGGL((k_fwd_pass2<A, B, C>), dim3(blocks), dim3((A > B ? A : B) * C), lds, M->stream, T, dst, dmap, \
                           pl.m1, M->wL2, mul);                                                                 \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
    PZ_P2_CASES(X)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "no forward pass-2 kernel for m2=%d", pl.m2);
}

int launch_inv_pass2(pz_module* M, int npolys, const double* src, PolyMap smap, cplx* T) {
    const FftPlan& pl = M->plan;
    const int blocks = npolys * (pl.m1 / pl.qb);
    if (blocks == 0) return PZ_OK;
    KTimer kt(M, PZ_K_INV_PASS2);
#define X(A, B, C)                                                                                              \
    if (pl.r2a == A && pl.r2b == B && pl.qb == C) {                                                             \
        const size_t lds = (size_t)A * B * C * sizeof(cplx);                                                    \
        PZ_TRY(set_lds(k_inv_pass2<A, B, C>, lds));                                                             \
        hipLaunchKernelGGL((k_inv_pass2<A, B, C>), dim3(blocks), dim3((A > B ? A : B) * C), lds, M->stream, src, smap, T, \
                           pl.m1, M->wL2, M->tw12);                                                             \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
    PZ_P2_CASES(X)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "no inverse pass-2 kernel for m2=%d", pl.m2);
}

int launch_inv_pass1(pz_module* M, int npolys, const cplx* T, long long* dst, PolyMap dmap) {
    const FftPlan& pl = M->plan;
    const int blocks = npolys * (pl.m2 / pl.cb);
    if (blocks == 0) return PZ_OK;
    KTimer kt(M, PZ_K_INV_PASS1);
#define X(A, B, C)                                                                                              \
    if (pl.r1a == A && pl.r1b == B && pl.cb == C) {                                                             \
        const size_t lds = (size_t)(A + 1) * C * B * sizeof(cplx);                                              \
        if (M->probe) {                                                                                         \
            PZ_TRY(set_lds(k_inv_pass1<A, B, C, true>, lds));                                                   \
            hipLaunchKernelGGL((k_inv_pass1<A, B, C, true>), dim3(blocks), dim3((A > B ? A : B) * C), lds, M->stream, T, \
                               dst, dmap, pl.m2, M->tw1inv, M->wL1, M->margin);                                 \
        } else {                                                                                                \
            PZ_TRY(set_lds(k_inv_pass1<A, B, C, false>, lds));                                                  \
            hipLaunchKernelGGL((k_inv_pass1<A, B, C, false>), dim3(blocks), dim3((A > B ? A : B) * C), lds, M->stream, T, \
                               dst, dmap, pl.m2, M->tw1inv, M->wL1, M->margin);                                 \
        }                                                                                                       \
        PZ_HIP(hipGetLastError());                                                                              \
        return PZ_OK;                                                                                           \
    }
    PZ_P1_CASES(X)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "no inverse pass-1 kernel for m1=%d", pl.m1);
}


}  // namespace pz
