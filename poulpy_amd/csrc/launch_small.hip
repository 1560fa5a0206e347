// launch_small.hip — the two-kernel GLWE product pipeline for N = 4096 (device_small.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "internal.hpp"
#include "device_small.hpp"

namespace pz {

bool small_supported(const pz_module* M, int npi, int npo_limbs) {
    return M->plan.m1 == kSmallM1 && M->plan.m2 == kSmallM2 && npi >= 1 && npo_limbs >= 1 && npo_limbs <= 4;
}

int launch_small_fwd(pz_module* M, int npolys, const long long* src, PolyMap smap, cplx* S) {
    if (npolys <= 0) return PZ_OK;
    SmallFwdArgs g;
    g.src = src; g.smap = smap; g.S = S; g.npolys = npolys; g.tw1 = M->tw1; g.tw12t = M->tw12t; g.wL2 = M->wL2;
    const size_t lds = ((size_t)2 * 16 * kSmallRS + kSmallM2) * sizeof(cplx);
    KTimer kt(M, PZ_K_FWD_PASS1);
    PZ_TRY(set_lds(k_small_fwd, lds));
    hipLaunchKernelGGL(k_small_fwd, dim3((npolys + 1) / 2), dim3(256), lds, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

int launch_small_inv(pz_module* M, int batch, const cplx* S, const cplx* Pp, int npi, int nrows, int ncols, int cols_out, int ksz,
                     long long* res, long long res_bs, int res_cols, int res_size, const long long* small, long long small_bs,
                     int small_cols, int small_size, int base2k, int body_col) {
    if (batch <= 0) return PZ_OK;
    SmallInvArgs g;
    g.S = S; g.Pp = Pp; g.res = res; g.small = small; g.res_bs = res_bs; g.small_bs = small_bs;
    g.batch = batch; g.npi = npi; g.nrows = nrows; g.ncols = ncols; g.cols_out = cols_out; g.ksz = ksz;
    g.res_cols = res_cols; g.res_size = res_size; g.small_cols = small_cols; g.small_size = small_size; g.base2k = base2k; g.body_col = body_col;
    g.tw12t = M->tw12t; g.wL2 = M->wL2; g.tw1inv = M->tw1inv;
    static const int skip = getenv("POULPY_DBG_SMALL_SKIP") ? atoi(getenv("POULPY_DBG_SMALL_SKIP")) : 0;
    g.dbg = skip;
    const size_t lds = ((size_t)ksz * 16 * kSmallRS + kSmallM2) * sizeof(cplx);
    // workgroup id -> (xcd = id & 7, slot = id >> 3): ciphertext (slot / cols_out) * 8 + xcd, column slot % cols_out
    const int grid = ((batch + 7) / 8) * 8 * cols_out;
    KTimer kt(M, PZ_K_FUSED_TAIL);
#define X(KS_)                                                                                      \
    if (ksz == KS_) {                                                                               \
        PZ_TRY(set_lds(k_small_inv<KS_>, lds));                                                     \
        hipLaunchKernelGGL(k_small_inv<KS_>, dim3(grid), dim3(1024), lds, M->stream, g);             \
        PZ_HIP(hipGetLastError());                                                                  \
        return PZ_OK;                                                                               \
    }
    X(1) X(2) X(3) X(4)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "small-ring pipeline: %d key limbs", ksz);
}

}  // namespace pz
