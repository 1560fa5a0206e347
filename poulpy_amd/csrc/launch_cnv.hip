// launch_cnv.hip — bivariate-convolution kernels (device_cnv.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "internal.hpp"
#include "device_cnv.hpp"

namespace pz {

int launch_cnv_apply(pz_module* M, int batch, double* res, long long res_bs, int res_cols, int res_col, int min_size, int offset,
                     const double* a, long long a_bs, int a_size, int a_i, int a_j, const double* b, long long b_bs, int b_size, int b_i,
                     int b_j) {
    if (batch <= 0 || min_size <= 0) return PZ_OK;
    CnvArgs g;
    g.res = res; g.a = (const cplx*)a; g.b = (const cplx*)b;
    g.res_bs = res_bs / 2; g.a_bs = a_bs / 2; g.b_bs = b_bs / 2;
    g.res_cols = res_cols; g.res_col = res_col; g.min_size = min_size; g.offset = offset;
    g.a_size = a_size; g.a_i = a_i; g.a_j = a_j; g.b_size = b_size; g.b_i = b_i; g.b_j = b_j;
    g.m = (int)M->m; g.batch = batch;
    KTimer kt(M, PZ_K_VMP);
    // operands staged in LDS, all output limbs per workgroup (POULPY_DBG_CNV_LDS=0: one thread per (point, output limb), operands from L2)
    static const bool cnv_lds = (exp_knob("POULPY_DBG_CNV_LDS", 1) != 0);
    if (cnv_lds && (M->m % 128) == 0 && a_size + b_size <= 64 && a != (const double*)res && b != (const double*)res) {
        const size_t lds = (size_t)(a_size + b_size) * 128 * sizeof(cplx);
        PZ_TRY(set_lds(k_cnv_apply_lds, lds));
        for (int b0 = 0; b0 < batch; b0 += 65535) {   // gridDim.y limit
            CnvArgs gb = g;
            gb.res = res + (long long)b0 * res_bs;
            gb.a = g.a + (long long)b0 * g.a_bs;
            gb.b = g.b + (long long)b0 * g.b_bs;
            const int nb = std::min(65535, batch - b0);
            hipLaunchKernelGGL(k_cnv_apply_lds, dim3((unsigned)(M->m / 128), (unsigned)nb), dim3(256), lds, M->stream, gb);
        }
        PZ_HIP(hipGetLastError());
        return PZ_OK;
    }
    for (int b0 = 0; b0 < batch; b0 += 65535) {   // gridDim.z limit
        CnvArgs gb = g;
        gb.res = res + (long long)b0 * res_bs;
        gb.a = g.a + (long long)b0 * g.a_bs;
        gb.b = g.b + (long long)b0 * g.b_bs;
        const int nb = std::min(65535, batch - b0);
        hipLaunchKernelGGL(k_cnv_apply, dim3((unsigned)((M->m + 255) / 256), (unsigned)min_size, (unsigned)nb), dim3(256), 0, M->stream, gb);
    }
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

// one term of a tensoring on the row-major pipeline layout (device_cnv.hpp, k_mid_cnv); a/b: T' of the operand limbs (see MidCnvArgs)
bool mid_cnv_supported(const pz_module* M, int a_size, int b_size, int min_size) {
    return M->plan.m2 == 128 && (M->plan.m1 % 16) == 0 && tail_supported(M) && tail_rsh_supported(M) &&   // (the tensoring tail exists for the plans of the shifted-store tail)
           a_size >= 1 && b_size >= 1 && min_size >= 1 && min_size <= 32 &&
           ((size_t)std::max(a_size + b_size, min_size) * kMidCnvRS + 256) * sizeof(cplx) <= (size_t)160 * 1024;
}
int launch_mid_cnv(pz_module* M, int batch, const cplx* a_main, const cplx* a_last, const cplx* b_main, const cplx* b_last, cplx* T2, int cols,
                   int a_size, int b_size, int a_i, int a_j, int b_i, int b_j, int min_size, int offset) {
    if (batch <= 0 || min_size <= 0) return PZ_OK;
    MidCnvArgs g;
    g.a_main = a_main; g.a_last = a_last; g.b_main = b_main; g.b_last = b_last; g.T2 = T2; g.cols = cols;
    g.a_size = a_size; g.b_size = b_size; g.a_i = a_i; g.a_j = a_j; g.b_i = b_i; g.b_j = b_j; g.min_size = min_size; g.offset = offset;
    g.m1 = M->plan.m1; g.batch = batch; g.wL2 = M->wL2; g.tw12t = M->tw12t;
    const size_t lds = ((size_t)std::max(a_size + b_size, min_size) * kMidCnvRS + 256) * sizeof(cplx);
    KTimer kt(M, PZ_K_FUSED_MID);
    PZ_TRY(set_lds(k_mid_cnv, lds));
    hipLaunchKernelGGL(k_mid_cnv, dim3((unsigned)((long long)batch * g.m1)), dim3(256), lds, M->stream, g);
    dispatch_note(M, "k_mid_cnv (a %d + b %d limbs -> %d, lds=%zu)", a_size, b_size, min_size, lds);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

// all three terms of a rank-1 tensoring in one launch (device_cnv.hpp, k_mid_cnv3): T2 = [term][pair][limb < min_size][m]
bool mid_cnv3_supported(const pz_module* M, int cols, int a_size, int b_size, int min_size) {
    static const bool on = (rt_knob("POULPY_DBG_TENSOR_ALLTERMS", 1) != 0);
    return on && cols == 2 && mid_cnv_supported(M, a_size, b_size, min_size) && a_size == b_size && (a_size == 16 || a_size == 8) && min_size <= 21 &&
           3 * min_size <= 64;
}
int launch_mid_cnv3(pz_module* M, int batch, const cplx* a_main, const cplx* a_last, const cplx* b_main, const cplx* b_last, cplx* T2, int a_size,
                    int min_size, int offset) {
    if (batch <= 0 || min_size <= 0) return PZ_OK;
    MidCnv3Args g;
    g.a_main = a_main; g.a_last = a_last; g.b_main = b_main; g.b_last = b_last; g.T2 = T2;
    g.min_size = min_size; g.offset = offset; g.m1 = M->plan.m1; g.batch = batch; g.wL2 = M->wL2; g.tw12t = M->tw12t;
    const size_t lds = ((size_t)64 * kMidCnvRS + 384) * sizeof(cplx);   // tile | wL2 | two twiddle rows
    KTimer kt(M, PZ_K_FUSED_MID);
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, M->device);
    if (M->cu_count > 0) ncu = M->cu_count;
    const dim3 grid((unsigned)std::min<long long>((long long)batch * g.m1, ncu));   // persistent: one workgroup per CU
    // b = a (glwe_tensor_square_apply): the square form - half the operand rows, symmetric limb products
    const bool sq = a_main == b_main && a_last == b_last;
    // the static window: multiply-adds with i + j < WLO are not compiled in; WLO = a_size - 4 when the offset allows it (CKKS keeps the top limbs
    // of the product: offset ~ a_size), else 0
    const int wlo = offset >= a_size - 4 ? a_size - 4 : 0;
#define PZ_CNV3_LAUNCH(K_) { PZ_TRY(set_lds((K_), lds)); hipLaunchKernelGGL((K_), grid, dim3(512), lds, M->stream, g); }
#define PZ_CNV3_FORMS(AS_)                                                                                                  \
    {                                                                                                                       \
        if (sq && wlo) PZ_CNV3_LAUNCH((k_mid_cnv3<AS_, AS_, true, AS_ - 4>))                                                \
        else if (sq) PZ_CNV3_LAUNCH((k_mid_cnv3<AS_, AS_, true, 0>))                                                        \
        else if (wlo) PZ_CNV3_LAUNCH((k_mid_cnv3<AS_, AS_, false, AS_ - 4>))                                                \
        else PZ_CNV3_LAUNCH((k_mid_cnv3<AS_, AS_, false, 0>))                                                               \
    }
    if (a_size == 16) PZ_CNV3_FORMS(16) else PZ_CNV3_FORMS(8)
#undef PZ_CNV3_FORMS
#undef PZ_CNV3_LAUNCH
    dispatch_note(M, "k_mid_cnv3<%d,%d> SQ=%d WLO=%d (3 terms, %d limbs each, product limbs [%d, %d))", a_size, a_size, (int)sq, wlo, min_size, offset, offset + min_size);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

int launch_cnv_by_const(pz_module* M, long long* res, int res_cols, int res_col, int min_size, int offset, const long long* a, int a_cols,
                        int a_size, int a_col, const long long* bconst, int b_size) {
    if (min_size <= 0) return PZ_OK;
    CnvConstArgs g;
    g.res = res; g.a = a; g.b = bconst;
    g.res_cols = res_cols; g.res_col = res_col; g.a_cols = a_cols; g.a_col = a_col; g.a_size = a_size; g.b_size = b_size;
    g.min_size = min_size; g.offset = offset; g.n = (int)M->n;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_cnv_by_const, dim3((unsigned)((M->n + 255) / 256), (unsigned)min_size), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

}  // namespace pz
