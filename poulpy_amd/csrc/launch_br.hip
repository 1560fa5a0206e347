// launch_br.hip — blind-rotation kernels: the one-kernel rotation (device_br.hpp), the fused block step and the
// accumulation kernels of the composed path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "device_br.hpp"
#include "device_br_ops.hpp"
#include "br_forms.hpp"
#ifndef PZ_BRL_94
#define PZ_BRL_94 1   // rank 2 (9 rows, 12 columns): three column groups of 4 instead of four of 3 - 21 stages instead of 28, 5.36 -> 5.22 ms per 82 block steps (profiles/r05_ab_brl94.txt)
#endif

namespace pz {

int br_try_fused(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                 const pz_blind_rotation_params* p, size_t batch, bool* launched_out) {
    *launched_out = false;
    // whole rotation in one kernel, accumulators resident in LDS (device_br.hpp), when the shape fits and has an instantiation
    // (br_forms.hpp); everything else runs the composed path
    BrFusedPlan pl;
    if (!br_fused_plan(M, p, batch, &pl)) return PZ_OK;
    BrFusedArgs g;
    g.res = (long long*)res; g.lut = (const long long*)lut; g.lwe = (const long long*)lwe_2n; g.brk = (const cplx*)brk;
    g.w2n = M->w2n; g.key_stride = (long long)(pl.pmat_doubles / 2);
    g.n_lwe = (int)p->n_lwe; g.blk = (int)p->block_size; g.cols = (int)p->rank + 1; g.rsz = (int)p->res_size; g.dnum = (int)p->dnum;
    g.bsz = (int)p->brk_size; g.lut_size = (int)p->lut_size; g.base2k = (int)p->base2k; g.m = (int)M->m; g.batch = (int)batch;
    g.dbg_skip = exp_knob("POULPY_DBG_BR_SKIP", 0); g.margin = M->probe ? M->margin : nullptr;
    KTimer kt(M, PZ_K_FUSED_MID);
    // the rounding-margin instantiation of the same form while the module's probe is on (launch_br_probe.hip)
    PZ_TRY(M->probe ? br_fused_launch_probe(M, g, pl) : br_fused_launch<false>(M, g, pl));
    PZ_HIP(hipGetLastError());
    *launched_out = true;
    return PZ_OK;
}

int br_block_step(pz_module* M, const double* acc_dft, long long a_bs, double* acc_add, long long o_bs, const double* brk,
                  size_t pmat_doubles, int row_max, int ncols, int B, int i0, int blk, const int64_t* lwe_2n, long long lwe_bs,
                  bool* launched_out) {
    *launched_out = false;
    if (!(row_max <= 12 && blk <= 64)) return PZ_OK;
                // zero + block_size x (vmp, svp, add, sub) in one kernel, nothing but acc_add written (:321-337)
                BrBlockArgs g;
                g.acc_dft = (const cplx*)acc_dft; g.acc_add = (cplx*)acc_add; g.a_bs = a_bs / 2; g.o_bs = o_bs / 2;
                g.brk = (const cplx*)brk; g.key_stride = (long long)(pmat_doubles / 2);
                g.row_max = row_max; g.ncols = ncols; g.m = (int)M->m; g.batch = B; g.i0 = i0; g.blk = blk;
                g.lwe = (const long long*)lwe_2n; g.lwe_bs = lwe_bs; g.w2n = M->w2n;
                static const int brl_dbg = exp_knob("POULPY_DBG_BRL", 0);
                g.dbg = brl_dbg; g.gx = g.gy = g.gz = 1; g.xcd = 0; g.allcg = 0;
                constexpr int CT = 2;
                KTimer kt(M, PZ_K_VMP);
                const int nc = ncols;
                const unsigned gx = (unsigned)((B + CT - 1) / CT), gy = (unsigned)((M->m + 255) / 256);
                // (input polynomials kept in registers, output columns per workgroup): rank 2 with 3-4 decomposition rows (the
                // circuit-bootstrapping shape) has 9 or 12 inputs, so fewer columns per workgroup there
                int mr, cgs;
                if (row_max > 9 || (row_max > 8 && nc % 3 != 0)) { mr = 12; cgs = 2; }
                else if (row_max > 8) { mr = 9; cgs = (nc % 4 == 0 && PZ_BRL_94) ? 4 : 3; }
                else if (nc % 3 == 0 && nc % 4 != 0) { mr = row_max <= 6 ? 6 : 8; cgs = 3; }   // 3, 6 columns: groups of 3
                else { mr = row_max <= 4 ? 4 : 8; cgs = 4; }
                const int ngroups = (nc + cgs - 1) / cgs;
                // keys staged in LDS once per 4 waves x 2 ciphertexts (k_br_block_lds): POULPY_DBG_BR_LDS = 0 never, 1 only for
                // more than 8 inputs, 2 always
                // (four ciphertexts per wave for <= 6 inputs - every staged key value serving 16 ciphertexts instead of 8, at 256 registers -
                //  measured slower at N = 2048: 11.06 vs 9.68 ms per 82 block steps)
                static const int br_lds = exp_knob("POULPY_DBG_BR_LDS", 2);
                const bool use_lds = M->m % 64 == 0 && (br_lds >= 2 || (br_lds == 1 && row_max > 8));
                bool launched = false;
                if (use_lds) {
                    g.gx = (B + 7) / 8; g.gy = (int)(M->m / 64); g.gz = ngroups;
                    static const int br_xcd = exp_knob("POULPY_DBG_BR_XCD", 1);
                    g.xcd = (br_xcd && (g.gx * g.gy) % 8 == 0) ? 1 : 0;
                    static const int br_allcg = exp_knob("POULPY_DBG_BR_ALLCG", 1);
                    g.allcg = br_allcg ? 1 : 0;
                    const unsigned total = (unsigned)(g.gx * g.gy * (g.allcg ? 1 : g.gz));
#define PZ_BRB(MR_, CG_)                                                                                           \
    if (!launched && mr == MR_ && cgs == CG_) {                                                                    \
        hipLaunchKernelGGL((k_br_block_lds<2, MR_, CG_>), dim3(total), dim3(256), 0, M->stream, g);                \
        dispatch_note(M, "k_br_block_lds<2,MR=%d,CG=%d> (8 ciphertexts per staged key value)", MR_, CG_);          \
        launched = true;                                                                                           \
    }
                    PZ_BRB(12, 2) PZ_BRB(9, 3) PZ_BRB(9, 4) PZ_BRB(6, 3) PZ_BRB(8, 3) PZ_BRB(4, 4) PZ_BRB(8, 4)
#undef PZ_BRB
                } else {
#define PZ_BRB(MR_, CG_)                                                                                           \
    if (!launched && mr == MR_ && cgs == CG_) {                                                                    \
        hipLaunchKernelGGL((k_br_block<CT, MR_, CG_>), dim3(gx, gy, (unsigned)ngroups), dim3(256), 0, M->stream, g); \
        dispatch_note(M, "k_br_block<2,MR=%d,CG=%d>", MR_, CG_);                                                    \
        launched = true;                                                                                           \
    }
                    PZ_BRB(12, 2) PZ_BRB(9, 3) PZ_BRB(6, 3) PZ_BRB(8, 3) PZ_BRB(4, 4) PZ_BRB(8, 4)
#undef PZ_BRB
                }
                PZ_HIP(hipGetLastError());
    *launched_out = true;
    return PZ_OK;
}

int launch_xai_acc(pz_module* M, double* acc_add, long long acc_bs, const double* v, long long v_bs, int polys, int B,
                   const int64_t* lwe_2n, long long lwe_bs, int idx) {
    XaiArgs g;
    g.acc = (cplx*)acc_add; g.v = (const cplx*)v; g.acc_bs = acc_bs / 2; g.v_bs = v_bs / 2;
    g.polys = polys; g.m = (int)M->m; g.batch = B;
    g.lwe = (const long long*)lwe_2n; g.lwe_bs = lwe_bs; g.idx = idx; g.w2n = M->w2n;
    const long long total = (long long)B * g.polys * g.m;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_xai_acc, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 16)), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

int launch_xai_ext(pz_module* M, double* acc_add, const double* v, int polys, int log_ext, int B, const int64_t* lwe_2n, long long lwe_bs,
                   int idx) {
    XaiExtArgs g;
    g.acc = (cplx*)acc_add; g.v = (const cplx*)v; g.polys = polys; g.m = (int)M->m; g.log_ext = log_ext; g.batch = B;
    g.lwe = (const long long*)lwe_2n; g.lwe_bs = lwe_bs; g.idx = idx; g.w2n = M->w2n;
    const long long total = ((long long)B << log_ext) * g.polys * g.m;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_xai_ext, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 16)), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

int launch_br_ext_init(pz_module* M, int64_t* acc, const int64_t* lut, const int64_t* lwe_2n, long long lwe_bs, int log_ext, int cols,
                       int rsz, int lut_size, int B) {
    BrExtInitArgs g;
    g.acc = (long long*)acc; g.lut = (const long long*)lut; g.lwe = (const long long*)lwe_2n; g.lwe_bs = lwe_bs;
    g.n = (int)M->n; g.log_ext = log_ext; g.cols = cols; g.rsz = rsz; g.lut_size = lut_size;
    g.nl = std::min(rsz, lut_size); g.batch = B;
    const int BE = B << log_ext;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_br_ext_init, dim3((unsigned)((M->n / 2 + 255) / 256), (unsigned)g.nl, (unsigned)BE), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

}  // namespace pz
