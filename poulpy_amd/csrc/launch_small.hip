// launch_small.hip — the two-kernel GLWE product pipeline for N = 1024 / 2048 / 4096 (device_small.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "device_small.hpp"
#include "device_small_one.hpp"

namespace pz {

static inline int small_m1(const pz_module* M) { return (int)(M->m / kSmallM2); }

bool small_supported(const pz_module* M, int npi, int key_limbs) {
    const int m1 = small_m1(M);
    return (M->m % kSmallM2) == 0 && (m1 == 4 || m1 == 8 || m1 == 16) && npi >= 1 && key_limbs >= 1 && key_limbs <= 4;
}

// The small pipeline splits m = M1 x 128 whatever the module's own plan is (N = 1024 / 2048 use 16 x 32 / 32 x 32 for the per-op kernels):
// its four tables are built on first use — the module's own where the plan already is M1 x 128 (N = 4096).
static int ensure_small_tables(pz_module* M) {
    if (M->s_tw1) return PZ_OK;
    const int m1 = small_m1(M);
    if (M->plan.m1 == m1 && M->plan.m2 == kSmallM2) {
        M->s_tw1 = M->tw1; M->s_tw1inv = M->tw1inv; M->s_tw12t = M->tw12t; M->s_wL2 = M->wL2;
        M->s_owned = false;
        return PZ_OK;
    }
    const long long m = (long long)M->m;
    std::vector<cplx> h((size_t)m1);
    double c, s;
    for (int j1 = 0; j1 < m1; ++j1) { root_of_unity(j1, 4ll * m1, c, s); h[(size_t)j1] = make_double2(c, s); }
    PZ_TRY(upload_table(&M->s_tw1, h));
    const double inv_m = 1.0 / (double)m;
    for (int j1 = 0; j1 < m1; ++j1) { h[(size_t)j1].x = h[(size_t)j1].x * inv_m; h[(size_t)j1].y = -h[(size_t)j1].y * inv_m; }
    PZ_TRY(upload_table(&M->s_tw1inv, h));
    h.resize(kSmallM2);
    for (int t = 0; t < kSmallM2; ++t) { root_of_unity(t, kSmallM2, c, s); h[(size_t)t] = make_double2(c, s); }
    PZ_TRY(upload_table(&M->s_wL2, h));
    h.resize((size_t)m);
    for (long long q1 = 0; q1 < m1; ++q1)
        for (long long j2 = 0; j2 < kSmallM2; ++j2) {
            root_of_unity(j2 * (4 * q1 + 1), 4 * m, c, s);
            h[(size_t)(q1 * kSmallM2 + j2)] = make_double2(c, s);
        }
    PZ_TRY(upload_table(&M->s_tw12t, h));
    M->s_owned = true;
    return PZ_OK;
}

int launch_small_permute(pz_module* M, const double* P, cplx* Pp, int npolys) {
    const long long total = (long long)M->m * npolys;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    const dim3 grid((unsigned)((total + 255) / 256));
    switch (small_m1(M)) {
        case 4: hipLaunchKernelGGL(k_small_permute<4>, grid, dim3(256), 0, M->stream, reinterpret_cast<const cplx*>(P), Pp, npolys); break;
        case 8: hipLaunchKernelGGL(k_small_permute<8>, grid, dim3(256), 0, M->stream, reinterpret_cast<const cplx*>(P), Pp, npolys); break;
        default: hipLaunchKernelGGL(k_small_permute<16>, grid, dim3(256), 0, M->stream, reinterpret_cast<const cplx*>(P), Pp, npolys); break;
    }
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

// N = 1024 / 2048 / 4096: the per-op transforms in one kernel each (whole polynomial in LDS)
bool small_transform_supported(const pz_module* M) {
    const int m1 = small_m1(M);
    static const int on = exp_knob("POULPY_DBG_SMALL_FFT", 1);   // 0: two-pass per-op transforms (A/B)
    return on && (M->m % kSmallM2) == 0 && (m1 == 4 || m1 == 8 || m1 == 16);
}
int launch_small_idft(pz_module* M, int npolys, const double* a, PolyMap smap, long long* res, PolyMap dmap) {
    if (npolys <= 0) return PZ_OK;
    PZ_TRY(ensure_small_tables(M));
    SmallIdftArgs g;
    g.S = (const cplx*)a; g.smap = smap; g.res = res; g.dmap = dmap; g.npolys = npolys;
    g.tw12t = M->s_tw12t; g.wL2 = M->s_wL2; g.tw1inv = M->s_tw1inv; g.margin = M->probe ? M->margin : nullptr;
    const int m1 = small_m1(M);
    const size_t lds = ((size_t)2 * m1 * kSmallIdftRS + kSmallM2 + m1) * sizeof(cplx);
    KTimer kt(M, PZ_K_INV_PASS1);
    const dim3 grid((unsigned)((npolys + 1) / 2));
#define X(M1_)                                                                                      \
    if (m1 == M1_) {                                                                                \
        PZ_TRY(set_lds(k_small_idft<M1_>, lds));                                                    \
        hipLaunchKernelGGL(k_small_idft<M1_>, grid, dim3(256), lds, M->stream, g);                  \
        PZ_HIP(hipGetLastError());                                                                  \
        return PZ_OK;                                                                               \
    }
    X(4) X(8) X(16)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "small-ring transform: unsupported ring degree");
}
int launch_small_fwd(pz_module* M, int npolys, const long long* src, PolyMap smap, cplx* S, bool natural_order, const PolyMap* dmap,
                     const cplx* mul) {
    if (npolys <= 0) return PZ_OK;
    PZ_TRY(ensure_small_tables(M));
    SmallFwdArgs g;
    g.src = src; g.smap = smap; g.S = S; g.npolys = npolys; g.tw1 = M->s_tw1; g.tw12t = M->s_tw12t; g.wL2 = M->s_wL2; g.natural = natural_order ? 1 : 0;
    g.use_dmap = dmap ? 1 : 0; g.dmap = dmap ? *dmap : smap; g.mul = mul;
    if ((dmap || mul) && !natural_order) return fail(PZ_ERR_INVALID, "small-ring forward transform: a destination map / factor needs the standard order");
    const int m1 = small_m1(M);
    const size_t lds = ((size_t)2 * m1 * kSmallRS + kSmallM2) * sizeof(cplx);
    KTimer kt(M, PZ_K_FWD_PASS1);
    const dim3 grid((unsigned)((npolys + 1) / 2));
#define X(M1_)                                                                                      \
    if (m1 == M1_) {                                                                                \
        PZ_TRY(set_lds(k_small_fwd<M1_>, lds));                                                     \
        hipLaunchKernelGGL(k_small_fwd<M1_>, grid, dim3(256), lds, M->stream, g);                   \
        PZ_HIP(hipGetLastError());                                                                  \
        return PZ_OK;                                                                               \
    }
    X(4) X(8) X(16)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "small-ring pipeline: m1 = %d", m1);
}

int launch_small_inv(pz_module* M, int batch, const cplx* S, const cplx* Pp, int npi, int nrows, int ncols, int cols_out, int ksz,
                     long long* res, long long res_bs, int res_cols, int res_size, const long long* small, long long small_bs,
                     int small_cols, int small_size, int base2k, int body_col, bool noprod, cplx* fwd_S, int fwd_limbs,
                     bool au, unsigned au_p, int au_mode, bool post_rsh, int acc32) {
    if (batch <= 0) return PZ_OK;
    PZ_TRY(ensure_small_tables(M));
    SmallInvArgs g;
    g.S = S; g.Pp = Pp; g.res = res; g.small = small; g.res_bs = res_bs; g.small_bs = small_bs;
    g.batch = batch; g.npi = npi; g.nrows = nrows; g.ncols = ncols; g.cols_out = cols_out; g.ksz = ksz;
    g.res_cols = res_cols; g.res_size = res_size; g.small_cols = small_cols; g.small_size = small_size; g.base2k = base2k; g.body_col = body_col;
    g.tw12t = M->s_tw12t; g.wL2 = M->s_wL2; g.tw1inv = M->s_tw1inv;
    static const int skip = exp_knob("POULPY_DBG_SMALL_SKIP", 0);
    g.dbg = skip; g.margin = M->probe ? M->margin : nullptr;
    if (acc32 && !(noprod && base2k <= 31)) return fail(PZ_ERR_INVALID, "small-ring pipeline: 32-bit accumulator digits need the product-free form and base2k <= 31");
    g.acc32 = acc32;
    g.S_out = fwd_S; g.tw1 = M->s_tw1; g.fwd_limbs = fwd_S ? fwd_limbs : 0;
    g.au_p = au_p; g.au_mode = au_mode; g.post_rsh = (post_rsh && au) ? 1 : 0;
    {   // p^-1 mod 2^32 by Newton steps (p odd), reduced mod 2n in the kernel
        unsigned x = au_p | 1u;
        for (int it = 0; it < 5; ++it) x *= 2u - (au_p | 1u) * x;
        g.au_pinv = x;
    }
    if (post_rsh && !(au && au_mode != 0 && base2k <= 29)) return fail(PZ_ERR_INVALID, "small-ring pipeline: shifted stores need an automorphism form with an operand, base2k <= 29");
    if (au && (noprod || fwd_S || small == nullptr || small_cols != cols_out))
        return fail(PZ_ERR_INVALID, "small-ring pipeline: the automorphism variant needs the key-switch operand");
    if (fwd_S && !(noprod && fwd_limbs >= 1 && fwd_limbs <= ksz && fwd_limbs <= res_size && fwd_limbs <= 8))
        return fail(PZ_ERR_INVALID, "small-ring pipeline: forward transform of %d limbs behind the inverse of %d", fwd_limbs, ksz);
    const int m1 = small_m1(M);
    const size_t lds = ((size_t)ksz * m1 * small_inv_rs(m1, noprod) + kSmallM2 + m1) * sizeof(cplx);   // tile + wL2 + tw1inv
    // workgroup id -> (xcd = id & 7, slot = id >> 3): ciphertext (slot / cols_out) * 8 + xcd, column slot % cols_out
    const int grid = ((batch + 7) / 8) * 8 * cols_out;
    KTimer kt(M, PZ_K_FUSED_TAIL);
#define XL(K_)                                                                                                \
    {                                                                                                         \
        PZ_TRY(set_lds((K_), lds));                                                                           \
        hipLaunchKernelGGL((K_), dim3(grid), dim3(64 * m1), lds, M->stream, g);                               \
    }
    // (the product-free form with the chained forward transform - the blind rotation's tail - exists for N <= 2048: at N = 4096 the rotation runs
    //  on the three-kernel pipeline, and these four instantiations carried 28 B of scratch each)
    if (noprod && g.fwd_limbs && m1 == 16) return fail(PZ_ERR_UNSUPPORTED, "small-ring pipeline: no chained forward transform at N = 4096");
#define XFWD_4(KS_) XL((k_small_inv<4, KS_, true, true>))
#define XFWD_8(KS_) XL((k_small_inv<8, KS_, true, true>))
#define XFWD_16(KS_) {}
#define X(M1_, KS_)                                                                                           \
    if (m1 == M1_ && ksz == KS_) {                                                                            \
        if (noprod && g.fwd_limbs) XFWD_##M1_(KS_)                                                            \
        else if (noprod) XL((k_small_inv<M1_, KS_, true>))                                                    \
        else if (au) XL((k_small_inv<M1_, KS_, false, false, true>))                                          \
        else XL((k_small_inv<M1_, KS_>))                                                                      \
        PZ_HIP(hipGetLastError());                                                                            \
        return PZ_OK;                                                                                         \
    }
    X(4, 1) X(4, 2) X(4, 3) X(4, 4) X(8, 1) X(8, 2) X(8, 3) X(8, 4) X(16, 1) X(16, 2) X(16, 3) X(16, 4)
#undef X
#undef XFWD_4
#undef XFWD_8
#undef XFWD_16
#undef XL
    return fail(PZ_ERR_UNSUPPORTED, "small-ring pipeline: m1 = %d, %d key limbs", m1, ksz);
}

// N = 1024 / 2048, plain product / key switch of a rank-1 ciphertext: ONE kernel per call (device_small_one.hpp)
bool small_one_supported(const pz_module* M, int npi, int nrows, int ncols, int cols_out, int ksz, int batch) {
    static const bool on = (exp_knob("POULPY_DBG_SMALL_ONE", 1) != 0);
    const int m1 = small_m1(M);
    // measured (profiles/r06_ab_small_one.txt, 1024 per call): N = 1024 + 7 ... + 20 % on every shape; N = 2048 + 4 % for the external product (8 input
    // polynomials), - 5 ... 8 % for the key switch (4 inputs: the two-kernel pipeline has two workgroups per CU there, this kernel one) - which stays on two kernels
    // below ~3000 ciphertexts per call (at 4096 the one-kernel form is + 7 %: the two-kernel path's spectra no longer sit in the Infinity Cache).  After the
    // both-columns form (tools/dbg/r6_run35.sh): the key switch still - 3 ... 4 % at 4 / 3 limbs, but the 2-limb external product (4 inputs, 2 key limbs) + 15 %:
    // shapes with at most two key limbs take the one-kernel form too
#ifndef PZ_SMALL_ONE_ALL
#define PZ_SMALL_ONE_ALL 0   // A/B builds: 1 = also the N = 2048 shapes with <= 4 input polynomials
#endif
    return on && (M->m % kSmallM2) == 0 && (m1 == 4 || (m1 == 8 && (npi > 4 || batch >= 3072 || (npi == 4 && ksz <= 2) || PZ_SMALL_ONE_ALL))) && cols_out == 2 && ksz >= 1 && ksz <= 4 && ncols == ksz * cols_out && npi >= 1 &&
           npi <= 8 && nrows >= 1;
}
int launch_small_one(pz_module* M, int batch, const long long* src, PolyMap smap, const cplx* Pp, int npi, int nrows, int ncols, int ksz, long long* res,
                     long long res_bs, int res_cols, int res_size, const long long* small, long long small_bs, int small_cols, int small_size, int base2k,
                     int body_col) {
    if (batch <= 0) return PZ_OK;
    PZ_TRY(ensure_small_tables(M));
    SmallOneArgs g;
    g.src = src; g.smap = smap; g.Pp = Pp; g.res = res; g.small = small; g.res_bs = res_bs; g.small_bs = small_bs;
    g.batch = batch; g.npi = npi; g.nrows = nrows; g.ncols = ncols; g.ksz = ksz;
    g.res_cols = res_cols; g.res_size = res_size; g.small_cols = small_cols; g.small_size = small_size; g.base2k = base2k; g.body_col = body_col;
    g.tw1 = M->s_tw1; g.tw12t = M->s_tw12t; g.wL2 = M->s_wL2; g.tw1inv = M->s_tw1inv; g.margin = M->probe ? M->margin : nullptr;
    const int m1 = small_m1(M);
    const size_t lds = ((size_t)8 * m1 * kSmallRS + kSmallM2 + m1) * sizeof(cplx);   // tile of 8 polynomials + wL2 + tw1inv
    KTimer kt(M, PZ_K_FUSED_TAIL);
#define X(M1_, KS_)                                                                                           \
    if (m1 == M1_ && ksz == KS_) {                                                                            \
        PZ_TRY(set_lds((k_small_one<M1_, KS_>), lds));                                                        \
        hipLaunchKernelGGL((k_small_one<M1_, KS_>), dim3(batch), dim3(512), lds, M->stream, g);               \
        dispatch_note(M, "k_small_one<M1=%d,KS=%d> (one kernel per ciphertext, %d input polynomials)", M1_, KS_, npi); \
        PZ_HIP(hipGetLastError());                                                                            \
        return PZ_OK;                                                                                         \
    }
    X(4, 1) X(4, 2) X(4, 3) X(4, 4) X(8, 1) X(8, 2) X(8, 3) X(8, 4)
#undef X
    return fail(PZ_ERR_UNSUPPORTED, "one-kernel product: m1 = %d, %d key limbs", m1, ksz);
}

}  // namespace pz
