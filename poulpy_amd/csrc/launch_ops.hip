// launch_ops.hip — elementwise / permutation / shift / normalize / VMP launches (device_ops.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "device_ops.hpp"

#ifndef PZ_VMP_RB
#define PZ_VMP_RB 2
#endif

namespace pz {

int launch_ew(pz_module* M, int op, void* res, long long res_bs, long long res_ls, const void* a, long long a_bs,
                     long long a_ls, const void* b, long long b_bs, long long b_ls, int nlimbs, int batch) {
    if (nlimbs <= 0 || batch <= 0) return PZ_OK;
    EwArgs g;
    g.res = res; g.a = a; g.b = b;
    g.res_bs = res_bs; g.res_ls = res_ls; g.a_bs = a_bs; g.a_ls = a_ls; g.b_bs = b_bs; g.b_ls = b_ls;
    g.nlimbs = nlimbs; g.n = (int)M->n; g.batch = batch; g.op = op;
    const long long total = (long long)batch * nlimbs * (long long)(M->n / 2);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_ew, dim3(blocks), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}


// dst = +-src(X^p-gather with multiplier mul) [+ add]; see k_automorphism
int launch_automorphism(pz_module* M, int npolys, const long long* src, PolyMap sm, long long* dst, PolyMap dm, unsigned mul,
                               int flags, const long long* add, PolyMap am, short* dst16) {
    if (npolys <= 0) return PZ_OK;
    AutoArgs g;
    g.src = src; g.dst = dst; g.add = add; g.sm = sm; g.dm = dm; g.am = am;
    g.npolys = npolys; g.n = (int)M->n; g.mul = mul; g.flags = flags & 7;
    if (dst16 && (flags & 16)) g.flags |= 16;   // (16-bit output only: `add` is subtracted)
    g.dst16 = dst16; g.wide = M->wide16();
    g.cond_total = 0;
    const bool cond = (flags & 32) != 0;   // the i64 fallback of the 16-bit body pre-pass: only if the flag is up (k_automorphism: cond_total)
    g.t16_m1 = M->plan.f1a * M->plan.f1b; g.t16_cb = M->plan.cb; g.t16_m2sh = 0;
    while ((1 << g.t16_m2sh) < M->plan.m2) ++g.t16_m2sh;
    if (dst16) {   // 16-bit tile-order output (the spectral automorphism forms' body operand; dm addresses limbs of n int16)
        if (M->n < 1024 || (M->plan.m2 % M->plan.cb) != 0 || M->plan.cb % 4 != 0) return fail(PZ_ERR_UNSUPPORTED, "automorphism pre-pass: no 16-bit tile-order output on this plan");
        g.flags |= 8;
        // through LDS (one workgroup per polynomial, every source line read once) where the polynomial fits it as int16 and no second operand is added;
        // POULPY_DBG_AUTO_T16_LDS=0: the gather kernels with a 16-bit store
        static const int lds_knob = exp_knob("POULPY_DBG_AUTO_T16_LDS", 1);
        if (lds_knob && M->n <= 65536 && M->n >= 4096) {
            KTimer kt(M, PZ_K_ELEMENTWISE);
            const size_t lds = (size_t)M->n * sizeof(short);
            PZ_TRY(set_lds(k_automorphism_t16, lds));
            hipLaunchKernelGGL(k_automorphism_t16, dim3(npolys), dim3(1024), lds, M->stream, g);
            PZ_HIP(hipGetLastError());
            return PZ_OK;
        }
        if (g.flags & 16) return fail(PZ_ERR_UNSUPPORTED, "automorphism pre-pass: the subtracting 16-bit form exists through LDS only");
    }
    KTimer kt(M, PZ_K_ELEMENTWISE);
    // Galois elements whose gather has no locality (neither g nor -g small): chunks of 4 outputs per thread, sources read in runs
    // (k_automorphism_chunk; POULPY_DBG_AUTO_CHUNK=0: always the plain gather, =2: always the chunked form)
    static const int chunk_knob = exp_knob("POULPY_DBG_AUTO_CHUNK", 1);
    const unsigned two_n = 2u * (unsigned)M->n, gm = mul & (two_n - 1u);
    unsigned hinv = gm;   // g^-1 mod 2N by Newton's iteration (g odd): x <- x (2 - g x), 3 -> 6 -> 12 -> 24 -> 48 correct bits
    for (int it = 0; it < 5; ++it) hinv *= 2u - gm * hinv;
    hinv &= two_n - 1u;
    // (the plain gather has locality when the multiplier OR its inverse is small in absolute value: sources g apart share lines, or
    //  outputs g^-1 apart do and the workgroup's 512 outputs cover the 16 of a line: measured equal or better up to 25, profiles/r04_ab_auto_chunk.txt)
    const unsigned dist = std::min(std::min(gm, two_n - gm), std::min(hinv, two_n - hinv));
    if (M->n >= 1024 && (chunk_knob == 2 || (chunk_knob == 1 && dist > 32))) {
        const int bpp = (int)(M->n / 1024);
        const int blocks = ((npolys + 7) / 8) * 8 * bpp;
        if (cond) {
            g.cond_total = blocks;
            hipLaunchKernelGGL(k_automorphism_chunk_cond, dim3(std::min(blocks, 4096)), dim3(256), 0, M->stream, g, hinv);
        } else hipLaunchKernelGGL(k_automorphism_chunk, dim3(blocks), dim3(256), 0, M->stream, g, hinv);
        PZ_HIP(hipGetLastError());
        return PZ_OK;
    }
    const int bpp = M->n >= 512 ? (int)(M->n / 512) : 1;
    const int blocks = ((npolys + 7) / 8) * 8 * bpp;
    if (cond) {
        g.cond_total = blocks;
        hipLaunchKernelGGL(k_automorphism_cond, dim3(std::min(blocks, 4096)), dim3(256), 0, M->stream, g);
    } else hipLaunchKernelGGL(k_automorphism, dim3(blocks), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

int launch_rotate(pz_module* M, int npolys, const long long* src, PolyMap sm, long long* dst, PolyMap dm, int mode,
                         int polys_per_batch, const long long* shift, long long shift_bs, long long shift_idx, long long shift_const) {
    if (npolys <= 0) return PZ_OK;
    RotArgs g;
    g.src = src; g.dst = dst; g.sm = sm; g.dm = dm; g.npolys = npolys; g.n = (int)M->n;
    g.polys_per_batch = std::max(polys_per_batch, 1); g.mode = mode;
    g.shift = shift; g.shift_bs = shift_bs; g.shift_idx = shift_idx; g.shift_const = shift_const;
    const int bpp = M->n >= 512 ? (int)(M->n / 512) : 1;
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_rotate, dim3(npolys * bpp), dim3(256), 0, M->stream, g);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}
// Zero-fill as a KERNEL (round 6).  hipMemsetAsync inside a captured call becomes a memset node of the HIP graph; replayed under the HIP runtime that
// PyTorch ships (7.0) such a graph gave non-deterministic results at launch-bound sizes - the blind rotation at N = 2^14 with 2 ciphertexts per call,
// found by tests/test_gpu_structured.py: round-5 library alike, plain launches and the system runtime unaffected (tools/dbg/br_graph_repro.py).
// The composite calls that can run under capture zero through this kernel instead: the graph then holds kernel nodes only.
__global__ void __launch_bounds__(256) k_zero_words(unsigned long long* p, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) p[i] = 0ull;
}
int launch_zero_bytes(pz_module* M, void* ptr, size_t bytes) {
    if (bytes == 0) return PZ_OK;
    if (((uintptr_t)ptr & 7) != 0 || (bytes & 7) != 0) {   // (never the case for the polynomial buffers; kept correct for any caller)
        PZ_HIP(hipMemsetAsync(ptr, 0, bytes, M->stream));
        return PZ_OK;
    }
    const size_t nwords = bytes / 8;
    const unsigned grid = (unsigned)std::min<size_t>((nwords + 255) / 256, 4096);
    KTimer kt(M, PZ_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_zero_words, dim3(grid), dim3(256), 0, M->stream, (unsigned long long*)ptr, nwords);
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

int launch_rsh(pz_module* M, int batch, long long* data, long long bs, int cols, int size, int col0, int ncols, int base2k, int k) {
    if (batch <= 0 || ncols <= 0 || size <= 0) return PZ_OK;
    RshArgs g;
    g.data = data; g.bs = bs; g.cols = cols; g.size = size; g.col0 = col0; g.ncols = ncols; g.n = (int)M->n; g.batch = batch;
    g.base2k = base2k; g.k = k;
    KTimer kt(M, PZ_K_NORMALIZE);
    for (int b0 = 0; b0 < batch; b0 += 65535) {   // gridDim.z limit
        RshArgs gb = g;
        gb.data = data + (long long)b0 * bs;
        const int nb = std::min(65535, batch - b0);
        hipLaunchKernelGGL(k_rsh_assign, dim3((unsigned)((M->n / 2 + 255) / 256), (unsigned)ncols, (unsigned)nb), dim3(256), 0, M->stream, gb);
    }
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}

// vmp_apply_dft_to_dft  [vmp.rs:144-264, zero-tail semantics for limb_offset > 0]
int dev_vmp(pz_module* M, int batch, DV res, DV a, const double* pmat, int rows, int cols_in, int cols_out, int size,
                   int limb_offset) {
    const int nrows = rows * cols_in, ncols = cols_out * size;
    const int a_polys = a.cols * a.size, res_polys = res.cols * res.size;
    const int row_max = std::min(nrows, a_polys);
    const int off = limb_offset * cols_out;
    const int ncomp = off < ncols ? std::min(res_polys, ncols - off) : 0;
    if (res_polys == 0) return PZ_OK;
    if (ncomp == 0 || row_max == 0) {  // nothing to read from the key, or an empty input (a.size = 0: dsize > a.size): the sum is empty
        return launch_ew(M, EW_ZERO, res.p, res.bs, (long long)M->n, nullptr, 0, 0, nullptr, 0, 0, res_polys, batch);
    }
    const int m = (int)M->m;
    KTimer kt(M, PZ_K_VMP);
    if (batch >= 6) {
        const int n_pb = (m + 63) / 64, n_cg = (res_polys + 15) / 16, n_ct = (batch + 7) / 8;
        hipLaunchKernelGGL((k_vmp_lds<8, 2, PZ_VMP_RB>), dim3(n_pb * n_cg * n_ct), dim3(512), 0, M->stream, (double*)res.p, res.bs, res_polys,
                           (const double*)a.p, a.bs, pmat, ncols, off, row_max, ncomp, m, batch, n_pb, n_cg, n_ct);
    } else if (batch >= 4) {
        dim3 grid((m + 63) / 64, (res_polys + 15) / 16, (batch + 3) / 4);
        hipLaunchKernelGGL((k_vmp<4, 4>), grid, dim3(256), 0, M->stream, (double*)res.p, res.bs, res_polys, (const double*)a.p,
                           a.bs, pmat, ncols, off, row_max, ncomp, m, batch);
    } else {
        dim3 grid((m + 63) / 64, (res_polys + 15) / 16, batch);
        hipLaunchKernelGGL((k_vmp<1, 4>), grid, dim3(256), 0, M->stream, (double*)res.p, res.bs, res_polys, (const double*)a.p,
                           a.bs, pmat, ncols, off, row_max, ncomp, m, batch);
    }
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}


// vec_znx_(big_)normalize on one column  [normalize.rs:18-401]
int dev_normalize(pz_module* M, int batch, DV res, int res_base2k, long long res_offset, int res_col, DV a, int a_base2k,
                         int a_col, const NzCombine* cb) {
    if (cb && res_base2k != a_base2k) return fail(PZ_ERR_INVALID, "normalize with combined stores: one base2k only");
    NzArgs g;
    g.mode = cb ? cb->mode : 1;
    for (int u = 0; u < 2; ++u) { g.col2[u] = cb ? cb->col2[u] : 0; g.mode2[u] = cb ? cb->mode2[u] : 0; }
    g.res = (long long*)res.p; g.a = (const long long*)a.p;
    g.res_bs = res.bs; g.a_bs = a.bs;
    g.n = (int)M->n; g.batch = batch;
    g.res_cols = res.cols; g.res_size = res.size; g.res_col = res_col;
    g.a_cols = a.cols; g.a_size = a.size; g.a_col = a_col;
    g.res_base2k = res_base2k; g.a_base2k = a_base2k;
    g.lsh = g.res_end = g.res_start = g.a_end = g.a_start = 0;
    const long long total = (long long)batch * (long long)M->n;
    const int blocks = (int)((total + 255) / 256);
    if (blocks == 0) return PZ_OK;
    KTimer kt(M, PZ_K_NORMALIZE);
    if (res_base2k == a_base2k) {
        const long long k = res_base2k;
        long long lsh = res_offset % k, lo = res_offset / k;
        if (res_offset < 0 && lsh != 0) { lsh = (lsh + k) % k; lo -= 1; }
        auto cl = [](long long v, long long lo_, long long hi_) { return v < lo_ ? lo_ : (v > hi_ ? hi_ : v); };
        g.lsh = (int)lsh;
        g.res_end = (int)cl(-lo, 0, res.size);
        g.res_start = (int)cl((long long)a.size - lo, 0, res.size);
        g.a_end = (int)cl(lo, 0, a.size);
        g.a_start = (int)cl((long long)res.size + lo, 0, a.size);
        if (cb) hipLaunchKernelGGL(k_normalize_inter<true>, dim3(blocks), dim3(256), 0, M->stream, g);
        else hipLaunchKernelGGL(k_normalize_inter<false>, dim3(blocks), dim3(256), 0, M->stream, g);
    } else {
        hipLaunchKernelGGL(k_normalize_cross, dim3(blocks), dim3(256), 0, M->stream, g, res_offset);
    }
    PZ_HIP(hipGetLastError());
    return PZ_OK;
}


}  // namespace pz
