// device_ops.hpp — DFT-domain products (VMP / SVP), limb-wise elementwise ops
// and the base-2^k carry/normalize kernels.  All kernels are batched: object b
// of a batch sits `*_bs` scalars after object b-1.
#pragma once
#include "device_fft.hpp"

namespace pz {

// =================================================================================
// VMP: res[b][c][q] = sum_{r < row_max} a[b][r][q] * P[r][c + off][q]
//   (reference/fft64/vmp.rs:186-264 computes the same sums 4 points at a time)
// a, res: VecZnxDft in device order, polynomial r at r*m cplx.
// P     : device VmpPMat = the rows*cols_in x size*cols_out matrix of spectra,
//         entry (r, c) at (r*ncols + c)*m cplx  (DESIGN.md 3).
// One lane per frequency point; each lane keeps a CT x CC block of accumulators
// (CT ciphertexts x CC output polynomials) so that every P and a value fetched is
// used CT resp. CC times.  Waves of a workgroup take different column tiles of the
// same points/ciphertexts, so their `a` reads hit the CU's L1.
// Columns c >= ncomp are written as zero (OVERWRITE semantics, all limbs written).
// =================================================================================
template <int CT, int CC>
__global__ void __launch_bounds__(256)
k_vmp(double* __restrict__ res, long long res_bs, int res_polys,
      const double* __restrict__ a, long long a_bs,
      const double* __restrict__ pmat, int ncols, int off, int row_max, int ncomp,
      int m, int batch) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int c0 = (blockIdx.y * 4 + wave) * CC;
    const int b0 = blockIdx.z * CT;
    if (c0 >= res_polys || q >= m) return;

    cplx acc[CT][CC];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CC; ++j) acc[i][j] = make_double2(0.0, 0.0);

    const cplx* ap[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int b = min(b0 + i, batch - 1);
        ap[i] = reinterpret_cast<const cplx*>(a + (long long)b * a_bs) + q;
    }
    const cplx* pp[CC];
#pragma unroll
    for (int j = 0; j < CC; ++j) {
        const int c = min(c0 + j, ncomp - 1);
        pp[j] = reinterpret_cast<const cplx*>(pmat) + (long long)(c + off) * m + q;
    }
    const long long prow = (long long)ncols * m;

    if (c0 < ncomp) {
        for (int r = 0; r < row_max; ++r) {
            cplx av[CT], pv[CC];
#pragma unroll
            for (int i = 0; i < CT; ++i) av[i] = ap[i][(long long)r * m];
#pragma unroll
            for (int j = 0; j < CC; ++j) pv[j] = pp[j][(long long)r * prow];
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < CC; ++j) {
                    acc[i][j].x = __builtin_fma(av[i].x, pv[j].x, acc[i][j].x);
                    acc[i][j].x = __builtin_fma(-av[i].y, pv[j].y, acc[i][j].x);
                    acc[i][j].y = __builtin_fma(av[i].x, pv[j].y, acc[i][j].y);
                    acc[i][j].y = __builtin_fma(av[i].y, pv[j].x, acc[i][j].y);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int b = b0 + i;
        if (b >= batch) continue;
        cplx* rp = reinterpret_cast<cplx*>(res + (long long)b * res_bs) + q;
#pragma unroll
        for (int j = 0; j < CC; ++j) {
            const int c = c0 + j;
            if (c >= res_polys) continue;
            rp[(long long)c * m] = (c < ncomp) ? acc[i][j] : make_double2(0.0, 0.0);
        }
    }
}

// Batched variant for CT >= 8 ciphertexts that share the key: the `a` values of the current
// input row are staged once per workgroup in LDS (double-buffered, prefetched one row ahead)
// and shared by the four waves, each of which owns 4 output columns; P streams straight from
// L2/HBM into registers, one row ahead.  Per row and workgroup: 8 KiB of `a` + 16 KiB of P
// feed 8*64*16 complex MACs, so neither the L1 nor the LDS pipe limits the FP64 issue rate.
template <int CT, int CC, int RB>
__global__ void __launch_bounds__(64 * (16 / CC))
k_vmp_lds(double* __restrict__ res, long long res_bs, int res_polys,
          const double* __restrict__ a, long long a_bs,
          const double* __restrict__ pmat, int ncols, int off, int row_max, int ncomp,
          int m, int batch, int n_pb, int n_cg, int n_ct) {
    constexpr int NW = 16 / CC;          // waves per workgroup; the workgroup covers 16 output columns
    constexpr int NLD = (CT + NW - 1) / NW;  // staging loads per thread and row
    __shared__ cplx a_s[2][RB][NLD * NW][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware decode of the 1-D grid (n_pb point blocks x n_cg column groups x n_ct ciphertext tiles):
    // workgroups are dealt round-robin over the 8 XCDs, so bid % 8 labels the XCD.  Every ciphertext tile
    // of one point block is given to the same XCD, back to back, so that the 256 KiB slice of P they share
    // is fetched from HBM once and then served by that XCD's L2 (speed only; any placement is correct).
    int pb, cg, ct;
    {
        const int bid = blockIdx.x;
        if ((n_pb & 7) == 0) {
            const int xcd = bid & 7, local = bid >> 3;
            ct = local % n_ct;
            const int t = local / n_ct;
            cg = t % n_cg;
            pb = (t / n_cg) * 8 + xcd;
        } else {
            ct = bid % n_ct;
            const int t = bid / n_ct;
            cg = t % n_cg;
            pb = t / n_cg;
        }
    }
    const int q = min(pb * 64 + lane, m - 1);
    const bool q_ok = pb * 64 + lane < m;
    const int c0 = (cg * NW + wave) * CC;
    const int b0 = ct * CT;

    cplx acc[CT][CC];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CC; ++j) acc[i][j] = make_double2(0.0, 0.0);

    // cooperative staging: wave w fetches ciphertexts w + NW*i at point `lane`
    const cplx* ap[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int b = min(b0 + wave + NW * i, batch - 1);
        ap[i] = reinterpret_cast<const cplx*>(a + (long long)b * a_bs) + q;
    }
    const cplx* pp[CC];
#pragma unroll
    for (int j = 0; j < CC; ++j) {
        const int c = min(c0 + j, ncomp - 1);
        pp[j] = reinterpret_cast<const cplx*>(pmat) + (long long)(c + off) * m + q;
    }
    const long long prow = (long long)ncols * m;

    // rows are processed in blocks of RB; the loads of block k+1 are in flight while block k is computed
    // (RB*CT*CC*4 FMAs per wave ~ 1000+ cycles of cover for an HBM miss)
    cplx pn[RB * CC], pv[RB * CC];
    // `a` rows go HBM -> LDS directly (global_load_lds, 16 B per lane, lane-linear destination);
    // P rows go to registers.  Both are issued one block of RB rows ahead.
#define PZ_LDS_DMA(GPTR, LPTR)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(GPTR),                \
                                     (__attribute__((address_space(3))) void*)(LPTR), 16, 0, 0)
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        const int r = min(k, row_max - 1);
#pragma unroll
        for (int i = 0; i < NLD; ++i) PZ_LDS_DMA(ap[i] + (long long)r * m, &a_s[0][k][wave + NW * i][0]);
#pragma unroll
        for (int j = 0; j < CC; ++j) pn[k * CC + j] = pp[j][(long long)r * prow];
    }
    __syncthreads();
    for (int r0 = 0; r0 < row_max; r0 += RB) {
        const int half = (r0 / RB) & 1;
#pragma unroll
        for (int t = 0; t < RB * CC; ++t) pv[t] = pn[t];
        if (r0 + RB < row_max) {
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const int r = min(r0 + RB + k, row_max - 1);
#pragma unroll
                for (int i = 0; i < NLD; ++i) PZ_LDS_DMA(ap[i] + (long long)r * m, &a_s[half ^ 1][k][wave + NW * i][0]);
#pragma unroll
                for (int j = 0; j < CC; ++j) pn[k * CC + j] = pp[j][(long long)r * prow];
            }
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            if (r0 + k < row_max) {
#pragma unroll
                for (int i = 0; i < CT; ++i) {
                    const cplx av = a_s[half][k][i][lane];
#pragma unroll
                    for (int j = 0; j < CC; ++j) {
                        const cplx p_ = pv[k * CC + j];
                        acc[i][j].x = __builtin_fma(av.x, p_.x, acc[i][j].x);
                        acc[i][j].x = __builtin_fma(-av.y, p_.y, acc[i][j].x);
                        acc[i][j].y = __builtin_fma(av.x, p_.y, acc[i][j].y);
                        acc[i][j].y = __builtin_fma(av.y, p_.x, acc[i][j].y);
                    }
                }
            }
        }
        __syncthreads();
    }
#undef PZ_LDS_DMA
    if (!q_ok || c0 >= res_polys) return;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int b = b0 + i;
        if (b >= batch) continue;
        cplx* rp = reinterpret_cast<cplx*>(res + (long long)b * res_bs) + q;
#pragma unroll
        for (int j = 0; j < CC; ++j) {
            const int c = c0 + j;
            if (c >= res_polys) continue;
            rp[(long long)c * m] = (c < ncomp) ? acc[i][j] : make_double2(0.0, 0.0);
        }
    }
}

// =================================================================================
// limb-wise elementwise ops on whole polynomials (f64 spectra or i64 limbs)
// =================================================================================

// polynomial (b, j) of operand X sits at X + b*bs + j*ls ; n scalars per polynomial;
// one thread per 2 scalars (16 B)
struct EwArgs {
    void* res; const void* a; const void* b;
    long long res_bs, res_ls, a_bs, a_ls, b_bs, b_ls;
    int nlimbs, n, batch, op;
};

__global__ void __launch_bounds__(256) k_ew(EwArgs g) {
    const long long half = g.n >> 1;
    const long long total = (long long)g.batch * g.nlimbs * half;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long e = t % half;
        const long long pj = t / half;
        const int j = (int)(pj % g.nlimbs);
        const long long b = pj / g.nlimbs;
        double2* r = reinterpret_cast<double2*>((double*)g.res + b * g.res_bs + j * g.res_ls) + e;
        const double2* x = g.a ? reinterpret_cast<const double2*>((const double*)g.a + b * g.a_bs + j * g.a_ls) + e : nullptr;
        const double2* y = g.b ? reinterpret_cast<const double2*>((const double*)g.b + b * g.b_bs + j * g.b_ls) + e : nullptr;
        switch (g.op) {
            case EW_ZERO: *r = make_double2(0.0, 0.0); break;
            case EW_COPY: *r = *x; break;
            case EW_NEG: { double2 v = *x; *r = make_double2(-v.x, -v.y); } break;
            case EW_ADD: { double2 v = *x, w = *y; *r = make_double2(v.x + w.x, v.y + w.y); } break;
            case EW_SUB: { double2 v = *x, w = *y; *r = make_double2(v.x - w.x, v.y - w.y); } break;
            case EW_CMUL: { double2 v = *x, w = *y; *r = cmul(v, w); } break;
            case EW_ADD_I64: {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(x);
                const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(y);
                *reinterpret_cast<ulonglong2*>(r) = make_ulonglong2(v.x + w.x, v.y + w.y);
            } break;
            case EW_SUB_I64: {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(x);
                const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(y);
                *reinterpret_cast<ulonglong2*>(r) = make_ulonglong2(v.x - w.x, v.y - w.y);
            } break;
            case EW_NEG_I64: {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(x);
                *reinterpret_cast<ulonglong2*>(r) = make_ulonglong2(0ull - v.x, 0ull - v.y);
            } break;
        }
    }
}

// =================================================================================
// vec_znx_automorphism / vec_znx_big_automorphism  X -> X^p on i64 polynomials
// (reference/znx/automorphism.rs:1-17: res[(i*p) mod 2n] = a[i], negated when the index wraps past n).
// Gather form, so that stores are coalesced: with g = p^-1 mod 2n,  i0 = (j*g) mod 2n,
//   res[j] = a[i0]  if i0 < n,   res[j] = -a[i0 - n]  otherwise.
// The gathers of one polynomial (n*8 B: 512 KiB at n = 2^16) hit in L2: the 1-D grid is decoded so that every block of
// a polynomial runs on the same XCD, i.e. each source line is fetched from HBM once.
// flags: 1 apply the sign; 2 negate every output; 4 `add` only for polynomials whose innermost PolyMap index is 0.
// =================================================================================
struct AutoArgs {
    const long long* src;
    long long* dst;
    const long long* add;  // optional second operand (same coefficient order as dst): dst = +-gather(src) + add
    PolyMap sm, dm, am;
    int npolys, n;
    unsigned mul;          // gather multiplier g (odd, < 2n)
    int flags;
    // flags & 8: the output leaves as 16-bit values in the fused tail's tile order (k_inv_tail, TailArgs::d16*) at dst16 + map_off(dm) - the body
    // operand of the spectral automorphism forms at 2 B per coefficient.  A value beyond 16 bits sets *wide: the tail then gathers the body
    // itself (the older "fold" form) instead of reading the copies - un-normalized inputs stay correct, only slower.
    short* dst16;
    unsigned* wide;
    int t16_m1, t16_cb, t16_m2sh;   // tile: m1 rows, cb columns per block, log2 m2
    // cond_total != 0: the launch only works if *wide is up (the i64 fallback of the 16-bit body pre-pass: values that did not fit) - a small grid
    // whose blocks return at once otherwise, and walk the cond_total block positions of the plain launch if it is
    int cond_total;
};
// tile-order position of coefficient j (TailArgs::d16*): j = h m + j1 m2 + cc  ->  (cc / cb) 2 m1 cb + (h m1 + j1) cb + cc % cb
__device__ __forceinline__ long long auto_t16_pos(const AutoArgs& g, unsigned j) {
    const unsigned row = j >> g.t16_m2sh;                       // h m1 + j1
    const unsigned cc = j & ((1u << g.t16_m2sh) - 1u);
    return (long long)(cc / (unsigned)g.t16_cb) * (2 * g.t16_m1 * g.t16_cb) + (long long)row * g.t16_cb + (cc % (unsigned)g.t16_cb);
}
__device__ __forceinline__ bool auto_wide(unsigned long long v) { return (v + 32768ull) >= 65536ull; }

__device__ __forceinline__ void automorphism_block(const AutoArgs& g, int bid) {
    const int bpp = g.n >= 512 ? g.n / 512 : 1;  // blocks per polynomial, 2 coefficients per thread
    const int xcd = bid & 7, r = bid >> 3;
    const int poly = (r / bpp) * 8 + xcd, blk = r % bpp;
    if (poly >= g.npolys) return;
    const long long* src = g.src + map_off(g.sm, poly);
    long long* dst = g.dst + map_off(g.dm, poly);
    const bool has_add = g.add != nullptr && (!(g.flags & 4) || (poly % g.am.ni) == 0);
    const long long* add = has_add ? g.add + map_off(g.am, poly) : nullptr;
    const unsigned mask2 = 2u * (unsigned)g.n - 1u, nn = (unsigned)g.n;
    const int j = blk * 512 + threadIdx.x * 2;
    if (j >= g.n) return;
    unsigned long long out[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned i0 = ((unsigned)(j + e) * g.mul) & mask2;
        unsigned long long v = (unsigned long long)src[i0 & (nn - 1u)];
        bool neg = (g.flags & 1) && i0 >= nn;
        if (g.flags & 2) neg = !neg;
        out[e] = neg ? 0ull - v : v;
    }
    if (has_add) {
        const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(add + j);
        out[0] += w.x;
        out[1] += w.y;
    }
    if (g.flags & 8) {   // (j even: both values in one row, one column block)
        short2 s2;
        s2.x = (short)out[0]; s2.y = (short)out[1];
        *reinterpret_cast<short2*>(g.dst16 + map_off(g.dm, poly) + auto_t16_pos(g, (unsigned)j)) = s2;
        if (auto_wide(out[0]) || auto_wide(out[1])) atomicOr(g.wide, 1u);
        return;
    }
    *reinterpret_cast<ulonglong2*>(dst + j) = make_ulonglong2(out[0], out[1]);
}

__global__ void __launch_bounds__(256) k_automorphism(AutoArgs g) { automorphism_block(g, blockIdx.x); }
// (the conditional form, AutoArgs::cond_total - a kernel of its own: the walk over block positions would cost the plain launch registers)
__global__ void __launch_bounds__(256) k_automorphism_cond(AutoArgs g) {
    if (__builtin_amdgcn_readfirstlane((int)*g.wide) == 0) return;
    for (int vb = blockIdx.x; vb < g.cond_total; vb += gridDim.x) automorphism_block(g, vb);
}

// The same map for Galois elements without locality (round 4).  k_automorphism lets consecutive lanes own consecutive OUTPUT coefficients and
// gather their sources g apart: for g = 5^k, the steps of glwe_trace, ... every 8-byte load touches a line of its own and the 128 bytes that
// come back from L2 per element (16 x the useful bytes) pace the kernel - 3.0 ms per 1024 ciphertexts x 8 limbs against 1.8 for g = 5.  Here
// a thread owns a CHUNK of 4 consecutive outputs at j0 = g^-1 * 4e mod 2N (a multiple of 4), whose sources are 4e + x g (x = 0..3): for each x
// the lanes of a wave read every fourth coefficient of one contiguous run (4 x amplification instead of 16), and the outputs (and the `add`
// operand) move as whole 32-byte sectors.  Chunk e and chunk e + N/4 are the two halves of the index range mod 2N: e runs over [0, N/4)
// and a wrapped base (j0 >= N) is the same position with every sign flipped.
__device__ __forceinline__ void automorphism_chunk_block(const AutoArgs& g, unsigned hinv, int bid) {
    const int bpp = g.n / 1024;   // blocks per polynomial (launcher: n >= 1024)
    const int xcd = bid & 7, r = bid >> 3;
    const int poly = (r / bpp) * 8 + xcd, blk = r % bpp;
    if (poly >= g.npolys) return;
    const long long* src = g.src + map_off(g.sm, poly);
    long long* dst = g.dst + map_off(g.dm, poly);
    const bool has_add = g.add != nullptr && (!(g.flags & 4) || (poly % g.am.ni) == 0);
    const long long* add = has_add ? g.add + map_off(g.am, poly) : nullptr;
    const unsigned mask2 = 2u * (unsigned)g.n - 1u, nn = (unsigned)g.n;
    const unsigned e4 = 4u * (unsigned)(blk * 256 + threadIdx.x);
    const unsigned j0v = (hinv * e4) & mask2;
    const unsigned j0 = j0v & (nn - 1u);
    const unsigned i0 = e4 + (j0v >= nn ? nn : 0u);
    unsigned long long out[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const unsigned i = (i0 + (unsigned)x * g.mul) & mask2;
        unsigned long long v = (unsigned long long)src[i & (nn - 1u)];
        bool neg = (g.flags & 1) && i >= nn;
        if (g.flags & 2) neg = !neg;
        out[x] = neg ? 0ull - v : v;
    }
    if (has_add) {
        const ulonglong2 w0 = *reinterpret_cast<const ulonglong2*>(add + j0), w1 = *reinterpret_cast<const ulonglong2*>(add + j0 + 2);
        out[0] += w0.x; out[1] += w0.y; out[2] += w1.x; out[3] += w1.y;
    }
    if (g.flags & 8) {   // (j0 a multiple of 4: the four values share a row and a column block)
        short4 s4;
        s4.x = (short)out[0]; s4.y = (short)out[1]; s4.z = (short)out[2]; s4.w = (short)out[3];
        *reinterpret_cast<short4*>(g.dst16 + map_off(g.dm, poly) + auto_t16_pos(g, j0)) = s4;
        if (auto_wide(out[0]) || auto_wide(out[1]) || auto_wide(out[2]) || auto_wide(out[3])) atomicOr(g.wide, 1u);
        return;
    }
    *reinterpret_cast<ulonglong2*>(dst + j0) = make_ulonglong2(out[0], out[1]);
    *reinterpret_cast<ulonglong2*>(dst + j0 + 2) = make_ulonglong2(out[2], out[3]);
}

__global__ void __launch_bounds__(256) k_automorphism_chunk(AutoArgs g, unsigned hinv) { automorphism_chunk_block(g, hinv, blockIdx.x); }
__global__ void __launch_bounds__(256) k_automorphism_chunk_cond(AutoArgs g, unsigned hinv) {
    if (__builtin_amdgcn_readfirstlane((int)*g.wide) == 0) return;
    for (int vb = blockIdx.x; vb < g.cond_total; vb += gridDim.x) automorphism_chunk_block(g, hinv, vb);
}

// The 16-bit tile-order output through LDS (round 6): a polynomial of 16-bit values is n x 2 B = 128 KiB at N = 2^16 - it fits the CU.  One workgroup per
// polynomial: the source is read ONCE in whole lines (the gathers above fetch every line 4 - 5 x from L2: 1.76 ms per 512 x 16 limbs against the
// 0.8 its bytes take), narrowed to int16 into LDS, and the outputs are produced in the tile order itself - 4 consecutive values per thread, whole
// 512-byte runs per wave - by reading LDS at the Galois-permuted index.  Any Galois element alike.  A value outside +-32767 raises *wide
// (-32768 included: its negation does not fit).  flags as k_automorphism (1 sign, 2 negate all); `add` (natural order, i64) is added to - flags & 16:
// subtracted from - the permuted value, and a sum beyond 16 bits raises *wide too (the add / sub forms' body operand phi(body) +- a0).
__global__ void __launch_bounds__(1024) k_automorphism_t16(AutoArgs g) {
    extern __shared__ short a16[];   // the source polynomial, natural order
    const int poly = blockIdx.x;
    const long long* src = g.src + map_off(g.sm, poly);
    const int tid = threadIdx.x;
    bool wide = false;
    for (int i = tid * 2; i < g.n; i += 2048 * 4) {   // 4 x 16 B in flight per thread
        typedef unsigned long long pz_u64x2 __attribute__((ext_vector_type(2)));
        pz_u64x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 2048 < g.n) v[u] = __builtin_nontemporal_load(reinterpret_cast<const pz_u64x2*>(src + i + u * 2048));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 2048 < g.n) {
                wide = wide || (v[u].x + 32767ull) >= 65535ull || (v[u].y + 32767ull) >= 65535ull;
                short2 s2;
                s2.x = (short)v[u].x; s2.y = (short)v[u].y;
                *reinterpret_cast<short2*>(a16 + i + u * 2048) = s2;
            }
    }
    if (wide) atomicOr(g.wide, 1u);
    __syncthreads();
    short* dst = g.dst16 + map_off(g.dm, poly);
    const long long* add = g.add ? g.add + map_off(g.am, poly) : nullptr;
    bool wide2 = false;
    const unsigned mask2 = 2u * (unsigned)g.n - 1u, nn = (unsigned)g.n;
    const unsigned m2 = 1u << g.t16_m2sh, tile = 2u * (unsigned)g.t16_m1 * (unsigned)g.t16_cb;
    for (unsigned t = (unsigned)tid * 4u; t < nn; t += 4096u) {
        // tile-order position t -> coefficient j: block t / tile, row (t % tile) / cb = h m1 + j1, column block * cb + (t % cb)
        const unsigned blk = t / tile, rem = t % tile;
        const unsigned j = (rem / (unsigned)g.t16_cb) * m2 + blk * (unsigned)g.t16_cb + (rem % (unsigned)g.t16_cb);
        short o[4];
        long long w[4] = {0, 0, 0, 0};
        if (add) {   // (j a multiple of 4: 32 contiguous bytes)
            const longlong2 w0 = *reinterpret_cast<const longlong2*>(add + j), w1 = *reinterpret_cast<const longlong2*>(add + j + 2);
            w[0] = w0.x; w[1] = w0.y; w[2] = w1.x; w[3] = w1.y;
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const unsigned i0 = ((j + (unsigned)x) * g.mul) & mask2;
            const int v = (int)a16[i0 & (nn - 1u)];
            bool neg = (g.flags & 1) && i0 >= nn;
            if (g.flags & 2) neg = !neg;
            const unsigned long long s = (unsigned long long)(long long)(neg ? -v : v) + ((g.flags & 16) ? 0ull - (unsigned long long)w[x] : (unsigned long long)w[x]);
            wide2 = wide2 || (s + 32767ull) >= 65535ull;
            o[x] = (short)s;
        }
        short4 s4;
        s4.x = o[0]; s4.y = o[1]; s4.z = o[2]; s4.w = o[3];
        *reinterpret_cast<short4*>(dst + t) = s4;
    }
    if (wide2) atomicOr(g.wide, 1u);
}

// =================================================================================
// vec_znx_rotate family with a per-ciphertext exponent (reference/znx/rotate.rs:3-27: res = X^k * src), gather form:
//   res[j] = +-src[(j - k) mod 2n]   (negated when that index is >= n).
// mode 0: dst = X^k src;  1: dst = X^k src - src  (vec_znx_mul_xp_minus_one, mul_xp_minus_one.rs:13-37);
// mode 2: dst += X^k src - src  (mul_xp_minus_one followed by glwe_add_assign: the standard blind-rotation step).
// k of polynomial p: shift[(p / polys_per_batch) * shift_bs + shift_idx] when shift != nullptr, else shift_const.
// =================================================================================
struct RotArgs {
    const long long* src;
    long long* dst;
    PolyMap sm, dm;
    int npolys, n, polys_per_batch, mode;
    const long long* shift;
    long long shift_bs, shift_idx, shift_const;
};

__global__ void __launch_bounds__(256) k_rotate(RotArgs g) {
    const int bpp = g.n >= 512 ? g.n / 512 : 1;
    const int poly = blockIdx.x / bpp, blk = blockIdx.x % bpp;
    if (poly >= g.npolys) return;
    const long long k = g.shift ? g.shift[(long long)(poly / g.polys_per_batch) * g.shift_bs + g.shift_idx] : g.shift_const;
    const unsigned mask2 = 2u * (unsigned)g.n - 1u, nn = (unsigned)g.n;
    const unsigned kk = (unsigned)((unsigned long long)k & (unsigned long long)mask2);
    const long long* src = g.src + map_off(g.sm, poly);
    long long* dst = g.dst + map_off(g.dm, poly);
    const int j = blk * 512 + threadIdx.x * 2;
    if (j >= g.n) return;
    unsigned long long out[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned i0 = ((unsigned)(j + e) - kk) & mask2;
        const unsigned long long v = (unsigned long long)src[i0 & (nn - 1u)];
        out[e] = i0 >= nn ? 0ull - v : v;
    }
    if (g.mode >= 1) {
        const ulonglong2 s = *reinterpret_cast<const ulonglong2*>(src + j);
        out[0] -= s.x;
        out[1] -= s.y;
    }
    if (g.mode == 2) {
        const ulonglong2 d = *reinterpret_cast<const ulonglong2*>(dst + j);
        out[0] += d.x;
        out[1] += d.y;
    }
    *reinterpret_cast<ulonglong2*>(dst + j) = make_ulonglong2(out[0], out[1]);
}

// =================================================================================
// normalize: reference/vec_znx/normalize.rs + reference/znx/normalization.rs.
// The carry chain runs across limbs, never across coefficients, so one thread owns
// one coefficient and walks the limbs with the carry in a register.  Arithmetic is
// exact-integer and mirrors the reference step functions bit for bit.
// =================================================================================
__device__ __forceinline__ long long nz_digit(int k, long long x) {
    return (long long)((unsigned long long)x << (64 - k)) >> (64 - k);
}
__device__ __forceinline__ long long nz_carry(int k, long long x, long long d) {
    return (long long)((unsigned long long)x - (unsigned long long)d) >> k;
}
__device__ __forceinline__ long long wadd(long long a, long long b) {
    return (long long)((unsigned long long)a + (unsigned long long)b);
}
__device__ __forceinline__ long long wshl(long long a, int s) { return (long long)((unsigned long long)a << s); }

struct NzArgs {
    long long* res; const long long* a;
    long long res_bs, a_bs;      // batch strides (scalars)
    int n, batch;
    int res_cols, res_size, res_col;
    int a_cols, a_size, a_col;
    int res_base2k, a_base2k;
    // same-base plan (normalize.rs:83-101)
    int lsh, res_end, res_start, a_end, a_start;
    // k_normalize_inter<COMBINE>: how the digits reach `res` (mode) and up to two further columns of the same container that take them too
    // (GLWE tensoring: the diagonal terms are stored in their own column and subtracted from the cross columns, operations/glwe.rs:762-805
    // - five element-wise passes over the tensor in round 2).  Modes: 1 = v, 2 = -v, 3 += v, 4 -= v, 5 (both further columns) = read-only: their digits are subtracted from v first (wrapping i64, as the reference's
    // vec_znx_{copy,negate,add_assign,sub_assign} on the normalized digits); 0: no such destination.
    int mode;
    int col2[2], mode2[2];
};

__device__ __forceinline__ void nz_put(long long* p, int mode, long long v) {
    if (mode == 1) *p = v;
    else if (mode == 2) *p = (long long)(0ull - (unsigned long long)v);
    else if (mode == 3) *p = (long long)((unsigned long long)*p + (unsigned long long)v);
    else if (mode == 4) *p = (long long)((unsigned long long)*p - (unsigned long long)v);
}
// normalize.rs:50-144 (vec_znx_normalize_inter_base2k)
template <bool COMBINE = false>
__global__ void __launch_bounds__(256) k_normalize_inter(NzArgs g) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)g.batch * g.n) return;
    const int i = (int)(t % g.n);
    const long long b = t / g.n;
    const long long* a = g.a + b * g.a_bs + (long long)g.a_col * g.n + i;
    long long* r = g.res + b * g.res_bs + (long long)g.res_col * g.n + i;
    long long* r2a = COMBINE && g.mode2[0] ? g.res + b * g.res_bs + (long long)g.col2[0] * g.n + i : nullptr;
    long long* r2b = COMBINE && g.mode2[1] ? g.res + b * g.res_bs + (long long)g.col2[1] * g.n + i : nullptr;
#define PZ_NZ_STORE(J_, V_)                                                                        \
    {                                                                                              \
        const long long off_ = (long long)(J_) * rls;                                              \
        const long long v_ = (V_);                                                                 \
        if constexpr (COMBINE) {                                                                   \
            if (g.mode2[0] == 5) {   /* the further columns' digits are subtracted on the way to the main column */ \
                long long w_ = (long long)((unsigned long long)v_ - (unsigned long long)r2a[off_]); \
                if (r2b) w_ = (long long)((unsigned long long)w_ - (unsigned long long)r2b[off_]);  \
                nz_put(r + off_, g.mode, w_);                                                      \
            } else {                                                                               \
                nz_put(r + off_, g.mode, v_);                                                      \
                if (r2a) nz_put(r2a + off_, g.mode2[0], v_);                                       \
                if (r2b) nz_put(r2b + off_, g.mode2[1], v_);                                       \
            }                                                                                      \
        } else r[off_] = v_;                                                                       \
    }
    const long long als = (long long)g.a_cols * g.n;
    const long long rls = (long long)g.res_cols * g.n;
    const int k = g.res_base2k;
    const int lsh = g.lsh;
    const int kk = lsh == 0 ? k : k - lsh;

    long long c = 0;
    const int a_out_range = g.a_size > g.a_start ? g.a_size - g.a_start : 0;
    for (int j = 0; j < a_out_range; ++j) {
        const long long x = a[(long long)(g.a_size - j - 1) * als];
        const long long d = nz_digit(kk, x);
        const long long cr = nz_carry(kk, x, d);
        if (j == 0) {
            c = cr;  // znx_normalize_first_step_carry_only, normalization.rs:24-41
        } else {     // znx_normalize_middle_step_carry_only, normalization.rs:107-129
            const long long dpc = wadd(wshl(d, lsh), c);
            c = wadd(cr, nz_carry(k, dpc, nz_digit(k, dpc)));
        }
    }
    for (int j = g.res_start; j < g.res_size; ++j) PZ_NZ_STORE(j, 0)
    const int mid = g.a_start > g.a_end ? g.a_start - g.a_end : 0;
    for (int j = 0; j < mid; ++j) {  // znx_normalize_middle_step<true>, normalization.rs:179-221
        const long long x = a[(long long)(g.a_start - j - 1) * als];
        const long long d = nz_digit(kk, x);
        const long long cr = nz_carry(kk, x, d);
        const long long dpc = wadd(wshl(d, lsh), c);
        const long long x1 = nz_digit(k, dpc);
        PZ_NZ_STORE(g.res_start - j - 1, x1)
        c = wadd(cr, nz_carry(k, dpc, x1));
    }
    for (int j = 0; j < g.res_end; ++j) {
        // limb zeroed, then middle_step_assign / final_step_assign on a zero limb:
        // digit(kk, 0) = 0 -> dpc = c  (normalization.rs:132-157, 254-272)
        const long long x1 = nz_digit(k, c);
        PZ_NZ_STORE(g.res_end - j - 1, x1)
        if (j != g.res_end - 1) c = nz_carry(k, c, x1);
    }
#undef PZ_NZ_STORE
}

// normalize.rs:147-401 (vec_znx_normalize_cross_base2k).  The control flow depends
// only on shapes, so every thread of a launch walks the same path; the three
// per-coefficient scratch values of the reference (a_norm, res_carry, a_carry) live
// in registers and the res limbs are read-modify-written in place.
__global__ void __launch_bounds__(256) k_normalize_cross(NzArgs g, long long res_offset) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)g.batch * g.n) return;
    const int i = (int)(t % g.n);
    const long long b = t / g.n;
    const long long* a = g.a + b * g.a_bs + (long long)g.a_col * g.n + i;
    long long* r = g.res + b * g.res_bs + (long long)g.res_col * g.n + i;
    const long long als = (long long)g.a_cols * g.n;
    const long long rls = (long long)g.res_cols * g.n;
    const int ak = g.a_base2k, rk = g.res_base2k;
    const int a_size = g.a_size, res_size = g.res_size;

    long long a_norm = 0, res_carry = 0, a_carry = 0;
    const long long a_tot_bits = (long long)a_size * ak;
    const long long res_tot_bits = (long long)res_size * rk;
    long long lsh = res_offset % ak;
    long long limbs_offset = res_offset / ak;
    if (res_offset < 0 && lsh != 0) {
        lsh = (lsh + ak) % ak;
        limbs_offset -= 1;
    }
    const int lsh_pos = (int)lsh;
    auto clampll = [](long long v, long long lo, long long hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const long long res_end_bit = clampll(-limbs_offset * ak, 0, res_tot_bits);
    const long long res_start_bit = clampll(a_tot_bits - limbs_offset * ak, 0, res_tot_bits);
    const long long a_end_bit = clampll(limbs_offset * ak, 0, a_tot_bits);
    const long long a_start_bit = clampll(res_tot_bits + limbs_offset * ak, 0, a_tot_bits);
    const int res_end = (int)(res_end_bit / rk);
    const int res_start = (int)((res_start_bit + rk - 1) / rk);
    const int a_end = (int)(a_end_bit / ak);
    const int a_start = (int)((a_start_bit + ak - 1) / ak);

    for (int j = 0; j < res_size; ++j) r[(long long)j * rls] = 0;
    if (res_start == 0) return;

    const int kk = lsh_pos == 0 ? ak : ak - lsh_pos;
    const int a_out_range = a_size > a_start ? a_size - a_start : 0;
    for (int j = 0; j < a_out_range; ++j) {
        const long long x = a[(long long)(a_size - j - 1) * als];
        const long long d = nz_digit(kk, x);
        const long long cr = nz_carry(kk, x, d);
        if (j == 0) a_carry = cr;
        else {
            const long long dpc = wadd(wshl(d, lsh_pos), a_carry);
            a_carry = wadd(cr, nz_carry(ak, dpc, nz_digit(ak, dpc)));
        }
    }
    int res_acc_left = rk;
    int res_limb = res_start - 1;
    const int mid = a_start > a_end ? a_start - a_end : 0;
    long long cur = 0;  // value of res[res_limb] (zero until touched)
    bool done = false;
    for (int j = 0; j < mid && !done; ++j) {
        const int a_limb = a_start - j - 1;
        int a_take_left = ak;
        {   // znx_normalize_middle_step<true>(a_base2k, lsh, a_norm, a_slice, a_carry)
            const long long x = a[(long long)a_limb * als];
            const long long d = nz_digit(kk, x);
            const long long cr = nz_carry(kk, x, d);
            const long long dpc = wadd(wshl(d, lsh_pos), a_carry);
            a_norm = nz_digit(ak, dpc);
            a_carry = wadd(cr, nz_carry(ak, dpc, a_norm));
        }
        if (j == 0) {
            if ((a_tot_bits - a_start_bit) % ak != 0) {
                const int take = (int)((a_tot_bits - a_start_bit) % ak);
                // znx_mul_power_of_two_assign(-take, a_norm), znx/mul.rs:30-49
                const long long sign_bit = (a_norm >> 63) & 1;
                const long long bias = ((long long)1 << (take - 1)) - sign_bit;
                a_norm = wadd(a_norm, bias) >> take;
                a_take_left -= take;
            } else if ((res_tot_bits - res_start_bit) % rk != 0) {
                res_acc_left -= (int)((res_tot_bits - res_start_bit) % rk);
            }
        }
        for (;;) {
            const int a_take = min(min(ak, a_take_left), res_acc_left);
            if (a_take != 0) {  // znx_extract_digit_addmul(a_take, scale, res_slice, a_norm)
                const int scale = rk - res_acc_left;
                const long long d = nz_digit(a_take, a_norm);
                a_norm = nz_carry(a_take, a_norm, d);
                cur = wadd(cur, wshl(d, scale));
                a_take_left -= a_take;
                res_acc_left -= a_take;
            }
            if (res_acc_left == 0 || a_limb == 0) {
                if (a_limb == 0 && a_take_left == 0) {
                    a_carry = wadd(a_carry, a_norm);
                    if (res_acc_left != 0) {
                        const int scale = rk - res_acc_left;
                        const long long d = nz_digit(res_acc_left, a_carry);
                        a_carry = nz_carry(res_acc_left, a_carry, d);
                        cur = wadd(cur, wshl(d, scale));
                    }
                    {   // znx_normalize_middle_step_assign(res_base2k, 0, res_slice, res_carry)
                        const long long d = nz_digit(rk, cur);
                        const long long cr = nz_carry(rk, cur, d);
                        const long long dpc = wadd(d, res_carry);
                        cur = nz_digit(rk, dpc);
                        res_carry = wadd(cr, nz_carry(rk, dpc, cur));
                    }
                    res_carry = wadd(res_carry, a_carry);
                    done = true;
                    break;
                }
                if (res_limb == 0) {
                    done = true;
                    break;
                }
                r[(long long)res_limb * rls] = cur;
                cur = 0;
                res_acc_left += rk;
                res_limb -= 1;
            }
            if (a_take_left == 0) {
                a_carry = wadd(a_carry, a_norm);
                break;
            }
        }
    }
    r[(long long)res_limb * rls] = cur;

    if (res_end != 0) {
        long long c = (a_start == a_end) ? a_carry : res_carry;
        for (int j = 0; j < res_end; ++j) {
            long long* rp = r + (long long)(res_end - j - 1) * rls;
            const long long x = *rp;
            const long long d = nz_digit(rk, x);
            if (j == res_end - 1) {
                *rp = nz_digit(rk, wadd(d, c));
            } else {
                const long long cr = nz_carry(rk, x, d);
                const long long dpc = wadd(d, c);
                const long long x1 = nz_digit(rk, dpc);
                *rp = x1;
                c = wadd(cr, nz_carry(rk, dpc, x1));
            }
        }
    }
}

// =================================================================================
// vec_znx_rsh_assign (reference/vec_znx/shift.rs:186-243): res >>= k bits in place through the shifted normalization steps
// (znx/normalization.rs, lsh = (base2k - k mod base2k) mod base2k).  The carry chain runs across limbs only: one thread owns
// two coefficients and replays the reference's step sequence literally, including its limb aliasing in the last loop.
// =================================================================================
struct RshArgs {
    long long* data;
    long long bs;        // scalars between batch objects
    int cols, size, col0, ncols, n, batch;
    int base2k, k;
};

__global__ void __launch_bounds__(256) k_rsh_assign(RshArgs g) {
    const int per = g.n / 2;                       // threads per polynomial column: two adjacent coefficients (16 B) each
    const int bk = g.base2k;
    int steps = g.k / bk;
    const int k_rem = g.k % bk;
    if (k_rem != 0) steps += 1;
    const int lsh = (bk - k_rem) % bk;
    const int kk = lsh == 0 ? bk : bk - lsh;
    if (steps > g.size) steps = g.size;
    const long long ls = (long long)g.cols * g.n;   // limb stride
    // grid = (blocks along a column, columns, batch): no per-thread division
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    long long* base = g.data + (long long)blockIdx.z * g.bs + (long long)(g.col0 + blockIdx.y) * g.n + 2 * e;
    long long carry[2] = {0, 0};
    for (int j = 0; j < steps; ++j) {   // limbs that fall off: carry only (:219-225)
        const longlong2 v2 = *reinterpret_cast<const longlong2*>(base + (long long)(g.size - j - 1) * ls);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long v = h ? v2.y : v2.x;
            const long long d = nz_digit(kk, v);
            const long long cr = nz_carry(kk, v, d);
            if (j == 0) {
                carry[h] = cr;
            } else {
                const long long dpc = wadd(wshl(d, lsh), carry[h]);
                carry[h] = wadd(cr, nz_carry(bk, dpc, nz_digit(bk, dpc)));
            }
        }
    }
    for (int j = 0; j + steps < g.size; ++j) {   // shifted normalization (:228-232)
        const longlong2 v2 = *reinterpret_cast<const longlong2*>(base + (long long)(g.size - steps - j - 1) * ls);
        long long nv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long v = h ? v2.y : v2.x;
            const long long d = nz_digit(kk, v);
            const long long cr = nz_carry(kk, v, d);
            const long long dpc = wadd(wshl(d, lsh), carry[h]);
            nv[h] = nz_digit(bk, dpc);
            carry[h] = wadd(cr, nz_carry(bk, dpc, nv[h]));
        }
        *reinterpret_cast<longlong2*>(base + (long long)(g.size - j - 1) * ls) = make_longlong2(nv[0], nv[1]);
    }
    for (int j = 0; j < steps; ++j) {   // top limbs (:235-242), literally: zero limb j, then step limb steps-1-j
        *reinterpret_cast<longlong2*>(base + (long long)j * ls) = make_longlong2(0, 0);
        const longlong2 v2 = *reinterpret_cast<const longlong2*>(base + (long long)(steps - j - 1) * ls);
        long long nv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long v = h ? v2.y : v2.x;
            const long long d = nz_digit(kk, v);
            const long long dpc = wadd(wshl(d, lsh), carry[h]);
            nv[h] = nz_digit(bk, dpc);
            if (j != 0) carry[h] = wadd(nz_carry(kk, v, d), nz_carry(bk, dpc, nv[h]));
        }
        *reinterpret_cast<longlong2*>(base + (long long)(steps - j - 1) * ls) = make_longlong2(nv[0], nv[1]);
    }
}

}  // namespace pz
