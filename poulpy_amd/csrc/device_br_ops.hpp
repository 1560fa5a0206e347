// device_br_ops.hpp — kernels of the composed blind-rotation path: the DFT-domain accumulation steps (k_xai_acc, k_xai_ext),
// the extended rotation's table set-up (k_br_ext_init) and the fused block step (k_br_block, k_br_block_lds).
#pragma once
#include "device_fft.hpp"

namespace pz {

// =================================================================================
// Blind-rotation accumulation step (poulpy-bin-fhe blind_rotation/algorithms/cggi/algorithm.rs:331-335):
//   acc[b][p] += DFT(X^a_b) (.) v[b][p] - v[b][p]      for the cols*size polynomials p of ciphertext b,
// a_b = lwe_2n[b][1 + idx] mod 2n.  The reference multiplies by a prepared monomial x_pow_a[a] (an SvpPPol); in this
// backend's spectrum order DFT(X^a)[q] = exp(2 pi i a (4q+1) / 2n), read from the 2n-entry root table w2n, so no
// 2n x n table is needed (it would be 4 GiB at n = 2^14).
// =================================================================================
struct XaiArgs {
    cplx* acc;
    const cplx* v;
    long long acc_bs, v_bs;      // points between ciphertexts
    int polys, m, batch;
    const long long* lwe;        // [batch][n_lwe + 1]
    long long lwe_bs, idx;       // a_b = lwe[b*lwe_bs + 1 + idx]
    const cplx* w2n;             // exp(2 pi i t / 2n), t < 2n = 4m
};

__global__ void __launch_bounds__(256) k_xai_acc(XaiArgs g) {
    const long long per_ct = (long long)g.polys * g.m;
    const long long total = (long long)g.batch * per_ct;
    const unsigned mask = 4u * (unsigned)g.m - 1u;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / per_ct, e = t % per_ct;
        const unsigned q = (unsigned)(e % g.m);
        const unsigned a = (unsigned)((unsigned long long)g.lwe[b * g.lwe_bs + 1 + g.idx] & (unsigned long long)mask);
        const cplx x = g.w2n[(a * (4u * q + 1u)) & mask];
        const cplx v = g.v[b * g.v_bs + e];
        cplx r = g.acc[b * g.acc_bs + e];
        const cplx xv = cmul(x, v);
        r.x = (r.x + xv.x) - v.x;
        r.y = (r.y + xv.y) - v.y;
        g.acc[b * g.acc_bs + e] = r;
    }
}

// =================================================================================
// execute_block_binary_extended (algorithm.rs:121-273): ext accumulators per ciphertext, laid out [b][e].
// k_br_ext_init (:180-190): acc[b][i] col 0 = X^(b_hi (+1)) * lut[j], with b_pos = lwe[b][0] mod 2 n ext, b_hi = b_pos / ext,
//   b_lo = b_pos mod ext; i < b_lo: j = ext - b_lo + i and one more unit of rotation, else j = i - b_lo.
// k_xai_ext (:205-254): acc_add[b][i] += DFT(X^mult) (.) v[b][j] - v[b][i] with (j, mult) chosen from a = lwe[b][1+idx] as the
//   reference does, INCLUDING its skipped updates when the multiplier would be X^0 (:217, :233, :244).
// =================================================================================
struct BrExtInitArgs {
    long long* acc;          // [batch*ext] GLWE(cols, rsz), zeroed beforehand
    const long long* lut;    // ext x VecZnx(1, lut_size)
    const long long* lwe;
    long long lwe_bs;
    int n, log_ext, cols, rsz, lut_size, nl, batch;
};
__global__ void __launch_bounds__(256) k_br_ext_init(BrExtInitArgs g) {
    const int ext = 1 << g.log_ext;
    const int be = blockIdx.z, limb = blockIdx.y;
    const int b = be >> g.log_ext, i = be & (ext - 1);
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (t0 >= g.n) return;
    const unsigned maskx = 2u * (unsigned)g.n * (unsigned)ext - 1u, mask2 = 2u * (unsigned)g.n - 1u;
    const unsigned b_pos = (unsigned)((unsigned long long)g.lwe[(long long)b * g.lwe_bs] & (unsigned long long)maskx);
    const unsigned b_hi = b_pos >> g.log_ext, b_lo = b_pos & (unsigned)(ext - 1);
    const int j = (unsigned)i < b_lo ? ext - (int)b_lo + i : i - (int)b_lo;
    const unsigned kk = ((unsigned)i < b_lo ? b_hi + 1u : b_hi) & mask2;
    const long long* src = g.lut + ((long long)j * g.lut_size + limb) * g.n;
    long long* dst = g.acc + (((long long)be * g.rsz + limb) * g.cols) * g.n;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned i0 = ((unsigned)(t0 + e) - kk) & mask2;
        const unsigned long long v = (unsigned long long)src[i0 & (unsigned)(g.n - 1)];
        dst[t0 + e] = (long long)(i0 >= (unsigned)g.n ? 0ull - v : v);
    }
}

struct XaiExtArgs {
    cplx* acc;               // [batch*ext][polys][m]
    const cplx* v;           // same layout
    int polys, m, log_ext, batch;
    const long long* lwe;
    long long lwe_bs, idx;
    const cplx* w2n;
};
__global__ void __launch_bounds__(256) k_xai_ext(XaiExtArgs g) {
    const int ext = 1 << g.log_ext;
    const long long per = (long long)g.polys * g.m;
    const long long total = (long long)g.batch * ext * per;
    const unsigned mask = 4u * (unsigned)g.m - 1u;                       // 2n - 1
    const unsigned maskx = (4u * (unsigned)g.m << g.log_ext) - 1u;       // 2 n ext - 1
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long be = t / per, e = t % per;
        const long long b = be >> g.log_ext;
        const int i = (int)(be & (ext - 1));
        const unsigned q = (unsigned)(e % g.m);
        const unsigned a = (unsigned)((unsigned long long)g.lwe[b * g.lwe_bs + 1 + g.idx] & (unsigned long long)maskx);
        const unsigned hi = a >> g.log_ext, lo = a & (unsigned)(ext - 1);
        int j;
        unsigned mult;
        bool skip;
        if (lo == 0) { j = i; mult = hi; skip = hi == 0; }
        else if ((unsigned)i < lo) { j = ext - (int)lo + i; mult = hi + 1u; skip = ((hi + 1u) & mask) == 0; }
        else { j = i - (int)lo; mult = hi; skip = hi == 0; }
        if (skip) continue;
        const cplx x = g.w2n[(mult * (4u * q + 1u)) & mask];
        const cplx vj = g.v[((b << g.log_ext) + j) * per + e];
        const cplx vi = g.v[be * per + e];
        cplx r = g.acc[be * per + e];
        const cplx xv = cmul(x, vj);
        r.x = (r.x + xv.x) - vi.x;
        r.y = (r.y + xv.y) - vi.y;
        g.acc[be * per + e] = r;
    }
}

// =================================================================================
// One block of the block-binary blind rotation in the DFT domain (algorithm.rs:319-337), fused:
//   acc_add[b][c] = sum_{i in block} (DFT(X^a_{b,i}) - 1) (.) ( sum_r acc_dft[b][r] (.) BRK_i[r][c] )
// i.e. vec_znx_dft_zero + block_size x (vmp_apply_dft_to_dft, svp_apply_dft_to_dft, dft_add_assign, dft_sub_assign)
// without vmp_res / vmp_xai ever leaving registers.  Thread = one frequency point q of CT ciphertexts: the key
// values of a point are loaded once and used by the CT ciphertexts; lanes run along q (contiguous in every operand).
// nrows <= MAXR (= dnum*cols rows actually present in acc_dft); the cols*brk_size outputs are split into groups of CG.
// =================================================================================
struct BrBlockArgs {
    const cplx* acc_dft;   // [batch][nrows_a][m]
    cplx* acc_add;         // [batch][ncols][m]
    long long a_bs, o_bs;  // points between ciphertexts
    const cplx* brk;       // prepared keys, key i at brk + i*key_stride; P[(r*ncols + c)*m + q]
    long long key_stride;
    int row_max, ncols, m, batch;
    int i0, blk;           // LWE coefficients i0 .. i0+blk-1
    const long long* lwe;  // [batch][n_lwe+1]
    long long lwe_bs;
    const cplx* w2n;
    int dbg;               // diagnostic (tools/dbg): bit 0 no key loads, bit 1 no products, bit 2 no accumulator loads, bit 3 no LDS staging
    int allcg;             // k_br_block_lds: one workgroup walks all gz column groups of its tile (accumulator tile read once)
    int gx, gy, gz, xcd;   // k_br_block_lds: logical grid (ciphertext tiles, 64-point slices, column groups) of the 1-D launch
};

// grid = (ceil(batch/CT), m/256, column groups of CG): the column group is uniform per workgroup, so the key row
// pointers stay in SGPRs and the CG x row_max loads of a coefficient are issued together (one exposed L2 latency per
// coefficient), as in the one-kernel path (device_br.hpp).
template <int CT, int MAXR, int CG>
__global__ void __launch_bounds__(256) k_br_block(BrBlockArgs g) {
    // ciphertext tiles fastest: the workgroups that run together share one 256-point slice of the block's keys (L2 resident)
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= g.m) return;
    const int b0 = blockIdx.x * CT;
    const int cg = blockIdx.z;
    const unsigned mask = 4u * (unsigned)g.m - 1u;
    const unsigned qf = 4u * (unsigned)q + 1u;
    cplx a[CT][MAXR];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int b = min(b0 + t, g.batch - 1);
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            a[t][r] = r < g.row_max ? g.acc_dft[(long long)b * g.a_bs + (long long)r * g.m + q] : make_double2(0.0, 0.0);
    }
    cplx out[CT][CG];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j) out[t][j] = make_double2(0.0, 0.0);
    // The rotation amounts of the block are loaded once (lane l holds coefficient i0 + l; blk <= 64) and DFT(X^a)[q] of
    // coefficient i+1 is fetched while coefficient i is multiplied: the lwe -> w2n chain is two dependent round trips that
    // would otherwise sit in front of every coefficient.
    unsigned aiv[CT];
    cplx xn[CT];
    {
        const int li = min((int)(threadIdx.x & 63), g.blk - 1);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int b = min(b0 + t, g.batch - 1);
            aiv[t] = (unsigned)((unsigned long long)g.lwe[(long long)b * g.lwe_bs + 1 + g.i0 + li] & (unsigned long long)mask);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t)
            xn[t] = g.w2n[((unsigned)__builtin_amdgcn_readlane((int)aiv[t], 0) * qf) & mask];
    }
    // (a second register set prefetching coefficient i+1 drops the occupancy to one wave per SIMD and is 15 % slower)
    for (int i = g.i0; i < g.i0 + g.blk; ++i) {
        const cplx* K = g.brk + (long long)i * g.key_stride;
        cplx kv[CG][MAXR];
#pragma unroll
        for (int j = 0; j < CG; ++j) {
            const int c = min(cg * CG + j, g.ncols - 1);
#pragma unroll
            for (int r = 0; r < MAXR; ++r) kv[j][r] = (K + (long long)(min(r, g.row_max - 1) * g.ncols + c) * g.m)[q];
        }
        cplx xm[CT];   // DFT(X^a)[q] for each ciphertext of the tile
        {
            const int nx = min(i + 1 - g.i0, g.blk - 1);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                xm[t] = xn[t];
                xm[t].x -= 1.0;   // d = DFT(X^a)[q] - 1
                xn[t] = g.w2n[((unsigned)__builtin_amdgcn_readlane((int)aiv[t], nx) * qf) & mask];
            }
        }
#pragma unroll
        for (int j = 0; j < CG; ++j) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                cplx s = make_double2(0.0, 0.0);
#pragma unroll
                for (int r = 0; r < MAXR; ++r) {
                    if (r < g.row_max) {
                        s.x = __builtin_fma(a[t][r].x, kv[j][r].x, s.x);
                        s.x = __builtin_fma(-a[t][r].y, kv[j][r].y, s.x);
                        s.y = __builtin_fma(a[t][r].x, kv[j][r].y, s.y);
                        s.y = __builtin_fma(a[t][r].y, kv[j][r].x, s.y);
                    }
                }
                // (DFT(X^a) - 1) s as four FMAs on the difference d = DFT(X^a) - 1 (round 5) instead of x s, + out, - s: eight operations
                // per (ciphertext, column, coefficient) beside the 4 * rows FMAs of the product
                out[t][j].x = __builtin_fma(xm[t].x, s.x, out[t][j].x);
                out[t][j].x = __builtin_fma(-xm[t].y, s.y, out[t][j].x);
                out[t][j].y = __builtin_fma(xm[t].x, s.y, out[t][j].y);
                out[t][j].y = __builtin_fma(xm[t].y, s.x, out[t][j].y);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int b = b0 + t;
        if (b < g.batch) {
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const int c = cg * CG + j;
                if (c < g.ncols) g.acc_add[(long long)b * g.o_bs + (long long)c * g.m + q] = out[t][j];
            }
        }
    }
}

// Same block step for the shapes whose keys no longer stay L2-resident per ciphertext pair (rank 2 with 3-4 decomposition
// rows: 9-12 input polynomials, 12-16 output columns): a workgroup is 4 waves x CT ciphertexts at 64 spectrum points, and the
// CG x row_max key values of a coefficient are staged once per workgroup in LDS (double-buffered: the loads of coefficient
// i+1 are in flight while coefficient i is multiplied), so every key value is fetched from L2 once per 4*CT ciphertexts
// instead of once per 2.  The per-(ciphertext, column) FMA chains are the ones of k_br_block, in the same order.
// Logical grid = (gx = ceil(batch / (4*CT)), gy = m/64, gz = column groups of CG), launched 1-D; requires m % 64 == 0.
// The gz workgroups that read the same accumulator tile run back to back on ONE XCD (workgroup ids go round-robin over
// the 8 XCDs), so the tile comes from HBM once and from that XCD's L2 afterwards, and the tiles an XCD works on at any
// time share their 64-point key slice.
template <int MAXR, int CG, int PER>
__device__ __forceinline__ void brl_fetch(cplx (&nxt)[PER], const BrBlockArgs& g, int i, int w, int cg, int q) {
    const cplx* K = g.brk + (long long)i * g.key_stride;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = min(u * 4 + w, CG * MAXR - 1);
        const int j = e / MAXR, r = e % MAXR;
        const int c = min(cg * CG + j, g.ncols - 1);
        nxt[u] = (PZ_DBG(g.dbg) & 1) ? make_double2(1.0, (double)e) : (K + (long long)(min(r, g.row_max - 1) * g.ncols + c) * g.m)[q];
    }
}
template <int PER, int NE>
__device__ __forceinline__ void brl_stage(const cplx (&nxt)[PER], cplx (*ks)[64], int w, int lane) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = u * 4 + w;
        if (e < NE) ks[e][lane] = nxt[u];
    }
}
#ifndef PZ_BRL_STAMP
#define PZ_BRL_STAMP 0   // diagnostic build: per-phase s_memtime totals of k_br_block_lds, printed by a few waves (tools/dbg/brl_stamps.sh)
#endif
// (Round 5, measured and dropped, both on the evidence of the stamps' 0.5 - 2.3 k cycles per stage in front of the LDS writes
//  (profiles/r05_brl_stamps.txt): (1) the stage's key values L2 -> LDS by global_load_lds instead of through registers + ds_write - 20 - 28
//  registers fewer, no ds_write, and 5 - 8 % SLOWER, 8.10 -> 8.50 ms per 82 block steps at N = 2048, 5.38 -> 5.81 for the rank-2 shape: an
//  LDS-DMA piece costs the issuing wave far more than a global_load + ds_write pair here (profiles/r05_ab_brl_dma.txt); (2) the key values
//  requested TWO stages ahead, two register sets in ping-pong with the stage loop unrolled by two: 5.38 -> 5.48 ms for the rank-2 shape (no
//  spill at 240 registers), 8.07 -> 8.47 for the 6-row shape (at two workgroups per CU instead of three) - profiles/r05_ab_brl_pf2.txt.
//  The wait the stamps show in that phase is not the requests' latency.)
template <int CT, int MAXR, int CG>
__global__ void __launch_bounds__(256, (MAXR == 6 ? 3 : 2)) k_br_block_lds(BrBlockArgs g) {   // 6 rows x 3 columns: three workgroups per CU (168 registers)
    constexpr int NW = 4, NE = CG * MAXR, PER = (NE + NW - 1) / NW;
    __shared__ cplx ks[2][NE][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#if PZ_BRL_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_t;
#define PZ_BSTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define PZ_BSTAMP(i)
#endif
    int tile, cg0;
    if (g.allcg) {
        tile = blockIdx.x;
        cg0 = 0;
    } else if (g.xcd) {
        const int L = blockIdx.x, k = L >> 3;
        cg0 = k % g.gz;
        tile = (k / g.gz) * 8 + (L & 7);
    } else {
        cg0 = blockIdx.x % g.gz;
        tile = blockIdx.x / g.gz;
    }
    const int q = (tile / g.gx) * 64 + lane;
    const int b0 = ((tile % g.gx) * NW + w) * CT;
    const unsigned mask = 4u * (unsigned)g.m - 1u;
    const unsigned qf = 4u * (unsigned)q + 1u;
    cplx a[CT][MAXR];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int b = min(b0 + t, g.batch - 1);
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            a[t][r] = (r < g.row_max && !(PZ_DBG(g.dbg) & 4)) ? g.acc_dft[(long long)b * g.a_bs + (long long)r * g.m + q] : make_double2(0.0, (double)b);
    }
    cplx out[CT][CG];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int j = 0; j < CG; ++j) out[t][j] = make_double2(0.0, 0.0);
    cplx nxt[PER];
    brl_fetch<MAXR, CG, PER>(nxt, g, g.i0, w, cg0, q);
    // rotation amounts of the block: lane l holds coefficient i0 + l (blk <= 64); DFT(X^a)[q] is fetched one coefficient ahead
    unsigned aiv[CT];
    cplx xn[CT];
    {
        const int li = min(lane, g.blk - 1);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int b = min(b0 + t, g.batch - 1);
            aiv[t] = (unsigned)((unsigned long long)g.lwe[(long long)b * g.lwe_bs + 1 + g.i0 + li] & (unsigned long long)mask);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t)
            xn[t] = g.w2n[((unsigned)__builtin_amdgcn_readlane((int)aiv[t], 0) * qf) & mask];
    }
    brl_stage<PER, NE>(nxt, ks[0], w, lane);
    __syncthreads();
#if PZ_BRL_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the accumulator tile has arrived: the prologue's share of the wave's life)
#endif
    PZ_BSTAMP(0)
    // stages = (column group, coefficient) pairs in order; the loads of stage s+1 are in flight while stage s is multiplied
    const int nst = (g.allcg ? g.gz : 1) * g.blk;
    int is = 0, cgs = cg0;
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        int is_n = is + 1, cg_n = cgs;
        if (is_n == g.blk) { is_n = 0; cg_n = cgs + 1; }
        if (s + 1 >= nst) { is_n = is; cg_n = cgs; }  // (unconditional fetch: a guarded array stays in scratch)
        brl_fetch<MAXR, CG, PER>(nxt, g, g.i0 + is_n, w, cg_n, q);
        cplx xm[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            xm[t] = xn[t];
            xm[t].x -= 1.0;   // d = DFT(X^a)[q] - 1 (k_br_block)
            xn[t] = g.w2n[((unsigned)__builtin_amdgcn_readlane((int)aiv[t], is_n) * qf) & mask];
        }
        cplx sacc[CT][CG];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int j = 0; j < CG; ++j) sacc[t][j] = make_double2(0.0, 0.0);
        // (rows beyond row_max carry a = 0 and a clamped key value: with row_max == MAXR - the usual shapes - the per-row test, a branch per
        //  row inside the unrolled loop, is dropped; round 3, as in k_br_fused)
#define PZ_BRL_FMAS(GUARD_)                                                                       \
    _Pragma("unroll") for (int r = 0; r < MAXR; ++r) {                                           \
        if ((!(GUARD_) || r < g.row_max) && !(PZ_DBG(g.dbg) & 2)) {                              \
            _Pragma("unroll") for (int j = 0; j < CG; ++j) {                                     \
                const cplx kv = ks[buf][j * MAXR + r][lane];                                     \
                _Pragma("unroll") for (int t = 0; t < CT; ++t) {                                 \
                    cplx& sv = sacc[t][j];                                                       \
                    sv.x = __builtin_fma(a[t][r].x, kv.x, sv.x);                                 \
                    sv.x = __builtin_fma(-a[t][r].y, kv.y, sv.x);                                \
                    sv.y = __builtin_fma(a[t][r].x, kv.y, sv.y);                                 \
                    sv.y = __builtin_fma(a[t][r].y, kv.x, sv.y);                                 \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    }
        // every row in use (the usual shapes): the staged key value of step (r, j) + 1 is read from LDS in front of step (r, j)'s FMAs - written as
        // above the compiler put most reads directly in front of their eight FMAs behind lgkmcnt(0) (round 5 ISA).  Same chains, same order.
#define PZ_BRL_FMAS_PIPE                                                                          \
    if (!(PZ_DBG(g.dbg) & 2)) {                                                                  \
        cplx kcur = ks[buf][0][lane], knxt;                                                      \
        _Pragma("unroll") for (int r = 0; r < MAXR; ++r) {                                       \
            _Pragma("unroll") for (int j = 0; j < CG; ++j) {                                     \
                if (!(r == MAXR - 1 && j == CG - 1)) knxt = ks[buf][(j + 1 < CG ? (j + 1) * MAXR + r : r + 1)][lane]; \
                _Pragma("unroll") for (int t = 0; t < CT; ++t) {                                 \
                    cplx& sv = sacc[t][j];                                                       \
                    sv.x = __builtin_fma(a[t][r].x, kcur.x, sv.x);                               \
                    sv.x = __builtin_fma(-a[t][r].y, kcur.y, sv.x);                              \
                    sv.y = __builtin_fma(a[t][r].x, kcur.y, sv.y);                               \
                    sv.y = __builtin_fma(a[t][r].y, kcur.x, sv.y);                               \
                }                                                                                \
                kcur = knxt;                                                                     \
            }                                                                                    \
        }                                                                                        \
    }
        PZ_BSTAMP(1)   // fetch issue + monomial gather issue
        if (g.row_max == MAXR) { PZ_BRL_FMAS_PIPE } else { PZ_BRL_FMAS(true) }
#undef PZ_BRL_FMAS
#undef PZ_BRL_FMAS_PIPE
        PZ_BSTAMP(2)   // LDS reads + FMA chains
#pragma unroll
        for (int j = 0; j < CG; ++j)
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                out[t][j].x = __builtin_fma(xm[t].x, sacc[t][j].x, out[t][j].x);
                out[t][j].x = __builtin_fma(-xm[t].y, sacc[t][j].y, out[t][j].x);
                out[t][j].y = __builtin_fma(xm[t].x, sacc[t][j].y, out[t][j].y);
                out[t][j].y = __builtin_fma(xm[t].y, sacc[t][j].x, out[t][j].y);
            }
        PZ_BSTAMP(3)   // monomial factor (waits for its gather)
        if (is == g.blk - 1) {  // last coefficient of this column group: store and restart the accumulators
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int b = b0 + t;
#pragma unroll
                for (int j = 0; j < CG; ++j) {
                    const int c = cgs * CG + j;
                    if (b < g.batch && c < g.ncols) g.acc_add[(long long)b * g.o_bs + (long long)c * g.m + q] = out[t][j];
                    out[t][j] = make_double2(0.0, 0.0);
                }
            }
        }
        PZ_BSTAMP(4)   // result stores (issue)
        if (!(PZ_DBG(g.dbg) & 8)) brl_stage<PER, NE>(nxt, ks[buf ^ 1], w, lane);
        PZ_BSTAMP(5)   // wait for the next stage's key values + LDS writes
        __syncthreads();
        PZ_BSTAMP(6)   // barrier
        is = is_n;
        cgs = cg_n;
    }
#if PZ_BRL_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PZ_BSTAMP(7)       // stores acknowledged
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1) && g.i0 == g.blk)
        printf("BSTAMP wg %d wave %d total %llu | prologue %llu | per stage (%d stages): issue %llu fma %llu monomial %llu stores %llu keywait+stage %llu barrier %llu | storeack %llu\n",
               (int)blockIdx.x, w, st_t - st_t0, st_acc[0], nst, st_acc[1] / nst, st_acc[2] / nst, st_acc[3] / nst, st_acc[4] / nst, st_acc[5] / nst, st_acc[6] / nst, st_acc[7]);
#endif
#undef PZ_BSTAMP
}


}  // namespace pz
