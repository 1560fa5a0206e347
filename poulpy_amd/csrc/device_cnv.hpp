// device_cnv.hpp — bivariate convolution over Z[X, Y]/(X^N + 1), Y = 2^-base2k (poulpy-hal api/convolution.rs; reference
// poulpy-cpu-ref/src/reference/fft64/convolution.rs), the arithmetic of GLWE tensoring (CKKS multiplication).
//
// Prepared operands (CnvPVecL / CnvPVecR, ScalarPrep = f64, backend-private bytes of the reference's size n*cols*size*8): the
// spectra of the limbs in device order, polynomial (col, limb) at (col*size + limb)*m points — the reference interleaves 4-point
// blocks of all limbs instead (convolution.rs:66-72); the layout is opaque to callers.
#pragma once
#include "device_fft.hpp"

namespace pz {

struct CnvArgs {
    double* res;              // VecZnxDft (res_cols columns), limb kk of column res_col at m*(kk*res_cols + res_col) points
    const cplx* a;
    const cplx* b;
    long long res_bs, a_bs, b_bs;   // batch strides in POINTS
    int res_cols, res_col, min_size, offset;
    int a_size, a_i, a_j;     // a_j < 0: plain; else the operand is a[a_i] + a[a_j]  (convolution.rs:323-324)
    int b_size, b_i, b_j;
    int m, batch;
};

// grid = (ceil(m/256), min_size, batch): one thread per frequency point of one output limb; lanes run along q (contiguous in
// every operand); the a_size + b_size input spectra of a point are re-read per output limb from L2.
__global__ void __launch_bounds__(256) k_cnv_apply(CnvArgs g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.m) return;
    const int kk = blockIdx.y, bt = blockIdx.z;
    const int k = kk + g.offset;
    cplx acc = make_double2(0.0, 0.0);
    if (k < g.a_size + g.b_size) {   // reim4/arithmetic_ref.rs:235-247
        const int j_min = k >= g.a_size - 1 ? k - (g.a_size - 1) : 0;
        const int j_max = min(k + 1, g.b_size);
        const cplx* a0 = g.a + (long long)bt * g.a_bs + (long long)g.a_i * g.a_size * g.m + q;
        const cplx* b0 = g.b + (long long)bt * g.b_bs + (long long)g.b_i * g.b_size * g.m + q;
        const cplx* a1 = g.a_j >= 0 ? g.a + (long long)bt * g.a_bs + (long long)g.a_j * g.a_size * g.m + q : nullptr;
        const cplx* b1 = g.b_j >= 0 ? g.b + (long long)bt * g.b_bs + (long long)g.b_j * g.b_size * g.m + q : nullptr;
        for (int j = j_min; j < j_max; ++j) {
            cplx av = a0[(long long)(k - j) * g.m];
            cplx bv = b0[(long long)j * g.m];
            if (a1) av = cadd(av, a1[(long long)(k - j) * g.m]);
            if (b1) bv = cadd(bv, b1[(long long)j * g.m]);
            acc.x = __builtin_fma(av.x, bv.x, acc.x);
            acc.x = __builtin_fma(-av.y, bv.y, acc.x);
            acc.y = __builtin_fma(av.x, bv.y, acc.y);
            acc.y = __builtin_fma(av.y, bv.x, acc.y);
        }
    }
    cplx* out = reinterpret_cast<cplx*>(g.res) + (long long)bt * g.res_bs + (long long)g.m * ((long long)kk * g.res_cols + g.res_col) + q;
    *out = acc;
}

// k_cnv_apply_lds (round 3): the same sums with the a_size + b_size operand limbs of 128 frequency points staged once in LDS (pairwise
// sums a[i] + a[j], b[i] + b[j] formed while staging) and all min_size output limbs computed from there - every operand value is read from
// global memory once per term instead of once per output limb it reaches (k_cnv_apply: two L2-served loads per multiply-add, 8 KiB per
// point and term at 16 x 16 limbs: 3.1 ms per term and 256 ciphertext pairs at N = 2^16).  grid = (m / 128, batch), 256 threads: thread
// = (point, output-limb parity).  Same order of the FMAs per output as k_cnv_apply (same bits).  Needs m % 128 == 0, a_size + b_size <= 64.
__global__ void __launch_bounds__(256) k_cnv_apply_lds(CnvArgs g) {
    extern __shared__ cplx lds[];   // [a_size + b_size][128]
    const int tid = threadIdx.x, pt = tid & 127, par = tid >> 7;
    const int q0 = blockIdx.x * 128, bt = blockIdx.y;
    const cplx* a0 = g.a + (long long)bt * g.a_bs + (long long)g.a_i * g.a_size * g.m + q0;
    const cplx* b0 = g.b + (long long)bt * g.b_bs + (long long)g.b_i * g.b_size * g.m + q0;
    const cplx* a1 = g.a_j >= 0 ? g.a + (long long)bt * g.a_bs + (long long)g.a_j * g.a_size * g.m + q0 : nullptr;
    const cplx* b1 = g.b_j >= 0 ? g.b + (long long)bt * g.b_bs + (long long)g.b_j * g.b_size * g.m + q0 : nullptr;
    cplx* la = lds;
    cplx* lb = lds + g.a_size * 128;
    for (int l = par; l < g.a_size; l += 2) {
        cplx v = a0[(long long)l * g.m + pt];
        if (a1) v = cadd(v, a1[(long long)l * g.m + pt]);
        la[l * 128 + pt] = v;
    }
    for (int l = par; l < g.b_size; l += 2) {
        cplx v = b0[(long long)l * g.m + pt];
        if (b1) v = cadd(v, b1[(long long)l * g.m + pt]);
        lb[l * 128 + pt] = v;
    }
    __syncthreads();
    for (int kk = par; kk < g.min_size; kk += 2) {
        const int k = kk + g.offset;
        cplx acc = make_double2(0.0, 0.0);
        if (k < g.a_size + g.b_size) {
            const int j_min = k >= g.a_size - 1 ? k - (g.a_size - 1) : 0;
            const int j_max = min(k + 1, g.b_size);
            for (int j = j_min; j < j_max; ++j) {
                const cplx av = la[(k - j) * 128 + pt], bv = lb[j * 128 + pt];
                acc.x = __builtin_fma(av.x, bv.x, acc.x);
                acc.x = __builtin_fma(-av.y, bv.y, acc.x);
                acc.y = __builtin_fma(av.x, bv.y, acc.y);
                acc.y = __builtin_fma(av.y, bv.x, acc.y);
            }
        }
        cplx* out = reinterpret_cast<cplx*>(g.res) + (long long)bt * g.res_bs + (long long)g.m * ((long long)kk * g.res_cols + g.res_col) + q0 + pt;
        *out = acc;
    }
}

// k_mid_cnv (round 3): one term of a GLWE tensoring on the row-major pipeline layout (m = m1 x 128 plans) - the convolution's counterpart of
// k_mid128.  Tile = one frequency row q1 of one ciphertext pair: the a_size + b_size operand rows T'[limb][q1][0..127] (pairwise terms: the
// sums a[i] + a[j], b[i] + b[j], formed while loading - the transforms are linear), forward row DFT in LDS (8 lanes per row, as
// k_mid128), the limb convolution point by point (as k_cnv_apply_lds), inverse row DFT of the min_size result rows x conj tw12 -> T2'.
// Replaces, per term: forward pass 2 of both operands, k_cnv_apply, inverse pass 2 (three HBM round trips of the spectra).
// 256 threads; LDS max(a_size + b_size, min_size) rows x 144 points + 2 x 128; min_size <= 32.
// (Measured and dropped: eight consecutive frequency rows per workgroup with the next row's operand loads in flight behind the current row's
//  forward transform - 255 registers, one wave per SIMD, 8.5 vs 7.2 ms per 256 pairs.)
constexpr int kMidCnvRS = 144;   // row stride of the tile, as k_mid128 (z[k1][o] at k1 * 9 + o)
struct MidCnvArgs {
    const cplx *a_main, *a_last, *b_main, *b_last;   // T' of the operand limbs: main [pair][limb < size - 1][col][m], last (masked limb) [pair][col][m]
    cplx* T2;                                         // [pair][kk < min_size][m], rows [q1][128]
    int cols, a_size, b_size, a_i, a_j, b_i, b_j;     // a_j / b_j < 0: plain term
    int min_size, offset, m1, batch;
    const cplx* wL2;
    const cplx* tw12t;
};
__global__ void __launch_bounds__(256) k_mid_cnv(MidCnvArgs g) {
    constexpr int M2 = 128, RS = kMidCnvRS;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x, row = tid >> 3, o = tid & 7;
    const int q1 = blockIdx.x % g.m1, bt = blockIdx.x / g.m1;
    const long long m = (long long)g.m1 * M2;
    const int A = g.a_size + g.b_size;
    cplx* wl = lds + max(A, g.min_size) * RS;
    cplx* twrow = wl + M2;
    if (tid < M2) { wl[tid] = g.wL2[tid]; twrow[tid] = g.tw12t[(long long)q1 * M2 + tid]; }
    __syncthreads();
    // ---- forward row DFT of the operand rows, 32 rows per sweep ----
    for (int r0 = 0; r0 < A; r0 += 32) {
        const int r = r0 + row;
        if (r < A) {
            const bool isb = r >= g.a_size;
            const int limb = isb ? r - g.a_size : r, size = isb ? g.b_size : g.a_size, ci = isb ? g.b_i : g.a_i, cj = isb ? g.b_j : g.a_j;
            const cplx* mainp = isb ? g.b_main : g.a_main;
            const cplx* lastp = isb ? g.b_last : g.a_last;
            auto rowptr = [&](int col) {
                return (limb < size - 1 ? mainp + (((long long)bt * (size - 1) + limb) * g.cols + col) * m : lastp + ((long long)bt * g.cols + col) * m) +
                       (long long)q1 * M2 + o;
            };
            cplx x[16];
            {
                const cplx* s0 = rowptr(ci);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) x[n1] = ld_stream(s0 + 8 * n1);
                if (cj >= 0) {
                    const cplx* s1 = rowptr(cj);
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) x[n1] = cadd(x[n1], ld_stream(s1 + 8 * n1));
                }
            }
            cplx* rowbuf = lds + r * RS;
            Bfly<16, false>::run(x);
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                cplx v = x[k1];
                if (k1 > 0) v = cmul(v, wl[o * k1]);
                rowbuf[k1 * 9 + o] = v;
            }
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) x[8 * h + oo] = rowbuf[(o + 8 * h) * 9 + oo];
            Bfly<8, false>::run(x);
            Bfly<8, false>::run(x + 8);
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 8 * h + 16 * k2] = x[8 * h + k2];
        }
    }
    __syncthreads();
    // ---- convolution over the limbs, point by point: thread = (point, output-limb parity); same sums as k_cnv_apply.  The results wait in
    // registers (<= 16 per thread: min_size <= 32) until every operand value has been read, then take the place of the first operand
    // rows: the tile is max(a_size + b_size, min_size) rows, two workgroups per CU at 16 + 16 limbs ----
    {
        const int pt = tid & 127, par = tid >> 7;
        const cplx* la = lds;
        const cplx* lb = lds + g.a_size * RS;
        cplx accs[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int kk = par + 2 * u;
            cplx acc = make_double2(0.0, 0.0);
            if (kk < g.min_size) {
                const int k = kk + g.offset;
                if (k < g.a_size + g.b_size) {
                    const int j_min = k >= g.a_size - 1 ? k - (g.a_size - 1) : 0;
                    const int j_max = min(k + 1, g.b_size);
                    for (int j = j_min; j < j_max; ++j) {
                        const cplx av = la[(k - j) * RS + pt], bv = lb[j * RS + pt];
                        acc.x = __builtin_fma(av.x, bv.x, acc.x);
                        acc.x = __builtin_fma(-av.y, bv.y, acc.x);
                        acc.y = __builtin_fma(av.x, bv.y, acc.y);
                        acc.y = __builtin_fma(av.y, bv.x, acc.y);
                    }
                }
            }
            accs[u] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int kk = par + 2 * u;
            if (kk < g.min_size) lds[kk * RS + pt] = accs[u];
        }
    }
    __syncthreads();
    // ---- inverse row DFT of the result rows, x conj tw12 -> T2' ----
    for (int r0 = 0; r0 < g.min_size; r0 += 32) {
        const int r = r0 + row;
        if (r < g.min_size) {
            cplx* rowbuf = lds + r * RS;
            cplx u[16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) u[8 * h + k2] = rowbuf[o + 8 * h + 16 * k2];
            Bfly<8, true>::run(u);
            Bfly<8, true>::run(u + 8);
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) {
                    cplx v = u[8 * h + oo];
                    const int k1 = o + 8 * h;
                    if (k1 > 0 && oo > 0) v = cmulc(v, wl[oo * k1]);
                    rowbuf[k1 * 9 + oo] = v;
                }
            row_sync();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 9 + o];
            Bfly<16, true>::run(u);
            cplx* dst = g.T2 + ((long long)bt * g.min_size + r) * m + (long long)q1 * M2 + o;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) st_stream(dst + 8 * n1, cmulc(u[n1], twrow[o + 8 * n1]));
        }
    }
}

// k_mid_cnv3 (round 4): ALL THREE terms of a rank-1 tensoring per tile - (0,0) = a0 b0, (1,1) = a1 b1 and the pairwise (0,1) = (a0 + a1)(b0 + b1)
// (operations/glwe.rs:609-913: the Karatsuba form the reference evaluates; each term is inverse-transformed and normalized on its own, so
// three result sets leave the kernel).  Why: profiles/r03_roofline.md had the tensoring at 0.12 of HBM; per term k_mid_cnv re-loaded and
// re-transformed the operand rows (128 row loads / 96 forward row DFTs per tile-triple where 64 / 64 suffice) and its convolution read two
// operand values from LDS per multiply-add (206 GB of LDS reads per 256 pairs: ~3 ms at the LDS rate - the phase was LDS-bound).  Here:
//   * tile = one frequency row q1 of one pair, 2 AS + 2 BS operand rows (a0, a1, b0, b1) loaded and forward-transformed ONCE (512 threads =
//     64 rows x 8 lanes, one sweep, as k_mid128);
//   * the limb convolution keeps one operand vector in REGISTERS: thread = (point, group); groups 0 / 1 = the diagonal terms (A vector of AS
//     values in registers, B streamed, all AS + BS - 1 product limbs accumulated in registers, statically unrolled), groups 2 / 3 = the even /
//     odd product limbs of the pairwise term (the sums a0 + a1, b0 + b1 formed from LDS on the way).  2 LDS reads per AS multiply-adds instead
//     of 2 per 1.  The window [offset, offset + min_size) of product limbs is selected when the accumulators are written back (static register
//     index, dynamic LDS row), limbs beyond the product are zero rows;
//   * waves 0-1 / 2-3 / 4-5 / 6-7 = groups 0 / 1 / 2 / 3: every SIMD carries one diagonal wave and one half-pairwise wave (balanced);
//   * inverse row DFT of the 3 x min_size result rows x conj tw12 -> T2'[term][pair][limb].
// AS, BS compile-time (the register arrays must not be indexed dynamically): instantiated for the 16- and 8-limb shapes; other shapes keep
// k_mid_cnv per term.
struct MidCnv3Args {
    const cplx *a_main, *a_last, *b_main, *b_last;   // as MidCnvArgs (cols = 2)
    cplx* T2;                                         // [term < 3][pair][kk < min_size][m]
    int min_size, offset, m1, batch;
    const cplx* wL2;
    const cplx* tw12t;
};
#ifndef PZ_CNV_STAMP
#define PZ_CNV_STAMP 0   // diagnostic build: per-phase s_memtime totals of k_mid_cnv3, printed by a few waves (tools/dbg/cnv_stamps.sh)
#endif
#ifndef PZ_CNV_PRIO
#define PZ_CNV_PRIO 0    // A/B: static wave priority (1 - 3) for the half-pairwise waves during the convolution
#endif
#ifndef PZ_CNV_EARLY
#define PZ_CNV_EARLY 1   // static-window forms: the next tile's operand loads travel under the convolution (0: under the inverse row pass, as the WLO = 0 forms)
#endif
#ifndef PZ_CNV_SPREAD
#define PZ_CNV_SPREAD 1  // the next tile's operand loads go out in four groups between the steps of the inverse row pass (0: all 16 in front of it, rounds 4 - 5)
#endif
// acc += x * y (four FMAs; the order of the reference's reim4 kernels is not needed: only the rounded integers are compared)
#define PZ_CMAC(ACC_, X_, Y_)                                    \
    {                                                            \
        (ACC_).x = __builtin_fma((X_).x, (Y_).x, (ACC_).x);      \
        (ACC_).x = __builtin_fma(-(X_).y, (Y_).y, (ACC_).x);     \
        (ACC_).y = __builtin_fma((X_).x, (Y_).y, (ACC_).y);      \
        (ACC_).y = __builtin_fma((X_).y, (Y_).x, (ACC_).y);      \
    }
// Round 6 (stamps: profiles/r06_cnv_stamps.txt - per tile of 30 k cycles the limb convolution takes 8 k, VALU-bound: 6.1 k cycles of FMA issue per
// SIMD; issuing the next tile's 16 loads 2.6 - 5 k):
//   * WLO (compile time): only the product limbs k in [offset, offset + min_size) leave the kernel (convolution.rs:235-247: res limb kk = product
//     limb kk + offset), yet every one of the AS x BS multiply-adds ran - at the configs[4] shape (offset 13, 17 limbs kept of 31) 36 % of them fed
//     accumulators that were dropped at write-back.  The multiply-adds with i + j < WLO are not compiled in (and their accumulators do not
//     exist); the launcher picks the largest instantiated WLO <= offset (0 or AS - 4).  Tested per block of 4 at run time against the offset the
//     same skip measured SLOWER than none (conv 5.2 -> 8.0 k cycles: a branch per block keeps the operand reads from travelling ahead);
//   * SQ (glwe_tensor_square_apply, b = a): 2 AS operand rows instead of 4 AS (half the row loads and forward row transforms), and the three
//     products are squares - sum_{i + j = k} a_i a_j = 2 sum_{i < j} a_i a_j + [k even] a_{k/2}^2: 136 multiply-adds per diagonal term
//     instead of 256, no operand streamed from LDS at all (both factors live in the thread's registers): 3.55 -> 2.55 ms per 256 squares;
//   * the next tile's operand loads in four groups between the steps of the inverse row pass (as k_mid128r), LDS-only barriers.
template <int AS, int BS, bool SQ = false, int WLO = 0>
__global__ void __launch_bounds__(512) k_mid_cnv3(MidCnv3Args g) {
    static_assert(!SQ || AS == BS, "square form: one operand");
    static_assert(BS == 16 || BS == 8, "k_mid_cnv3: the early operand loads are written for 16 / 8 limbs");
    constexpr int M2 = 128, RS = kMidCnvRS, NR = SQ ? 2 * AS : 2 * AS + 2 * BS, NK = AS + BS - 1;
    static_assert(NR <= 64, "k_mid_cnv3: at most 64 operand rows");
    static_assert(WLO >= 0 && WLO < NK, "k_mid_cnv3: the static window starts inside the product");
    constexpr bool EARLY = PZ_CNV_EARLY && WLO > 0 && !SQ;   // (square form: measured slower with the early loads, 2.62 vs 2.48 ms per 256 - half the rows load at all)
    extern __shared__ cplx lds[];   // 64 rows x RS | wL2[128] | tw12t rows [2][128]
    const int tid = threadIdx.x;
    const long long m = (long long)g.m1 * M2;
    cplx* wl = lds + 64 * RS;
    cplx* twrow2 = wl + M2;
    // PERSISTENT (one workgroup per CU: the tile fills LDS): tile L = (pair bt, frequency row q1), walked with stride gridDim.x; the next
    // tile's operand rows are requested during the inverse row pass of the current one and travel under it (the first version, one tile per
    // workgroup, left the CU idle while its only workgroup waited for its loads: 3.3 TB/s, profiles/r04_tensor_*).
    const long long ntiles = (long long)g.batch * g.m1;
    // Lane coordinates are re-derived from an OPAQUE copy of the thread index in every phase (as in k_mid128r): derived once, everything that
    // depends on them - row pointers, LDS offsets of the exchange passes, the output address - is hoisted out of the tile loop and sits beside the
    // 188 registers of the convolution (300 bytes of scratch).
    // the operand row a thread's 8-lane group loads and transforms: (operand, column, limb) - the same for every tile
    auto row_src = [&](long long L, int row, int o) {
        const bool isb = !SQ && row >= 2 * AS;
        const int rr_ = isb ? row - 2 * AS : row, size_ = isb ? BS : AS;
        const int col_ = rr_ / size_, limb_ = rr_ % size_;
        const long long Lc = L < ntiles ? L : ntiles - 1;
        const int q1 = (int)(Lc % g.m1);
        const long long bt = Lc / g.m1;
        return (limb_ < size_ - 1 ? (isb ? g.b_main : g.a_main) + ((bt * (size_ - 1) + limb_) * 2 + col_) * m
                                  : (isb ? g.b_last : g.a_last) + (bt * 2 + col_) * m) + (long long)q1 * M2 + o;
    };
    cplx x[16];
    long long L = blockIdx.x;
    if (L >= ntiles) return;
    {
        const int row = tid >> 3, o = tid & 7;
        if (row < NR) {
            const cplx* src = row_src(L, row, o);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) x[n1] = ld_stream(src + 8 * n1);
        }
    }
    if (tid < M2) { wl[tid] = g.wL2[tid]; twrow2[tid] = g.tw12t[(long long)(L % g.m1) * M2 + tid]; }
    __syncthreads();
#if PZ_CNV_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_t;
    int st_tiles = 0;
#define PZ_CSTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define PZ_CSTAMP(i)
#endif
    int par = 0;
    for (; L < ntiles; L += gridDim.x, par ^= 1) {
        const int q1 = (int)(L % g.m1);
        const long long bt = L / g.m1;
        // ---- forward row DFT of the operand rows (see k_mid128) ----
        const int tf = pz_opaque(tid);
        if (NR == 64 || (tf >> 3) < NR) {
            const int row = tf >> 3, o = tf & 7;
            cplx* rowbuf = lds + row * RS;
#if PZ_CNV_STAMP
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PZ_CSTAMP(0)   // wait for the operand rows
#endif
            Bfly<16, false>::run(x);
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                cplx v = x[k1];
                if (k1 > 0) v = cmul(v, wl[o * k1]);
                rowbuf[k1 * 9 + o] = v;
            }
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) x[8 * h + oo] = rowbuf[(o + 8 * h) * 9 + oo];
            Bfly<8, false>::run(x);
            Bfly<8, false>::run(x + 8);
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 8 * h + 16 * k2] = x[8 * h + k2];
        }
        PZ_CSTAMP(1)   // forward row transforms
        lds_barrier();   // (LDS ordering only: the previous tile's stores keep draining)
        PZ_CSTAMP(2)   // barrier
        // ---- limb convolution, one operand vector in registers ----
        {
            const int tc = pz_opaque(tid);
            const int pt = tc & 127, grp = __builtin_amdgcn_readfirstlane(tc >> 7);   // wave-uniform group, in a scalar register: the branches on it below are
                                                                                         // scalar branches (as a vector value both sides of every `if` were emitted under exec masks, and the
                                                                                         // second side waited for the first side's loads into the same registers)
            const cplx* A0 = lds + pt;
            const cplx* A1 = A0 + AS * RS;
            const cplx* B0 = A0 + 2 * AS * RS;        // (!SQ)
            const cplx* B1 = B0 + BS * RS;
            const int term = grp < 2 ? grp : 2;
            cplx* out = lds + (term * g.min_size) * RS + pt;   // result rows [term][kk]: written only after every operand value has been read
            constexpr int NKH = (NK + 1) / 2;
            const int hp = grp - 2;    // groups 2 / 3: this thread's product limbs are k = 2 u + hp
#if PZ_CNV_PRIO
            // the half-pairwise wave of a SIMD has half the multiply-adds of the diagonal wave beside it but starts later (32 + 32 operand reads and their sums) and,
            // dispatched second, loses the arbitration: with static priority it finishes first and the diagonal wave fills its gaps
            if (grp >= 2) __builtin_amdgcn_s_setprio(PZ_CNV_PRIO);
#endif
            cplx av[AS], acc[NK];      // (groups 2 / 3 use acc[0 .. NKH); product limbs below WLO are never touched: no registers)
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[k] = make_double2(0.0, 0.0);
            // EARLY (the forms with a static window: 163 registers instead of 210): the next tile's 16 operand loads go out one (two at 8 limbs)
            // per step of the j loop below - the convolution is VALU-bound and leaves the vector-memory path idle, while issued in front of the
            // inverse pass they cost a wave 2.6 - 5 k cycles (profiles/r06_cnv_stamps.txt)
            const bool xl_on = NR == 64 || (tc >> 3) < NR;
            const cplx* const xsrc = row_src(L + gridDim.x, xl_on ? (tc >> 3) : 0, tc & 7);
#define PZ_XL(J_)                                                                                              \
    if constexpr (EARLY) {                                                                                     \
        x[(J_) * (16 / BS)] = xl_on ? ld_stream(xsrc + 8 * ((J_) * (16 / BS))) : make_double2(0.0, 0.0);      \
        if constexpr (BS == 8) x[(J_) * 2 + 1] = xl_on ? ld_stream(xsrc + 8 * ((J_) * 2 + 1)) : make_double2(0.0, 0.0); \
    }
            if constexpr (SQ) {
                if (grp < 2) {
                    const cplx* A = grp == 0 ? A0 : A1;
#pragma unroll
                    for (int i = 0; i < AS; ++i) av[i] = A[i * RS];
                    PZ_XL(0)
#pragma unroll
                    for (int j = 1; j < AS; ++j) {
                        PZ_XL(j)
#pragma unroll
                        for (int i = 0; i < j; ++i)
                            if (i + j >= WLO) PZ_CMAC(acc[i + j], av[i], av[j])     // pairs i < j, doubled below
                    }
#pragma unroll
                    for (int k = WLO; k < NK; ++k) {
                        acc[k].x += acc[k].x; acc[k].y += acc[k].y;
                        if ((k & 1) == 0) PZ_CMAC(acc[k], av[k >> 1], av[k >> 1])
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < AS; ++i) av[i] = cadd(A0[i * RS], A1[i * RS]);
                    // hp (wave-uniform) selects the parity of i + j: two straight-line bodies
#define PZ_SQ_HALF(HP_)                                                                                       \
    {                                                                                                         \
        PZ_XL(0)                                                                                              \
        _Pragma("unroll") for (int j = 1; j < AS; ++j) {                                                      \
            PZ_XL(j)                                                                                          \
            _Pragma("unroll") for (int i = ((j & 1) ^ (HP_)); i < j; i += 2)                                  \
                if (i + j >= WLO) PZ_CMAC(acc[(i + j) >> 1], av[i], av[j])                                    \
        }                                                                                                     \
        _Pragma("unroll") for (int u = 0; u < NKH; ++u) {                                                     \
            if (2 * u + (HP_) >= WLO) {                                                                       \
                acc[u].x += acc[u].x; acc[u].y += acc[u].y;                                                   \
                if ((HP_) == 0 && u < AS) PZ_CMAC(acc[u], av[u], av[u])                                       \
            }                                                                                                 \
        }                                                                                                     \
    }
                    if (hp == 0) PZ_SQ_HALF(0) else PZ_SQ_HALF(1)
#undef PZ_SQ_HALF
                }
            } else if (grp < 2) {
                const cplx* A = grp == 0 ? A0 : A1;
                const cplx* B = grp == 0 ? B0 : B1;
#pragma unroll
                for (int i = 0; i < AS; ++i) av[i] = A[i * RS];
                // the operand of row j + 1 is read while row j's multiply-adds issue (read at the top of its own row, every row waited a full LDS
                // latency: `ds_read; s_waitcnt lgkmcnt(0)` sixteen times per thread in the ISA)
                constexpr int J0 = WLO > AS - 1 ? WLO - (AS - 1) : 0;   // first row that reaches the window
                cplx bn = B[J0 * RS];
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    PZ_XL(j)
                    if (j >= J0) {
                        const cplx bv = bn;
                        if (j + 1 < BS) bn = B[(j + 1) * RS];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < AS; ++i)
                            if (i + j >= WLO) PZ_CMAC(acc[i + j], av[i], bv)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < AS; ++i) av[i] = cadd(A0[i * RS], A1[i * RS]);
#define PZ_PW_HALF(HP_)                                                                                       \
    {                                                                                                         \
        constexpr int J0 = WLO > AS - 1 ? WLO - (AS - 1) : 0;                                                 \
        cplx b0n = B0[J0 * RS], b1n = B1[J0 * RS];   /* row j + 1 is read while row j's multiply-adds issue */ \
        _Pragma("unroll") for (int j = 0; j < BS; ++j) {                                                      \
            PZ_XL(j)                                                                                          \
            if (j >= J0) {                                                                                    \
                const cplx bv = cadd(b0n, b1n);                                                               \
                if (j + 1 < BS) { b0n = B0[(j + 1) * RS]; b1n = B1[(j + 1) * RS]; }                           \
                __builtin_amdgcn_sched_barrier(0);                                                            \
                _Pragma("unroll") for (int i = ((j & 1) ^ (HP_)); i < AS; i += 2)   /* i + j of parity HP_ */  \
                    if (i + j >= WLO) PZ_CMAC(acc[(i + j) >> 1], av[i], bv)                                   \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }                                                                                                 \
        }                                                                                                     \
    }
                if (hp == 0) PZ_PW_HALF(0) else PZ_PW_HALF(1)
#undef PZ_PW_HALF
            }
#undef PZ_XL
#if PZ_CNV_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            PZ_CSTAMP(3)   // limb convolution
            lds_barrier();   // every operand value has been read: the result rows take the place of the first operand rows
            if (grp < 2) {
#pragma unroll
                for (int k = WLO; k < NK; ++k) {
                    const int kk = k - g.offset;
                    if (kk >= 0 && kk < g.min_size) out[kk * RS] = acc[k];
                }
            } else {
#pragma unroll
                for (int u = 0; u < NKH; ++u) {
                    const int k = 2 * u + hp, kk = k - g.offset;
                    if (k >= WLO && k < NK && kk >= 0 && kk < g.min_size) out[kk * RS] = acc[u];
                }
            }
            // product limbs beyond a_size + b_size - 2 are zero (reim4/arithmetic_ref.rs:235-247); groups 0, 1, 2 fill them for their term
            if (grp < 3)
                for (int kk = max(NK - g.offset, 0); kk < g.min_size; ++kk) out[kk * RS] = make_double2(0.0, 0.0);
        }
        lds_barrier();
        PZ_CSTAMP(4)   // barrier + write-back + barrier
        // ---- inverse row DFT of the 3 x min_size result rows x conj tw12 -> T2'; the next tile's operand rows and twiddle row start travelling
        //      in its gaps (four groups of four loads: a vector-memory instruction blocks its wave until the path accepts it) ----
        cplx twn = make_double2(0.0, 0.0);
        __builtin_amdgcn_sched_barrier(0);   // (the loads below must not be hoisted above the convolution: 64 more live registers there spill 300 bytes)
        const int ti = pz_opaque(tid);
        const int row = ti >> 3, o = ti & 7;
        const long long Ln = L + gridDim.x;
        const cplx* nsrc = row_src(Ln, (NR == 64 || row < NR) ? row : 0, o);
#define PZ_XGROUP(G4)                                                                                          \
    {                                                                                                          \
        if (NR == 64 || row < NR) {                                                                            \
            _Pragma("unroll") for (int n1 = 4 * (G4); n1 < 4 * (G4) + 4; ++n1) x[n1] = ld_stream(nsrc + 8 * n1); \
        } else {                                                                                               \
            _Pragma("unroll") for (int n1 = 4 * (G4); n1 < 4 * (G4) + 4; ++n1) x[n1] = make_double2(0.0, 0.0); \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
        if (tid < M2) twn = g.tw12t[(long long)((Ln < ntiles ? Ln : ntiles - 1) % g.m1) * M2 + tid];
        if (!EARLY) PZ_XGROUP(0)
        if (!EARLY && !PZ_CNV_SPREAD) { PZ_XGROUP(1) PZ_XGROUP(2) PZ_XGROUP(3) }
        PZ_CSTAMP(5)   // issue of the next tile's first loads
        if (row < 3 * g.min_size) {   // (one block: with the four steps as four conditionals the working set is live across their joins - 256 registers + 64 B)
            const int term = row / g.min_size, kk = row % g.min_size;
            const cplx* twrow = twrow2 + par * M2;
            cplx* rowbuf = lds + row * RS;
            cplx u[16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) u[8 * h + k2] = rowbuf[o + 8 * h + 16 * k2];
            Bfly<8, true>::run(u);
            __builtin_amdgcn_sched_barrier(0);
            if (!EARLY && PZ_CNV_SPREAD) PZ_XGROUP(1)
            Bfly<8, true>::run(u + 8);
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) {
                    cplx v = u[8 * h + oo];
                    const int k1 = o + 8 * h;
                    if (k1 > 0 && oo > 0) v = cmulc(v, wl[oo * k1]);
                    rowbuf[k1 * 9 + oo] = v;
                }
            __builtin_amdgcn_sched_barrier(0);
            if (!EARLY && PZ_CNV_SPREAD) PZ_XGROUP(2)
            row_sync();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 9 + o];
            Bfly<16, true>::run(u);
            __builtin_amdgcn_sched_barrier(0);
            if (!EARLY && PZ_CNV_SPREAD) PZ_XGROUP(3)
            cplx* dst = g.T2 + (((long long)term * g.batch + bt) * g.min_size + kk) * m + (long long)q1 * M2 + o;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) st_stream(dst + 8 * n1, cmulc(u[n1], twrow[o + 8 * n1]));
        } else if (!EARLY && PZ_CNV_SPREAD) {
            PZ_XGROUP(1) PZ_XGROUP(2) PZ_XGROUP(3)
        }
#undef PZ_XGROUP
        if (tid < M2) twrow2[(par ^ 1) * M2 + tid] = twn;   // the other twiddle row: nobody reads it before the barrier below
        PZ_CSTAMP(6)   // inverse row transforms + stores (issue)
        lds_barrier();   // the result rows have been consumed: the forward pass of the next tile may overwrite the tile (LDS ordering only -
                         // the stores above drain under the next tile)
        PZ_CSTAMP(7)   // barrier
#if PZ_CNV_STAMP
        ++st_tiles;
#endif
    }
#if PZ_CNV_STAMP
    if ((tid & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 9 || blockIdx.x == 130 || blockIdx.x == 255))
        printf("CSTAMP wg %d wave %d tiles %d total %llu | xwait %llu fwd %llu bar1 %llu conv %llu wb %llu issue %llu inv %llu bar2 %llu\n",
               (int)blockIdx.x, tid >> 6, st_tiles, (unsigned long long)(st_t - st_t0), st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4],
               st_acc[5], st_acc[6], st_acc[7]);
#endif
#undef PZ_CSTAMP
}
#undef PZ_CMAC

// convolution.rs:147-203 + :395-421: res limb kk = sum_j a[kk + offset - j] * b[j], wrapping i64, coefficient-wise
struct CnvConstArgs {
    long long* res;
    const long long* a;
    const long long* b;       // b_size constants (device)
    int res_cols, res_col, a_cols, a_col, a_size, b_size, min_size, offset, n;
};
__global__ void __launch_bounds__(256) k_cnv_by_const(CnvConstArgs g) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.n) return;
    const int kk = blockIdx.y;
    const int k = kk + g.offset;
    unsigned long long acc = 0;
    if (k < g.a_size + g.b_size) {
        const int j_min = k >= g.a_size - 1 ? k - (g.a_size - 1) : 0;
        const int j_max = min(k + 1, g.b_size);
        for (int j = j_min; j < j_max; ++j)
            acc += (unsigned long long)g.a[(long long)g.n * ((long long)(k - j) * g.a_cols + g.a_col) + x] * (unsigned long long)g.b[j];
    }
    g.res[(long long)g.n * ((long long)kk * g.res_cols + g.res_col) + x] = (long long)acc;
}

}  // namespace pz
