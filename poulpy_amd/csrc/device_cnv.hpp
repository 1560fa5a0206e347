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
