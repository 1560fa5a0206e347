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
