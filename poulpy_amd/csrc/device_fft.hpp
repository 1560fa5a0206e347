// device_fft.hpp — negacyclic f64 FFT over Z[X]/(X^N+1) for gfx950, two-pass.
//
// Replaces (does not translate) reim/fft_ref.rs + reim/ifft_ref.rs of the
// reference: the reference runs an in-place DIF with bit-reversed output; here
// the m = N/2 point twisted DFT
//     D[q] = sum_j z_j * exp(2*pi*i * j*(4q+1) / (4m)),  z_j = a_j + i*a_{j+m}
// is computed in NATURAL frequency order ("device order") by a four-step split
// m = m1*m2 (j = j1*m2 + j2, q = q1 + m1*q2):
//   pass 1: for every column j2, a length-m1 twisted DFT over j1, times
//           tw12[j2][q1] = exp(2*pi*i * j2*(4*q1+1)/(4m)); written transposed
//           as T[j2][q1];
//   pass 2: for every q1, a plain length-m2 DFT over j2, written as D[q2][q1].
// Each pass is two in-register radix-R butterflies (R <= 16) with one LDS
// exchange.  Lanes always run along the contiguous axis of the global arrays
// (j2 or q1), so every global access is a run of >= 128 B.
// DFT-domain values are opaque to poulpy-hal callers, so natural
// order is legal; parity is on the normalized i64 outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pz {

typedef double2 cplx;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// a*b
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
// a*conj(b)
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) {
    return make_double2(__builtin_fma(a.x, b.x, a.y * b.y), __builtin_fma(a.y, b.x, -(a.x * b.y)));
}
template <bool CONJ>
__device__ __forceinline__ cplx cmul_t(cplx a, cplx b) { return CONJ ? cmulc(a, b) : cmul(a, b); }

// ---- in-register radix-R DFT, R in {1,2,4,8,16}; natural-order output --------
// X[k] = sum_n v[n] * w^(+nk) (INV=false) or w^(-nk) (INV=true), w = exp(2*pi*i/R).
#define PZ_C8 0.70710678118654752440084436210485   /* cos(pi/4) */
#define PZ_C16 0.92387953251128675612818318939679  /* cos(pi/8) */
#define PZ_S16 0.38268343236508977172845998403040  /* sin(pi/8) */

// d * w_R^(+-n), n < R/2; e = n*16/R is a compile-time constant after unrolling
template <int R, bool INV>
__device__ __forceinline__ cplx tw_small(cplx d, int n) {
    const int e = n * (16 / R);
    switch (e) {
        case 0: return d;
        case 4: return INV ? make_double2(d.y, -d.x) : make_double2(-d.y, d.x);
        case 2: return INV ? make_double2(PZ_C8 * (d.x + d.y), PZ_C8 * (d.y - d.x))
                           : make_double2(PZ_C8 * (d.x - d.y), PZ_C8 * (d.x + d.y));
        case 6: return INV ? make_double2(PZ_C8 * (d.y - d.x), -PZ_C8 * (d.x + d.y))
                           : make_double2(-PZ_C8 * (d.x + d.y), PZ_C8 * (d.x - d.y));
        case 1: return cmul_t<INV>(d, make_double2(PZ_C16, PZ_S16));
        case 3: return cmul_t<INV>(d, make_double2(PZ_S16, PZ_C16));
        case 5: return cmul_t<INV>(d, make_double2(-PZ_S16, PZ_C16));
        default: return cmul_t<INV>(d, make_double2(-PZ_C16, PZ_S16));
    }
}

template <int R, bool INV>
struct Bfly {
    static __device__ __forceinline__ void run(cplx* v) {
        cplx a[R / 2], b[R / 2];
#pragma unroll
        for (int n = 0; n < R / 2; ++n) {
            a[n] = cadd(v[n], v[n + R / 2]);
            b[n] = tw_small<R, INV>(csub(v[n], v[n + R / 2]), n);
        }
        Bfly<R / 2, INV>::run(a);
        Bfly<R / 2, INV>::run(b);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            v[2 * k] = a[k];
            v[2 * k + 1] = b[k];
        }
    }
};
template <bool INV>
struct Bfly<1, INV> {
    static __device__ __forceinline__ void run(cplx*) {}
};

// ---- addressing of the polynomials a launch works on --------------------------
// polynomial p = (b*nj + j)*ni + i lives at element offset b*sb + j*sj + i*si + s0
struct PolyMap {
    int nj, ni;
    long long sb, sj, si, s0;
};
__device__ __forceinline__ long long map_off(const PolyMap& mp, int p) {
    int i = p % mp.ni;
    int t = p / mp.ni;
    int j = t % mp.nj;
    int b = t / mp.nj;
    return (long long)b * mp.sb + (long long)j * mp.sj + (long long)i * mp.si + mp.s0;
}

// Rust `(x).round() as i64`: half away from zero, saturating, NaN -> 0
// (reim/conversion.rs:43-60)
__device__ __forceinline__ long long round_to_i64(double x) {
    double r = round(x);
    if (!(r == r)) return 0;
    if (r >= 9223372036854775808.0) return 0x7fffffffffffffffLL;
    if (r <= -9223372036854775808.0) return (long long)0x8000000000000000ULL;
    return (long long)r;
}

template <int A, int B>
struct MaxOf {
    static constexpr int v = A > B ? A : B;
};

// =================================================================================
// forward pass 1: i64 coefficients -> T[j2][q1]
//   grid.x = npolys * (m2/CB); block = max(R1,R2)*CB threads; LDS (R1+1)*CB*R2 cplx
// =================================================================================
template <int R1, int R2, int CB>
__global__ void __launch_bounds__((R1 > R2 ? R1 : R2) * CB)
k_fwd_pass1(const long long* __restrict__ src, PolyMap smap, cplx* __restrict__ T, int m2,
            const cplx* __restrict__ tw1, const cplx* __restrict__ wL1, const cplx* __restrict__ tw12) {
    constexpr int M1 = R1 * R2;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x;
    const int ncb = m2 / CB;
    const int p = blockIdx.x / ncb;
    const int c0 = (blockIdx.x % ncb) * CB;
    const long long m = (long long)M1 * m2;
    const long long* a = src + map_off(smap, p);
    cplx* Tp = T + (long long)p * m;

    if (tid < R2 * CB) {
        const int o = tid / CB, c = tid % CB;
        cplx v[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int j1 = o + R2 * n1;
            const long long idx = (long long)j1 * m2 + c0 + c;
            const double re = (double)a[idx];
            const double im = (double)a[idx + m];
            v[n1] = cmul(make_double2(re, im), tw1[j1]);
        }
        Bfly<R1, false>::run(v);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            cplx x = v[k1];
            if (R2 > 1 && k1 > 0) x = cmul(x, wL1[o * k1]);
            lds[(o * CB + c) * (R1 + 1) + k1] = x;
        }
    }
    __syncthreads();
    if (tid < R1 * CB) {
        const int k1 = tid % R1, c = tid / R1;
        cplx u[R2];
#pragma unroll
        for (int o = 0; o < R2; ++o) u[o] = lds[(o * CB + c) * (R1 + 1) + k1];
        Bfly<R2, false>::run(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const long long tix = (long long)(c0 + c) * M1 + k1 + R1 * k2;
            Tp[tix] = cmul(u[k2], tw12[tix]);
        }
    }
}

// =================================================================================
// forward pass 2: T[j2][q1] -> D[q2][q1] (natural order q = q1 + m1*q2), optional
// pointwise multiply by a prepared polynomial (svp_apply_dft fusion).
//   grid.x = npolys * (m1/QB); block = max(R1,R2)*QB; LDS R1*R2*QB cplx
// =================================================================================
template <int R1, int R2, int QB>
__global__ void __launch_bounds__((R1 > R2 ? R1 : R2) * QB)
k_fwd_pass2(const cplx* __restrict__ T, double* __restrict__ dst, PolyMap dmap, int m1,
            const cplx* __restrict__ wL2, const cplx* __restrict__ mul) {
    constexpr int M2 = R1 * R2;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x;
    const int nqb = m1 / QB;
    const int p = blockIdx.x / nqb;
    const int q0 = (blockIdx.x % nqb) * QB;
    const long long m = (long long)M2 * m1;
    const cplx* Tp = T + (long long)p * m;
    cplx* out = reinterpret_cast<cplx*>(dst + map_off(dmap, p));

    if (tid < R2 * QB) {
        const int o = tid / QB, qb = tid % QB;
        cplx v[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) v[n1] = Tp[(long long)(o + R2 * n1) * m1 + q0 + qb];
        Bfly<R1, false>::run(v);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            cplx x = v[k1];
            if (R2 > 1 && k1 > 0) x = cmul(x, wL2[o * k1]);
            lds[(k1 * R2 + o) * QB + qb] = x;
        }
    }
    __syncthreads();
    if (tid < R1 * QB) {
        const int k1 = tid / QB, qb = tid % QB;
        cplx u[R2];
#pragma unroll
        for (int o = 0; o < R2; ++o) u[o] = lds[(k1 * R2 + o) * QB + qb];
        Bfly<R2, false>::run(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const long long oix = (long long)(k1 + R1 * k2) * m1 + q0 + qb;
            cplx x = u[k2];
            if (mul) x = cmul(mul[oix], x);
            out[oix] = x;
        }
    }
}

// =================================================================================
// inverse pass 2: D[q2][q1] -> T[j2][q1] (times conj(tw12)); value scaled by m2
// =================================================================================
template <int R1, int R2, int QB>
__global__ void __launch_bounds__((R1 > R2 ? R1 : R2) * QB)
k_inv_pass2(const double* __restrict__ src, PolyMap smap, cplx* __restrict__ T, int m1,
            const cplx* __restrict__ wL2, const cplx* __restrict__ tw12) {
    constexpr int M2 = R1 * R2;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x;
    const int nqb = m1 / QB;
    const int p = blockIdx.x / nqb;
    const int q0 = (blockIdx.x % nqb) * QB;
    const long long m = (long long)M2 * m1;
    const cplx* in = reinterpret_cast<const cplx*>(src + map_off(smap, p));
    cplx* Tp = T + (long long)p * m;

    if (tid < R1 * QB) {
        const int k1 = tid / QB, qb = tid % QB;
        cplx u[R2];
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) u[k2] = in[(long long)(k1 + R1 * k2) * m1 + q0 + qb];
        Bfly<R2, true>::run(u);
#pragma unroll
        for (int o = 0; o < R2; ++o) {
            cplx x = u[o];
            if (R2 > 1 && k1 > 0 && o > 0) x = cmulc(x, wL2[o * k1]);
            lds[(k1 * R2 + o) * QB + qb] = x;
        }
    }
    __syncthreads();
    if (tid < R2 * QB) {
        const int o = tid / QB, qb = tid % QB;
        cplx v[R1];
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) v[k1] = lds[(k1 * R2 + o) * QB + qb];
        Bfly<R1, true>::run(v);
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const long long tix = (long long)(o + R2 * n1) * m1 + q0 + qb;
            Tp[tix] = cmulc(v[n1], tw12[tix]);
        }
    }
}

// =================================================================================
// inverse pass 1: T[j2][q1] -> i64 coefficients, round(x/m) half-away, saturating
// (tw1inv[j1] = conj(psi1^j1)/m carries the exact power-of-two scale).
// PROBE: record max |x - round(x)| (exactness margin) through atomicMax on bits.
// =================================================================================
template <int R1, int R2, int CB, bool PROBE>
__global__ void __launch_bounds__((R1 > R2 ? R1 : R2) * CB)
k_inv_pass1(const cplx* __restrict__ T, long long* __restrict__ dst, PolyMap dmap, int m2,
            const cplx* __restrict__ tw1inv, const cplx* __restrict__ wL1, unsigned long long* __restrict__ margin) {
    constexpr int M1 = R1 * R2;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x;
    const int ncb = m2 / CB;
    const int p = blockIdx.x / ncb;
    const int c0 = (blockIdx.x % ncb) * CB;
    const long long m = (long long)M1 * m2;
    const cplx* Tp = T + (long long)p * m;
    long long* out = dst + map_off(dmap, p);

    if (tid < R1 * CB) {
        const int k1 = tid % R1, c = tid / R1;
        cplx u[R2];
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) u[k2] = Tp[(long long)(c0 + c) * M1 + k1 + R1 * k2];
        Bfly<R2, true>::run(u);
#pragma unroll
        for (int o = 0; o < R2; ++o) {
            cplx x = u[o];
            if (R2 > 1 && k1 > 0 && o > 0) x = cmulc(x, wL1[o * k1]);
            lds[(o * CB + c) * (R1 + 1) + k1] = x;
        }
    }
    __syncthreads();
    if (tid < R2 * CB) {
        const int o = tid / CB, c = tid % CB;
        cplx v[R1];
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) v[k1] = lds[(o * CB + c) * (R1 + 1) + k1];
        Bfly<R1, true>::run(v);
        double worst = 0.0;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int j1 = o + R2 * n1;
            const cplx w = cmul(v[n1], tw1inv[j1]);
            const long long idx = (long long)j1 * m2 + c0 + c;
            out[idx] = round_to_i64(w.x);
            out[idx + m] = round_to_i64(w.y);
            if (PROBE) {
                worst = fmax(worst, fabs(w.x - round(w.x)));
                worst = fmax(worst, fabs(w.y - round(w.y)));
            }
        }
        if (PROBE) atomicMax(margin, (unsigned long long)__double_as_longlong(worst));
    }
}

}  // namespace pz
