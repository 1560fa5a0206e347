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

// Timing-ablation knobs (POULPY_DBG_MID_SKIP, POULPY_DBG_SMALL_SKIP, POULPY_DBG_BR_SKIP, POULPY_DBG_BRL) are compiled in only with
// -DPZ_ABLATE=1 (POULPY_BUILD_DEFS=-DPZ_ABLATE=1 POULPY_BUILD_TAG=ablate): a run-time test around a global load inside a software-pipelined
// loop makes the compiler's s_waitcnt insertion assume the worst case at every join (round 3: the product loop of k_mid128 waited with
// vmcnt(0) for the key row it had just requested instead of vmcnt(4) for the one requested a row earlier).
#ifndef PZ_ABLATE
#define PZ_ABLATE 0
#endif
#define PZ_DBG(x) (PZ_ABLATE ? (x) : 0)
// Streaming hints for data that is touched once per kernel (the i64 limbs, T', T2'): non-temporal loads / stores.  Measured on
// MI355X (profiles/r02_hbm_copy_tuned.txt): a 16 B-per-lane copy runs at 6.25 TB/s plain and 6.5-6.6 TB/s with both hints.
// PZ_STREAM_HINTS: bit 0 loads, bit 1 stores (build-time, for A/B runs; default both).
#ifndef PZ_STREAM_HINTS
#define PZ_STREAM_HINTS 3
#endif
typedef double pz_dbl2 __attribute__((ext_vector_type(2)));
typedef short pz_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cplx ld_stream(const cplx* p) {
    if (PZ_STREAM_HINTS & 1) { const pz_dbl2 v = __builtin_nontemporal_load(reinterpret_cast<const pz_dbl2*>(p)); return make_double2(v.x, v.y); }
    return *p;
}
__device__ __forceinline__ long long ld_stream(const long long* p) { return (PZ_STREAM_HINTS & 1) ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st_stream(cplx* p, cplx v) {
    if (PZ_STREAM_HINTS & 2) { const pz_dbl2 t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<pz_dbl2*>(p)); }
    else *p = v;
}
#ifndef PZ_STREAM_I64_NT
#define PZ_STREAM_I64_NT 1   // the tail's digit stores (A/B knob)
#endif
#ifndef PZ_TAIL_D16_ADDR
#define PZ_TAIL_D16_ADDR 1   // 16-bit digit stores: 1 one base per thread + constant offsets, 0 the element from the coefficient's row / column (A/B knob)
#endif
#ifndef PZ_TAIL_NZD
#define PZ_TAIL_NZD 1   // side-copy tensoring tails: "normalizing, not raw" known at compile time (0: tested at run time as in the other forms, A/B knob)
#endif
#ifndef PZ_TAIL_D16R_F64
#define PZ_TAIL_D16R_F64 0   // the pairwise tensoring tails that read 16-bit side copies: f64 normalization steps (0: the integer steps, A/B knob)
#endif
__device__ __forceinline__ void st_stream(long long* p, long long v) {
    if ((PZ_STREAM_HINTS & 2) && PZ_STREAM_I64_NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// A copy of v the optimizer cannot see through: everything derived from it is computed where it is used instead of being hoisted out
// of the enclosing loop and kept live across it.
__device__ __forceinline__ int pz_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// LDS traffic between the 16 lanes that own one row needs no workgroup barrier: the lanes are in one wave,
// whose LDS instructions execute in order; this only stops the compiler from moving them across.
__device__ __forceinline__ void row_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that only orders LDS traffic (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() also
// drains vmcnt, which would make every barrier wait for the tile's global stores and for the prefetched
// loads; threads of this kernel never exchange data through global memory, so LDS ordering is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// limb-wise elementwise ops of k_ew (device_ops.hpp); shared with the host-side composition code
enum EwOp : int {
    EW_ZERO = 0,
    EW_COPY = 1,     // res = a
    EW_NEG = 2,      // res = -a
    EW_ADD = 3,      // res = a + b
    EW_SUB = 4,      // res = a - b
    EW_CMUL = 5,     // res = a * b  (complex pointwise, interleaved; `a` is the prepared poly)
    EW_ADD_I64 = 6,  // res = a + b  (wrapping i64)
    EW_SUB_I64 = 7,  // res = a - b  (wrapping i64)
    EW_NEG_I64 = 8,  // res = -a
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

// Rust `(x).round() as i64`: half away from zero, saturating, NaN -> 0 (reim/conversion.rs:43-60).
// Branch-free: round = trunc(x) +- 1 when the (exactly computed) fraction reaches one half.
__device__ __forceinline__ double round_half_away(double x) {
    const double t = trunc(x);
    const double f = x - t;  // exact
    return t + ((fabs(f) >= 0.5) ? copysign(1.0, x) : 0.0);
}
__device__ __forceinline__ long long sat_i64_from_integral(double r) {
    // r is integral (or NaN/inf).  In-range conversion after clamping, then the saturated cases by select.
    const double rc = fmin(fmax(r, -9223372036854775808.0), 9223372036854774784.0);
    long long v = (long long)rc;
    v = (r >= 9223372036854775808.0) ? 0x7fffffffffffffffLL : v;
    v = (r != r) ? 0 : v;
    return v;
}
__device__ __forceinline__ long long round_to_i64(double x) { return sat_i64_from_integral(round_half_away(x)); }
// Rounding-margin probe (pz_module_set_margin_probe / pz_module_get_margin): the largest |x - round(x)| over every value an inverse
// transform rounds, on the kernels the product path dispatches (a run-time, wave-uniform `margin != nullptr`; no instantiation of its own
// and nothing live in registers when it is off: the distance is taken from the values still in registers, in a block of its own in front of
// the rounding loop).  Bits of a non-negative double order like the double; the plain read keeps all but the first few lanes off the atomic.
__device__ __forceinline__ double margin_dist(double x) { return fabs(x - round_half_away(x)); }
__device__ __forceinline__ void margin_note(unsigned long long* margin, double worst) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(worst);   // NaN / inf inputs read as a huge margin, on purpose
    if (bits > __atomic_load_n(margin, __ATOMIC_RELAXED)) atomicMax(margin, bits);
}
// exact for |r| < 2^51, r integral: the integer sits in the mantissa of r + 1.5*2^52
__device__ __forceinline__ long long fast_i64_from_integral(double r) {
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    return __double_as_longlong(r + magic) - __double_as_longlong(magic);
}

// =================================================================================
// forward pass 1: i64 coefficients -> T[j2][q1]
//   grid.x = npolys * (m2/CB); block = max(R1,R2)*CB threads; LDS (R1+1)*CB*R2 cplx
// =================================================================================
// ROWMAJOR = true writes T'[q1][j2] (rows of m2 contiguous points, what the fused middle kernel
// consumes; tw12 must then be the [q1][j2] copy of the table) instead of the transposed T[j2][q1].
// SRC32: the coefficients are 32-bit digits at the same element offsets (the blind rotation's accumulator between two blocks, api_br.hip)
// SRC = 0 i64 coefficients | 1 = SRC32 | 2: 16-bit digits in the fused tail's tile order (TailArgs::d16*: smap addresses a limb of n int16, the
// workgroup's tile [half h][row j1][CB columns] starts at (c0 / CB) 2 M1 CB) - the fused multiply + relinearize reads the pair column of a
// GLWETensor that only ever existed as those copies (api_cnv.hip)
// W16 (with SRC = 0): the coefficients just read also leave as 16-bit values in the fused tail's tile order, polynomial p at w16 + p n (the mask column of
// the add / sub automorphism forms, whose tail then takes it as a 16-bit operand); a value beyond 16 bits raises *wide (launch_fwd_pass1_w16)
template <int R1, int R2, int CB, bool ROWMAJOR, int SRC, bool W16 = false>
__device__ __forceinline__ void fwd_pass1_body(const long long* __restrict__ src, PolyMap smap, cplx* __restrict__ T, int m2,
                                               const cplx* __restrict__ tw1, const cplx* __restrict__ wL1, const cplx* __restrict__ tw12, long long mask, int npolys_xcd,
                                               short* __restrict__ w16 = nullptr, unsigned* __restrict__ wide = nullptr) {
    constexpr bool SRC32 = SRC == 1;
    constexpr int M1 = R1 * R2;
    constexpr int NT = (R1 > R2 ? R1 : R2) * CB;
    extern __shared__ cplx lds[];  // (R1+1)*CB*R2 exchange | tw1[M1] | wL1[M1]
    const int tid = threadIdx.x;
    const int ncb = m2 / CB;
    // npolys_xcd > 0: XCD-aware block order — workgroup ids go round-robin over the 8 XCDs; the ncb column blocks of one polynomial
    // are given to ONE XCD back to back, so that the 128-byte pieces of every 1 KiB coefficient row are requested through one L2
    // close in time (a copy with this access shape: 4.9 -> 5.7 TB/s, profiles/r02_hbm_pass_pattern.txt); grid = ceil(npolys / 8) * 8 * ncb
    int bid = blockIdx.x;
    if (npolys_xcd > 0) {
        const int xcd = bid & 7, slot = bid >> 3;
        const int px = (slot / ncb) * 8 + xcd;
        if (px >= npolys_xcd) return;
        bid = px * ncb + slot % ncb;
    }
    const int p = bid / ncb;
    const int c0 = (bid % ncb) * CB;
    const long long m = (long long)M1 * m2;
    const long long* a = src + map_off(smap, p);
    cplx* Tp = T + (long long)p * m;
    // twist and stage roots from LDS (a global gather in front of dependent arithmetic costs a memory latency)
    cplx* tws = lds + (R1 + 1) * CB * R2;
    cplx* wls = tws + M1;

    const bool is_a = tid < R2 * CB;
    const int o = tid / CB, c = tid % CB;
    // issue the coefficient loads first, stage the tables while they travel
    long long raw_re[R1], raw_im[R1];
    if (is_a) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const long long idx = (long long)(o + R2 * n1) * m2 + c0 + c;
            if constexpr (SRC == 2) {
                const short* a16 = reinterpret_cast<const short*>(src) + map_off(smap, p) + (long long)(c0 / CB) * (2 * M1 * CB) + (o + R2 * n1) * CB + c;
                raw_re[n1] = (long long)a16[0];
                raw_im[n1] = (long long)a16[M1 * CB];
            } else if constexpr (SRC32) {
                const int* a32 = reinterpret_cast<const int*>(src) + map_off(smap, p);
                raw_re[n1] = (long long)a32[idx];
                raw_im[n1] = (long long)a32[idx + m];
            } else {
                raw_re[n1] = (ROWMAJOR ? ld_stream(a + idx) : a[idx]) & mask;   // mask = -1 except for cnv_prepare's last active limb (reim/conversion.rs:31-40)
                raw_im[n1] = (ROWMAJOR ? ld_stream(a + idx + m) : a[idx + m]) & mask;
            }
        }
    }
    for (int t = tid; t < M1; t += NT) {
        tws[t] = tw1[t];
        wls[t] = wL1[t];
    }
    __syncthreads();
    if constexpr (W16) {
        if (is_a) {
            short* d = w16 + (long long)p * (2 * m) + (long long)(c0 / CB) * (2 * M1 * CB) + o * CB + c;
            bool wd = false;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                wd = wd || ((unsigned long long)raw_re[n1] + 32767ull) >= 65535ull || ((unsigned long long)raw_im[n1] + 32767ull) >= 65535ull;
                d[(R2 * n1) * CB] = (short)raw_re[n1];
                d[(M1 + R2 * n1) * CB] = (short)raw_im[n1];
            }
            if (wd) atomicOr(wide, 1u);
        }
    }
    if (is_a) {
        cplx v[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1)
            v[n1] = cmul(make_double2((double)raw_re[n1], (double)raw_im[n1]), tws[o + R2 * n1]);
        Bfly<R1, false>::run(v);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            cplx x = v[k1];
            if (R2 > 1 && k1 > 0) x = cmul(x, wls[o * k1]);
            lds[(o * CB + c) * (R1 + 1) + k1] = x;
        }
    }
    __syncthreads();
    if (tid < R1 * CB) {
        const int k1 = ROWMAJOR ? tid / CB : tid % R1;
        const int c = ROWMAJOR ? tid % CB : tid / R1;
        cplx u[R2];
#pragma unroll
        for (int o = 0; o < R2; ++o) u[o] = lds[(o * CB + c) * (R1 + 1) + k1];
        Bfly<R2, false>::run(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const long long tix = ROWMAJOR ? (long long)(k1 + R1 * k2) * m2 + c0 + c : (long long)(c0 + c) * M1 + k1 + R1 * k2;
            if (ROWMAJOR) st_stream(Tp + tix, cmul(u[k2], tw12[tix]));
            else Tp[tix] = cmul(u[k2], tw12[tix]);
        }
    }
}
template <int R1, int R2, int CB, bool ROWMAJOR = false, bool SRC32 = false>
__global__ void __launch_bounds__((R1 > R2 ? R1 : R2) * CB)
k_fwd_pass1(const long long* __restrict__ src, PolyMap smap, cplx* __restrict__ T, int m2,
            const cplx* __restrict__ tw1, const cplx* __restrict__ wL1, const cplx* __restrict__ tw12, long long mask, int npolys_xcd) {
    fwd_pass1_body<R1, R2, CB, ROWMAJOR, SRC32 ? 1 : 0>(src, smap, T, m2, tw1, wL1, tw12, mask, npolys_xcd);
}
template <int R1, int R2, int CB>
__global__ void __launch_bounds__((R1 > R2 ? R1 : R2) * CB)
k_fwd_pass1_w16(const long long* __restrict__ src, PolyMap smap, cplx* __restrict__ T, int m2,
                const cplx* __restrict__ tw1, const cplx* __restrict__ wL1, const cplx* __restrict__ tw12, int npolys_xcd, short* __restrict__ w16, unsigned* __restrict__ wide) {
    fwd_pass1_body<R1, R2, CB, true, 0, true>(src, smap, T, m2, tw1, wL1, tw12, -1ll, npolys_xcd, w16, wide);
}
template <int R1, int R2, int CB>
__global__ void __launch_bounds__((R1 > R2 ? R1 : R2) * CB)
k_fwd_pass1_t16(const long long* __restrict__ src, PolyMap smap, cplx* __restrict__ T, int m2,
                const cplx* __restrict__ tw1, const cplx* __restrict__ wL1, const cplx* __restrict__ tw12, int npolys_xcd) {
    fwd_pass1_body<R1, R2, CB, true, 2>(src, smap, T, m2, tw1, wL1, tw12, -1ll, npolys_xcd);
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
// PROBE: record max |x - round(x)| (exactness margin, margin_note above) - compile-time: as a run-time test in front of the store loop it cost
// this kernel 3.12 -> 4.57 ms per 16 Ki polynomials at N = 2^16 (round 5, profiles/r05_bench_lines_hal.txt vs r04).
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
                worst = fmax(worst, fmax(margin_dist(w.x), margin_dist(w.y)));
            }
        }
        if (PROBE) margin_note(margin, worst);
    }
}

// =================================================================================
// fused tail: inverse pass 1 + (optional) vec_znx_big_add_small_assign + same-base
// vec_znx_big_normalize (res_offset = 0), for one output column of one ciphertext.
//
// One workgroup owns a block of CB columns j2 of one (ciphertext, column) and walks its
// limbs from the least significant one to limb 0, exactly like the reference's carry chain
// (reference/vec_znx/normalize.rs:50-144 with lsh = 0): per limb it runs the inverse
// length-m1 transform of that limb's T block, rounds to i64 (the VecZnxBig value, which is
// never written to HBM), stages the 2*m1*CB coefficients in LDS and lets every thread
// normalize 2*min(R1,R2) of them with the carries held in registers, storing the balanced
// digits as full 128-byte runs.  Saves the 16 B/coefficient round trip of VecZnxBig.
//   T      : [poly p][j2][q1], p = (b*nlimbs + limb)*ncols + col  (output of inverse pass 2)
//   res    : VecZnx (res_cols, res_size), batch stride res_bs; column res_col0 + col
//   small  : optional VecZnx added to column 0 before normalizing (key-switch body, glwe.rs:237)
// grid.x = batch*col_count*(m2/CB)
// =================================================================================
struct TailArgs {
    const cplx* T;
    long long* res;
    const long long* small;      // may be null
    long long res_bs, small_bs;  // batch strides (scalars)
    int nlimbs, ncols;           // limbs / columns of the VecZnxBig being consumed
    int res_cols, res_size;      // output container
    int small_cols, small_size;
    int base2k, m2;
    const cplx* tw1inv;
    const cplx* wL1;
    unsigned long long* margin;
    int acc32;   // k_inv_tail<.., ACC32>: bit 0 `small` holds 32-bit digits, bit 1 `res` takes 32-bit digits (same element strides); bits 3 / 4: `small` holds /
                 // `res` takes 16-bit digits in the tile order below, every polynomial at its i64 element offset (the blind rotation's accumulator); bit 2: the operand is
                 // d16a[column][ciphertext][limb][n] int16 in the tile order below (body_bs = elements between columns): a GLWETensor kept as 16-bit digits
    // automorphism family (poulpy-core automorphism/glwe_ct.rs:96-275): the value that enters the carry chain is
    // s(n) * (big[n] + small[n]) with s(n) = -1 iff (n * auto_mul) mod 2N >= N (auto_neg flips every sign);
    // small_all: `small` has an operand for every column, not only the body column
    int small_all;
    unsigned auto_mul;  // 0: no sign
    int auto_neg;
    int col_base, col_count;  // this launch covers columns [col_base, col_base + col_count) of the ncols
    int body_col;             // without small_all: the column that receives `small` column 0 (0 for a key switch, `col` for ggsw_expand_row)
    // gather_mul != 0 (with small_all): the operand added before the carry chain is  -+phi^-1(small)[n] (+ small[col 0][n], the
    // key-switch body, for column 0), gathered here from the natural-order `small` instead of being prepared by a separate pass:
    //   phi^-1(a)[n] = +-a[(n * gather_mul) mod 2N]  (negated when that index is >= N); gather_neg: operand = -phi^-1(a) (sub modes).
    // xcd_map: decode blockIdx so that all column blocks of one (ciphertext, column) run on one XCD (its L2 then serves the 8-byte
    // gathers: every line of the 512 KiB limb is fetched from HBM once)
    unsigned gather_mul;
    int gather_neg, xcd_map;
    // pre_body (with small_all): the big value arrives already permuted (the middle kernel moved the spectrum, so the inverse
    // transform is phi(big)); the operand is one stream per column at the natural index: small[col][n], or on the body column
    // body_src[n] (phi(body) +- a0, prepared by k_automorphism in the workspace); small_neg negates it (sub forms)
    int pre_body, small_neg;
    int body_add;   // body column: the operand is body_src[n] + small[body column][n] (the pre-pass then only permutes, no second operand)
    const long long* body_src;
    long long body_bs, body_ls;
    // plain glwe_automorphism in the spectral form (res = phi(normalize(big)), glwe_ct.rs:65-71): the inverse transform is phi(big) with
    // phi's signs; auto_mul undoes them in front of the carry chain (so that it runs on the values the reference normalizes), post_neg
    // puts them back on the digits; body_only: only the body column has an operand (phi(body), from the workspace)
    int post_neg, body_only;
    // raw: store the rounded coefficients themselves (a VecZnxBig), no carry chain (launch_inv_tail_raw)
    int raw;
    // nz (GLWE tensoring, round 3): the values leave through the SAME-BASE steps of vec_znx_normalize with a bit offset (normalize.rs:50-144 as
    // k_normalize_inter walks them: carry-only steps for limbs >= nz_a_start, digit steps down to nz_a_end, then nz_res_end steps on the
    // carry alone) into column nz_col of `res` (mode nz_mode) and up to two further columns (NzCombine's modes: 1 = v, 2 = -v, 3 += v,
    // 4 -= v; 5 on BOTH further columns: their digits are read and subtracted from the value before it reaches nz_col); res limbs >= nz_zero_from are zero.  The limb count of the transformed value may be smaller than the normalizer's a.size:
    // the missing top-index limbs are zeros and come first in the chain, where they change nothing.
    int nz, nz_lsh, nz_res_end, nz_res_start, nz_a_end, nz_a_start, nz_zero_from, nz_col, nz_mode, nz_col2[2], nz_mode2[2];
    // 16-bit side copies of the diagonal terms' digits (round 6, TailD16 in internal.hpp; base2k <= 16): [pair][res limb][n] int16, a limb in the
    // tail's tile order - workgroup block (c0 / CB), then (half h, row j1), then the CB columns: element idx = h m + j1 m2 + c0 + c sits at
    // (c0 / CB) 2 M1 CB + (idx >> log2 m2) CB + c, so that a wave's store / load of one (h, n1) is one 128-byte run.
    // d16w: the NZ = 1 launch mirrors every digit it stores; d16a / d16b: the NZ = 2 launch's mode-5 prefetch reads them instead of the low
    // dwords of the two i64 columns
    short* d16w;
    const short *d16a, *d16b;
    // body16_wide != null: the body-column operand (+-phi(body), + - a0 in the add / sub forms) was left by the automorphism pre-pass as 16-bit values in the
    // tile order, d16a[ciphertext][limb][n] with body_bs int16 elements per ciphertext, and *body16_wide says whether a value did not fit.  The 16-bit-operand
    // form (NZF = 7) reads the copies and returns at once if the flag is up; an operand form (SMALL) launched with this pointer returns at once if it is down
    const unsigned* body16_wide;
};
__device__ __forceinline__ long long tz_digit(int k, long long x) { return (long long)((unsigned long long)x << (64 - k)) >> (64 - k); }
__device__ __forceinline__ long long tz_carry(int k, long long x, long long d) { return (long long)((unsigned long long)x - (unsigned long long)d) >> k; }
// (modes 1 / 2 are written once and read by a LATER launch at the earliest: streaming stores, as the product tails - with plain stores the
//  PMC write traffic of the tensoring tail was 1.45 x its bytes, profiles/r04_tensor_traffic.json)
__device__ __forceinline__ void tz_put(long long* p, int mode, long long v) {
    if (mode == 1) st_stream(p, v);
    else if (mode == 2) st_stream(p, (long long)(0ull - (unsigned long long)v));
    else if (mode == 3) *p = (long long)((unsigned long long)*p + (unsigned long long)v);
    else if (mode == 4) *p = (long long)((unsigned long long)*p - (unsigned long long)v);
}

// Workgroup = (R2 + R1)*CB threads in two wave-uniform roles (R2*CB must be a multiple of 64):
//   B' waves (tid <  R2*CB): second butterfly stage of limb j, rounding, carry chain, stores;
//   A' waves (tid >= R2*CB): loads + first butterfly stage of limb j-1 into the other half of the
//                            double-buffered exchange, and the loads of limb j-2 behind that.
// The two roles run concurrently (one barrier per limb), so the HBM latency of the next limbs hides
// behind the arithmetic of the current one, and each role only pays for its own registers.
// R1 = 16 (m1 = 256): the B' role is SPLIT over two threads per column position — the last radix-16 butterfly is one
// radix-2 step (sum half / twiddled difference half, both read all 16 inputs from the exchange buffer) followed by a
// radix-8 butterfly, so a thread owns 8 outputs and 16 carries like in the R1 = 8 plans instead of 16 and 32 (which spills).
template <int R1, int R2, int CB>
struct TailShape {
    static constexpr bool SPLIT = R1 == 16;
    static constexpr int RE = SPLIT ? 8 : R1;                   // outputs per B' thread
    static constexpr int NB = (SPLIT ? 2 : 1) * R2 * CB;        // B' threads
    static constexpr int NT = NB + R1 * CB;                     // + A' threads
};
// RSH (with SMALL): the digits leave the carry chain through vec_znx_rsh_assign by ONE bit (reference/vec_znx/shift.rs:186-243, the
// steps of znx/normalization.rs with lsh = base2k - 1) before they are stored - glwe_trace shifts its ciphertext between two
// automorphisms (glwe_trace.rs:164-166), and that walk runs in the same direction as the chain: the last limb of res only leaves a
// carry, the digit of every other limb j gives limb j + 1 of the shifted value, limb 0 is the digit of the last carry.  One read and
// one write of the ciphertext less per trace step.
// (32-bit: the digits that enter are balanced base2k-bit values, so every intermediate of the reference's i64 steps stays below
//  2^(base2k + 1) - the launcher requires base2k <= 29)
__device__ __forceinline__ int sx_digit(int k, int x) { return (int)((unsigned)x << (32 - k)) >> (32 - k); }
__device__ __forceinline__ int sx_carry(int k, int x, int d) { return (x - d) >> k; }
// NZ: the tensoring forms (TailArgs::raw / ::nz) are compiled in - a separate instantiation, so that the product tails keep their registers
// (with the two run-time modes in the common kernel the N = 2^16 tail spilled 84 bytes)
// (NZ = 2: every NzCombine mode, with the prefetch of the diagonal digits for mode 5 - the pairwise launch; NZ = 1: digits stored as they are
//  (mode 1, no second column: the diagonal launches of apply / square) - spared the prefetch's 32 registers (88 bytes of scratch otherwise)
//  and the tests in front of every store)
// SGN (round 4): the signs of X -> X^p (TailArgs::auto_mul / auto_neg / post_neg) without an operand - the columns of a plain spectral
// glwe_automorphism that carry no body: they ride on the f64 chain like a product's columns instead of the operand variant's integer chain
// ACC32 (round 5; with SMALL, every column its own operand): the blind rotation's accumulator between two blocks of the pipeline path - operand and /
// or result as 32-bit digits (TailArgs::acc32), half the bytes of the two streams this kernel moves beside T2'
template <int R1, int R2, int CB, bool ROWMAJOR = false, bool SMALL = false, bool RSH = false, int NZF = 0, bool SGN = false, bool PROBE = false, bool ACC32 = false>
__global__ void __launch_bounds__(((R1 == 16 ? 2 : 1) * R2 + R1) * CB, ((((R1 == 16 ? 2 : 1) * R2 + R1) * CB >= 512) ? 1 : (SMALL ? 2 : 3)))
k_inv_tail(TailArgs g) {
    static_assert(!RSH || SMALL || NZF == 7, "the shifted store rides on the operand forms (the integer chain, or the 16-bit operand on the f64 chain)");
    // NZF: 0 none | 1 diagonal tensoring tail | 2 pairwise / raw | 3 = 1 + 16-bit side copy of every digit (TailArgs::d16w) | 4 = 2 with the mode-5
    // prefetch reading 16-bit side copies (TailArgs::d16a / d16b) instead of the i64 columns' low dwords.  Forms of their own: with both prefetch
    // paths in one instantiation the pairwise tail went from 12 to 28 B of scratch at its 168-register cap
    // 5 / 6 = 3 / 4 whose digits leave ONLY as 16-bit copies (the fused multiply + relinearize never materializes the i64 GLWETensor; 6 writes the
    // pairwise column's values pair - d_i - d_j, which fit 16 bits while base2k <= 14)
    // 7 (with SGN): not a tensoring form - the sign-only tail with a 16-BIT OPERAND in front of the chain (OP16: the body column of a plain spectral
    // automorphism whose pre-pass left phi(body) as 16-bit values in the tile order, TailArgs::body16_wide); rides on the f64 chain like the operand-free
    // columns - the operand variant's integer chain ran that column at 1.3 x the time of a product column
    constexpr bool OP16 = NZF == 7;
    constexpr int NZ = OP16 ? 0 : ((NZF == 3 || NZF == 5) ? 1 : ((NZF == 4 || NZF == 6) ? 2 : NZF));
    static_assert(!OP16 || SGN, "the 16-bit operand rides on the sign-only form");
    constexpr bool D16W = NZF == 3 || NZF >= 5, D16R = NZF == 4 || NZF == 6, D16ONLY = NZF >= 5;
    // the side-copy forms are only ever launched as normalizing tails (launch_tail.hip: TailArgs::nz set, never raw): what the run-time tests below
    // would keep compiled in beside them - the plain chain, the raw store - costs registers and issue slots in the limb loop
    constexpr bool NZD = PZ_TAIL_NZD && (D16W || D16R);
#define PZ_TAIL_IS_NZ (NZ && (NZD || g.nz))
#define PZ_TAIL_IS_RAW (NZ && !NZD && g.raw)
    static_assert(!ACC32 || (ROWMAJOR && SMALL && !RSH && !NZ && !SGN), "32-bit accumulator digits: the plain operand form of the row-major pipeline");
    static_assert(!NZ || (ROWMAJOR && !SMALL && !RSH), "tensoring forms: row-major pipeline layout, no operand");
    static_assert(!SGN || (ROWMAJOR && !SMALL && (!RSH || NZF == 7) && !NZ), "sign-only form: row-major pipeline layout, no operand");
    constexpr bool SPLIT = TailShape<R1, R2, CB>::SPLIT;
    constexpr int RE = TailShape<R1, R2, CB>::RE;
    constexpr int NB = TailShape<R1, R2, CB>::NB;
    constexpr int NT = TailShape<R1, R2, CB>::NT;
    constexpr int M1 = R1 * R2;
    constexpr int XCH = (R1 + 1) * CB * R2;  // cplx per exchange buffer
    extern __shared__ cplx xch[];            // 2 * XCH | wL1[M1] | tw1inv[M1]
    // stage roots and the untwist table (both M1 entries) in LDS: a global gather in front of dependent
    // arithmetic costs a memory latency per use at this occupancy
    cplx* wl = xch + 2 * XCH;
    cplx* twi = wl + M1;
    const int tid = threadIdx.x;
    for (int t = tid; t < M1; t += NT) {
        wl[t] = g.wL1[t];
        twi[t] = g.tw1inv[t];
    }
    __syncthreads();
    // the two launches that share the body column of a plain spectral automorphism: the 16-bit-operand form does the work unless the pre-pass found a
    // value that did not fit (flag up), the gathering operand form only then (launch_inv_tail)
    if (OP16 && __builtin_amdgcn_readfirstlane((int)*g.body16_wide) != 0) return;
    if (SMALL && g.body16_wide && __builtin_amdgcn_readfirstlane((int)*g.body16_wide) == 0) return;
    const int ncb = g.m2 / CB;
    int bid = blockIdx.x;
    if (g.xcd_map) {  // gridDim.x = ceil(nbc / 8) * 8 * ncb, nbc = (ciphertext, column) pairs of this launch = g.xcd_map
        const int xcd = bid & 7, slot = bid >> 3;
        const int bcx = (slot / ncb) * 8 + xcd;
        if (bcx >= g.xcd_map) return;
        bid = bcx * ncb + slot % ncb;
    }
    const int c0 = (bid % ncb) * CB;
    const int bc = bid / ncb;
    const int col = g.col_base + bc % g.col_count;
    const int b = bc / g.col_count;
    const long long m = (long long)M1 * g.m2;
    const long long n = 2 * m;
    const int L = g.nlimbs;

    if (tid >= NB) {
        // ------------------------------ role A' ------------------------------
        const int ta = tid - NB;
        const int k1 = ROWMAJOR ? ta / CB : ta % R1;
        const int c = ROWMAJOR ? ta % CB : ta / R1;
        // element (q1 = k1 + R1*k2, j2 = c0 + c): T[j2][q1] or, ROWMAJOR, T'[q1][j2]
        const cplx* Tb = g.T + ((long long)b * L * g.ncols + col) * m +
                         (ROWMAJOR ? (long long)k1 * g.m2 + c0 + c : (long long)(c0 + c) * M1 + k1);
        const long long k2s = ROWMAJOR ? (long long)R1 * g.m2 : (long long)R1;  // stride of k2
        const long long limb_stride = (long long)g.ncols * m;
        cplx u[R2];
        if (L > 0) {
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) u[k2] = ROWMAJOR ? ld_stream(Tb + (long long)(L - 1) * limb_stride + k2s * k2) : Tb[(long long)(L - 1) * limb_stride + k2s * k2];
        }
        // iteration t produces limb L-1-t into buffer (L-1-t)&1; t = 0 is the prologue
        for (int t = 0; t <= L; ++t) {
            const int j = L - 1 - t;  // limb produced in this iteration (none when j < 0)
            if (j >= 0) {
                Bfly<R2, true>::run(u);
                cplx* buf = xch + (j & 1) * XCH;
#pragma unroll
                for (int o = 0; o < R2; ++o) {
                    cplx x = u[o];
                    if (R2 > 1 && k1 > 0 && o > 0) x = cmulc(x, wl[o * k1]);
                    buf[(o * CB + c) * (R1 + 1) + k1] = x;
                }
                if (j > 0) {
#pragma unroll
                    for (int k2 = 0; k2 < R2; ++k2) u[k2] = ROWMAJOR ? ld_stream(Tb + (long long)(j - 1) * limb_stride + k2s * k2) : Tb[(long long)(j - 1) * limb_stride + k2s * k2];
                }
            }
            __syncthreads();
        }
        return;
    }
    // ------------------------------ role B' ------------------------------
    const int k = g.base2k;
    const int hs = SPLIT ? tid / (R2 * CB) : 0;                  // which half of the last butterfly (wave-uniform)
    const int tb = SPLIT ? tid - hs * (R2 * CB) : tid;
    const int b_o = tb / CB, b_c = tb % CB;
#define PZ_TAIL_N1(E) (SPLIT ? 2 * (E) + hs : (E))              /* butterfly output index of this thread's E-th value */
    long long carry[2 * RE];
#pragma unroll
    for (int u = 0; u < 2 * RE; ++u) carry[u] = 0;
    // Without a body operand (external product) and digits of at most 31 bits the carry chain runs in f64 while every value is an exact
    // integer below 2^51 (the same test that allows the 3-instruction conversion): v = r + carry, q = floor((v + 2^(k-1)) 2^-k),
    // digit = v - q 2^k — the reference's two steps in one (digit(x + c) = digit(digit(x) + c), carry(x + c) = carry(x) +
    // carry(digit(x) + c) while nothing overflows), five f64 operations instead of ~25 32-bit integer ones.  The carries then hold the
    // bits of a double; the first limb that needs the saturating path converts them once and the thread stays on the integer chain.
    constexpr bool FCARRY = !SMALL;
    bool icarry = !FCARRY || k > 31;
    const double halfd = (double)(1ull << (k - 1)), twok = 2.0 * halfd, invk = 1.0 / twok;
    long long* res_col = g.res + (long long)b * g.res_bs + (long long)(PZ_TAIL_IS_NZ ? g.nz_col : col) * n;
    const long long res_ls = (long long)g.res_cols * n;
    long long* nz_r2a = (NZ && g.nz && g.nz_mode2[0]) ? g.res + (long long)b * g.res_bs + (long long)g.nz_col2[0] * n : nullptr;
    long long* nz_r2b = (NZ && g.nz && g.nz_mode2[1]) ? g.res + (long long)b * g.res_bs + (long long)g.nz_col2[1] * n : nullptr;
    // 16-bit side copies (TailArgs::d16*): this workgroup's tile of limb r starts at (b res_size + r) n + (c0 / CB) 2 M1 CB
    const long long d16_base = (long long)b * g.res_size * n + (long long)(c0 / CB) * (2 * M1 * CB);
    // (the thread's E-th value, half H, of limb R: one base per thread and constant offsets - the tile order is this kernel's own)
#if PZ_TAIL_D16_ADDR
    const long long d16_t = d16_base + (b_o * CB + b_c) + (SPLIT ? hs * R2 * CB : 0);
#define PZ_TAIL_D16_AT(R_, E_, H_, IDX_) (d16_t + (long long)(R_) * n + ((H_) * M1 * CB + (SPLIT ? 2 : 1) * R2 * CB * (E_)))
#else   // the same element from the coefficient index (recomputed at every store: nothing of it stays live across the limb)
    const int d16_sh = 31 - __builtin_clz((unsigned)g.m2);
#define PZ_TAIL_D16_AT(R_, E_, H_, IDX_) (d16_base + (long long)(R_) * n + (long long)((((IDX_) >> d16_sh) * CB) + ((IDX_) & (CB - 1))))
#endif
#define PZ_TAIL_NZ_STORE(R_, IDX_, V_, E_, H_)                                                     \
    {                                                                                              \
        const long long off_ = (long long)(R_) * res_ls + (IDX_);                                  \
        long long v_ = (V_);                                                                       \
        if (NZ == 1) {   /* the launcher sends only plain stores here (mode 1, no second column): no test in front of the store */ \
            if (!D16ONLY) st_stream(res_col + off_, v_);                                           \
            if (D16W) g.d16w[PZ_TAIL_D16_AT(R_, E_, H_, IDX_)] = (short)v_;                              \
        } else if (D16ONLY) {   /* (the zero limbs and the carry-only top limbs of the pairwise column; its digit limbs leave through the d5 store) */ \
            const long long at_ = PZ_TAIL_D16_AT(R_, E_, H_, IDX_);                                      \
            g.d16w[at_] = (short)((int)v_ - (int)g.d16a[at_] - (g.d16b ? (int)g.d16b[at_] : 0));   \
        } else if (g.nz_mode2[0] == 5) {   /* mode 5: a column whose digits are SUBTRACTED from the value on its way to the main column */ \
            v_ = (long long)((unsigned long long)v_ - (unsigned long long)nz_r2a[off_]);           \
            if (nz_r2b) v_ = (long long)((unsigned long long)v_ - (unsigned long long)nz_r2b[off_]); \
            tz_put(res_col + off_, g.nz_mode, v_);                                                 \
        } else {                                                                                   \
            tz_put(res_col + off_, g.nz_mode, v_);                                                 \
            if (nz_r2a) tz_put(nz_r2a + off_, g.nz_mode2[0], v_);                                  \
            if (nz_r2b) tz_put(nz_r2b + off_, g.nz_mode2[1], v_);                                  \
        }                                                                                          \
    }
    const long long* small_col =
        (g.small && (col == g.body_col || g.small_all)) ? g.small + (long long)b * g.small_bs + (g.small_all ? (long long)col * n : 0) : nullptr;
    const long long small_ls = (long long)g.small_cols * n;
    // limbs of res beyond the precision of the big value are zero (normalize.rs:118-120); RSH: limb L receives the bit shifted out of
    // limb L - 1
    int cy2[RSH ? 2 * RE : 1];
#pragma unroll
    for (int u = 0; u < (RSH ? 2 * RE : 1); ++u) cy2[u] = 0;
    for (int j = PZ_TAIL_IS_NZ ? g.nz_zero_from : L + (RSH ? 1 : 0); j < g.res_size; ++j)
#pragma unroll
        for (int e = 0; e < RE; ++e) {
            const int n1 = PZ_TAIL_N1(e);
            const long long idx = (long long)(b_o + R2 * n1) * g.m2 + c0 + b_c;
            if (PZ_TAIL_IS_NZ) {
                PZ_TAIL_NZ_STORE(j, idx, 0, e, 0)
                PZ_TAIL_NZ_STORE(j, idx + m, 0, e, 1)
            } else if (ACC32 && (g.acc32 & 16)) {   // (16-bit tile-order result, same element offsets: see the store below)
                short* r16 = reinterpret_cast<short*>(g.res) + (res_col - g.res) + (long long)j * res_ls + (long long)(c0 / CB) * (2 * M1 * CB) +
                             (b_o * CB + b_c) + (SPLIT ? hs * R2 * CB : 0) + (SPLIT ? 2 : 1) * R2 * CB * e;
                r16[0] = 0;
                r16[M1 * CB] = 0;
            } else if (ACC32 && (g.acc32 & 2)) {
                int* r32 = reinterpret_cast<int*>(g.res) + (res_col - g.res);
                r32[(long long)j * res_ls + idx] = 0;
                r32[(long long)j * res_ls + idx + m] = 0;
            } else {
                res_col[(long long)j * res_ls + idx] = 0;
                res_col[(long long)j * res_ls + idx + m] = 0;
            }
        }
    __syncthreads();  // matches the prologue iteration of role A'
    for (int j = L - 1; j >= 0; --j) {
        // (SMALL: the per-output address arithmetic is loop-invariant and would be hoisted out of the limb loop into ~32
        //  live registers; an opaque copy of the lane coordinates per limb makes the compiler recompute it instead)
        int b_ov = b_o, b_cv = b_c;
        if (SMALL || NZ || SGN) asm volatile("" : "+v"(b_ov), "+v"(b_cv));
        const cplx* buf = xch + (j & 1) * XCH;
        // NZ, mode 5 (the pairwise term of a tensoring reads the two diagonal columns' digits at its own store position and subtracts them):
        // requested HERE, at the top of the limb, so that their latency hides behind the butterfly - read at the store they were 32 dependent
        // HBM loads per thread and limb, and the pairwise launch ran at 2.8 TB/s against 4.6 for the diagonal ones (profiles/r04_tensor_*).
        // Low dwords only: the values are balanced base2k-bit digits (k <= 31 here; wider digits keep the loads at the store).
        int d5a[NZ == 2 ? 2 * RE : 1], d5b[(NZ == 2 && !D16R) ? 2 * RE : 1];
        pz_short2 d5p[D16R ? 2 * RE : 1];
        bool d5 = false;
        if (D16R && (NZD || (g.nz && g.nz_mode2[0] == 5)) && j >= g.nz_a_end && j < g.nz_a_start) {
            // the diagonal launches left 16-bit copies of their digits (TailArgs::d16*, base2k <= 16): 2 B per coefficient and column, in this
            // workgroup's own tile order (one base per thread, constant offsets), instead of the 8 B a line of the i64 column costs
            d5 = true;
            const long long rb_ = d16_base + (long long)(j - g.nz_a_start + g.nz_res_start) * n + (b_ov * CB + b_cv) + (SPLIT ? hs * R2 * CB : 0);
            const short* pa_ = g.d16a + rb_;
            const short* pb_ = g.d16b ? g.d16b + rb_ : pa_;   // (one diagonal column only: read twice, halved below - never dispatched today)
#pragma unroll
            for (int e = 0; e < RE; ++e)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int o_ = h * M1 * CB + (SPLIT ? 2 : 1) * R2 * CB * e;
                    pz_short2 v_;   // the two digits of a coefficient share one register (d16 loads into the low / high half)
                    v_.x = pa_[o_];
                    v_.y = pb_[o_];
                    d5p[D16R ? 2 * e + h : 0] = v_;
                }
        } else
        if (NZ == 2 && !D16R && g.nz && g.nz_mode2[0] == 5 && k <= 31 && j >= g.nz_a_end && j < g.nz_a_start) {
            d5 = true;
            const long long r5 = (long long)(j - g.nz_a_start + g.nz_res_start) * res_ls;
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                const long long idx = (long long)(b_ov + R2 * PZ_TAIL_N1(e)) * g.m2 + c0 + b_cv;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const long long off5 = r5 + idx + (h ? m : 0);
                    d5a[NZ == 2 ? 2 * e + h : 0] = reinterpret_cast<const int*>(nz_r2a + off5)[0];
                    d5b[NZ == 2 ? 2 * e + h : 0] = nz_r2b ? reinterpret_cast<const int*>(nz_r2b + off5)[0] : 0;
                }
            }
        }
        // OP16: this limb of the 16-bit operand (tile order: one base, constant offsets), requested first like the body limb below
        pz_short2 op16[OP16 ? RE : 1];   // (.x: the coefficient below m, .y: the one above - two 16-bit loads into one register)
        if constexpr (OP16) {
            if (j < g.small_size) {
                const short* s16 = g.d16a + (long long)b * g.body_bs + (long long)j * n + (long long)(c0 / CB) * (2 * M1 * CB) + (b_ov * CB + b_cv) +
                                   (SPLIT ? hs * R2 * CB : 0);
#pragma unroll
                for (int e = 0; e < RE; ++e) {
                    pz_short2 v_;
                    v_.x = s16[(SPLIT ? 2 : 1) * R2 * CB * e];
                    v_.y = s16[M1 * CB + (SPLIT ? 2 : 1) * R2 * CB * e];
                    op16[OP16 ? e : 0] = v_;
                }
            } else {
#pragma unroll
                for (int t = 0; t < RE; ++t) { op16[OP16 ? t : 0].x = 0; op16[OP16 ? t : 0].y = 0; }
            }
        }
        // key-switch body limb: requested first so that its latency hides behind the butterfly
        long long sm[SMALL ? 2 * RE : 1];
        if (SMALL && small_col && j < g.small_size && g.pre_body) {
            // body column: phi(body) (+ a0) either prepared by k_automorphism in the workspace (body_src) or, gather_mul != 0, gathered here
            // from the body itself (column 0 of `small`) - the pre-pass and its round trip through HBM are gone, the 8-byte gathers are
            // served by the XCD's L2 (all column blocks of one ciphertext column run on one XCD)
            // (one straight-line loop per form, chosen here: with the form tested per element - a branch around the second load - the compiler
            //  waits for every load on the spot, 32 exposed latencies per limb: that, not the gathers, is what the first builds of the
            //  gathered and of the two-stream form measured, 8.3 - 8.8 ms against 5.05)
            const bool isbody = col == g.body_col;
            const long long* bsrc = (isbody && !g.gather_mul) ? g.body_src + (long long)b * g.body_bs + (long long)j * g.body_ls : nullptr;
            const long long* gsrc = (isbody && g.gather_mul) ? g.small + (long long)b * g.small_bs + (long long)j * small_ls : nullptr;
            const long long* scol = small_col + (long long)j * small_ls;
#define PZ_TAIL_OPERAND(EXPR_)                                                                      \
    _Pragma("unroll") for (int e = 0; e < RE; ++e) {                                                \
        const long long idx = (long long)(b_ov + R2 * PZ_TAIL_N1(e)) * g.m2 + c0 + b_cv;             \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                             \
            const long long ih = idx + (h ? m : 0);                                                 \
            unsigned long long v = (EXPR_);                                                         \
            if (g.small_neg) v = 0ull - v;                                                          \
            sm[2 * e + h] = (long long)v;                                                           \
        }                                                                                           \
    }
#define PZ_TAIL_GATHERED(IH_) ([&]() -> unsigned long long {                                        \
        const unsigned i0 = ((unsigned)(IH_) * g.gather_mul) & (unsigned)(2 * n - 1);               \
        const unsigned long long w = (unsigned long long)gsrc[(long long)(i0 & (unsigned)(n - 1))]; \
        return ((i0 >= (unsigned)n) != (g.gather_neg != 0)) ? 0ull - w : w; }())
            if constexpr (RSH) {   // (the shifted-store variant sits at the register cap: one loop, the source picked by a select - round 3's form)
                PZ_TAIL_OPERAND(bsrc ? (unsigned long long)bsrc[ih] : (g.body_only ? 0ull : (unsigned long long)scol[ih]))
            } else

            if (!RSH && bsrc && g.body_add) PZ_TAIL_OPERAND((unsigned long long)bsrc[ih] + (unsigned long long)scol[ih])   /* (not in the shifted-store variant: no registers left there) */
            else if (bsrc) PZ_TAIL_OPERAND((unsigned long long)bsrc[ih])
            else if (!RSH && gsrc && g.body_only) PZ_TAIL_OPERAND(PZ_TAIL_GATHERED(ih))   /* (the host never asks the shifted-store variant for the gathered or two-stream forms) */
            else if (!RSH && gsrc) PZ_TAIL_OPERAND((unsigned long long)scol[ih] + PZ_TAIL_GATHERED(ih))
            else if (g.body_only) PZ_TAIL_OPERAND(0ull)
            else PZ_TAIL_OPERAND((unsigned long long)scol[ih])
#undef PZ_TAIL_GATHERED
#undef PZ_TAIL_OPERAND
        } else if (SMALL && small_col && j < g.small_size && g.gather_mul) {
            const long long* body = col == g.body_col ? g.small + (long long)b * g.small_bs + (long long)j * small_ls : nullptr;
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                const long long idx = (long long)(b_ov + R2 * PZ_TAIL_N1(e)) * g.m2 + c0 + b_cv;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned ih = (unsigned)(idx + (h ? m : 0));
                    const unsigned i0 = (ih * g.gather_mul) & (unsigned)(2 * n - 1);
                    unsigned long long v = (unsigned long long)small_col[(long long)j * small_ls + (long long)(i0 & (unsigned)(n - 1))];
                    if ((i0 >= (unsigned)n) != (g.gather_neg != 0)) v = 0ull - v;
                    if (body) v += (unsigned long long)body[ih];
                    sm[2 * e + h] = (long long)v;
                }
            }
        } else if (ACC32 && (g.acc32 & 4) && small_col && j < g.small_size) {
            // the operand is a GLWETensor column that exists only as 16-bit digits in this kernel's own tile order (fused multiply + relinearize,
            // api_cnv.hip): d16a[column][ciphertext][limb][n], body_bs = int16 elements between columns
            const short* s16 = g.d16a + (long long)col * g.body_bs + ((long long)b * g.small_size + j) * n + (long long)(c0 / CB) * (2 * M1 * CB) +
                               (b_ov * CB + b_cv) + (SPLIT ? hs * R2 * CB : 0);
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                const int o_ = (SPLIT ? 2 : 1) * R2 * CB * e;
                sm[SMALL ? 2 * e : 0] = (long long)s16[o_];
                sm[SMALL ? 2 * e + 1 : 0] = (long long)s16[M1 * CB + o_];
            }
        } else if (ACC32 && (g.acc32 & 8) && small_col && j < g.small_size) {
            // the operand as 16-bit digits in the tile order at the SAME element offsets as the i64 container (the blind rotation's accumulator
            // between two blocks, base2k <= 15: api_br.hip)
            const short* s16 = reinterpret_cast<const short*>(g.small) + (small_col - g.small) + (long long)j * small_ls + (long long)(c0 / CB) * (2 * M1 * CB) +
                               (b_ov * CB + b_cv) + (SPLIT ? hs * R2 * CB : 0);
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                const int o_ = (SPLIT ? 2 : 1) * R2 * CB * e;
                sm[SMALL ? 2 * e : 0] = (long long)s16[o_];
                sm[SMALL ? 2 * e + 1 : 0] = (long long)s16[M1 * CB + o_];
            }
        } else if (ACC32 && (g.acc32 & 1) && small_col && j < g.small_size) {
            const int* s32 = reinterpret_cast<const int*>(g.small) + (small_col - g.small) + (long long)j * small_ls;
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                const long long idx = (long long)(b_ov + R2 * PZ_TAIL_N1(e)) * g.m2 + c0 + b_cv;
                sm[SMALL ? 2 * e : 0] = (long long)s32[idx];
                sm[SMALL ? 2 * e + 1 : 0] = (long long)s32[idx + m];
            }
        } else if (SMALL && small_col && j < g.small_size) {
#pragma unroll
            for (int e = 0; e < RE; ++e) {
                const long long idx = (long long)(b_ov + R2 * PZ_TAIL_N1(e)) * g.m2 + c0 + b_cv;
                sm[2 * e] = small_col[(long long)j * small_ls + idx];
                sm[2 * e + 1] = small_col[(long long)j * small_ls + idx + m];
            }
        } else if (SMALL) {
#pragma unroll
            for (int t = 0; t < 2 * RE; ++t) sm[t] = 0;
        }
        cplx v[RE];
        if (SPLIT) {
            // first step of the inverse radix-16 butterfly: this thread keeps the sums (hs = 0: even outputs) or the
            // twiddled differences (hs = 1: odd outputs), then a radix-8 butterfly
            const cplx* in = buf + (b_ov * CB + b_cv) * (R1 + 1);
            if (hs == 0) {
#pragma unroll
                for (int n = 0; n < RE; ++n) v[n] = cadd(in[n], in[n + R1 / 2]);
            } else {
#pragma unroll
                for (int n = 0; n < RE; ++n) v[n] = tw_small<R1, true>(csub(in[n], in[n + R1 / 2]), n);
            }
        } else {
#pragma unroll
            for (int k1 = 0; k1 < RE; ++k1) v[k1] = buf[(b_ov * CB + b_cv) * (R1 + 1) + k1];
        }
        Bfly<RE, true>::run(v);
        const bool writes = j < g.res_size;
        const bool first = j == L - 1;
        const bool add_small = small_col && j < g.small_size;
        // scale/untwist in place and bound the magnitudes: below 2^51 the 3-instruction conversion is exact
        double big = 0.0;
#pragma unroll
        for (int e = 0; e < RE; ++e) {
            v[e] = cmul(v[e], twi[b_ov + R2 * PZ_TAIL_N1(e)]);
            big = fmax(big, fmax(fabs(v[e].x), fabs(v[e].y)));
        }
        if (PROBE) {   // rounding-margin probe (margin_note): every value this limb rounds, in every form of the tail.  Compile-time here (its own
                       // instantiation of each form): a run-time test in front of the carry loop cost the operand forms 10 - 45 registers (200 B of
                       // scratch at N = 2^16), the block boundary keeps the operand prefetch from being scheduled across it
            double worst = 0.0;
#pragma unroll
            for (int e = 0; e < RE; ++e) worst = fmax(worst, fmax(margin_dist(v[e].x), margin_dist(v[e].y)));
            margin_note(g.margin, worst);
        }
        if (NZ == 2 && d5) {   // the two diagonal digits are only ever used as their sum: one register per coefficient from here on (the
                               // pairwise instantiation's 88 B of scratch were 2.7 GB of extra HBM writes per launch, profiles/r04_tensor_traffic.json)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < (NZ == 2 ? 2 * RE : 1); ++t) {
                if constexpr (D16R) d5a[t] = (int)d5p[D16R ? t : 0].x + (g.d16b ? (int)d5p[D16R ? t : 0].y : 0);
                else d5a[t] += d5b[(NZ == 2 && !D16R) ? t : 0];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long half = 1ull << (k - 1), mask = (1ull << k) - 1;
        // One pass per coefficient: round, convert, (+ body), carry step, store.  digit(x) = ((x + 2^(k-1)) mod 2^k) -
        // 2^(k-1), carry(x) = (x + 2^(k-1)) >> k: the values of the reference's shift pairs
        // (reference/znx/normalization.rs:4-11,24-41,107-129,179-221) with fewer 64-bit operations.
        // (RSH) the digit leaves through vec_znx_rsh_assign by one bit: limb j's digit gives limb j + 1 of the shifted value
#define PZ_TAIL_RSH_STORE(X1_)                                                                               \
    {                                                                                                        \
        int& c2 = cy2[RSH ? 2 * n1 + h : 0];                                                                 \
        const int xd = (int)(X1_);                                                                           \
        const int d1 = -(xd & 1);                                                                            \
        const int cr1 = (xd - d1) >> 1;                                                                      \
        if (j == g.res_size - 1) {                                                                           \
            c2 = cr1;                                                                                        \
        } else {                                                                                             \
            const int dpc = d1 * (1 << (k - 1)) + c2;                                                        \
            const int nv = sx_digit(k, dpc);                                                                 \
            c2 = cr1 + sx_carry(k, dpc, nv);                                                                 \
            st_stream(res_col + (long long)(j + 1) * res_ls + idx, (long long)nv);                           \
        }                                                                                                    \
        if (j == 0) st_stream(res_col + idx, (long long)sx_digit(k, c2));                                    \
    }
#define PZ_TAIL_COEFFS(CONVERT)                                                                               \
    _Pragma("unroll") for (int n1 = 0; n1 < RE; ++n1) {                                                      \
        const int j1 = b_ov + R2 * PZ_TAIL_N1(n1);                                                            \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                      \
            const long long idx = (long long)j1 * g.m2 + c0 + b_cv + (h ? m : 0);                             \
            const double val = h ? v[n1].y : v[n1].x;                                                        \
            const double r = round_half_away(val);                                                           \
            long long x = CONVERT(r);                                                                        \
            if (SMALL && add_small) x = (long long)((unsigned long long)x + (unsigned long long)sm[SMALL ? 2 * n1 + h : 0]); \
            if (OP16) { const long long op_ = (long long)(h ? op16[OP16 ? n1 : 0].y : op16[OP16 ? n1 : 0].x); x = (long long)((unsigned long long)x + (unsigned long long)(g.small_neg ? -op_ : op_)); }   \
            bool ng_ = false;                                                                                \
            if ((SMALL || SGN) && g.auto_mul) {                                                              \
                ng_ = (((unsigned)idx * g.auto_mul) & (unsigned)(2 * n - 1)) >= (unsigned)n;                 \
                if (ng_ != (g.auto_neg != 0)) x = (long long)(0ull - (unsigned long long)x);                 \
            }                                                                                                \
            if (PZ_TAIL_IS_RAW) {                                                                            \
                if (writes) { if (ROWMAJOR) st_stream(res_col + (long long)j * res_ls + idx, x); else res_col[(long long)j * res_ls + idx] = x; } \
                continue;                                                                                    \
            }                                                                                                \
            if (PZ_TAIL_IS_NZ) {   /* k_normalize_inter's steps on limb j (carry starts at 0: its first-step special case is the general step) */ \
                if (j >= g.nz_a_end) {                                                                       \
                    long long& c_ = carry[2 * n1 + h];                                                       \
                    const int kk_ = g.nz_lsh == 0 ? k : k - g.nz_lsh;                                        \
                    const long long d_ = tz_digit(kk_, x);                                                   \
                    const long long cr_ = tz_carry(kk_, x, d_);                                              \
                    const long long dpc_ = (long long)(((unsigned long long)d_ << g.nz_lsh) + (unsigned long long)c_); \
                    const long long x1_ = tz_digit(k, dpc_);                                                 \
                    if (j < g.nz_a_start) {                                                                  \
                        if (d5 && D16ONLY) g.d16w[PZ_TAIL_D16_AT(j - g.nz_a_start + g.nz_res_start, n1, h, idx)] = (short)((int)x1_ - d5a[NZ == 2 ? 2 * n1 + h : 0]); \
                        else if (d5) tz_put(res_col + (long long)(j - g.nz_a_start + g.nz_res_start) * res_ls + idx, g.nz_mode,  \
                                       (long long)((unsigned long long)x1_ - (unsigned long long)(long long)d5a[NZ == 2 ? 2 * n1 + h : 0])); \
                        else if constexpr (!D16R) PZ_TAIL_NZ_STORE(j - g.nz_a_start + g.nz_res_start, idx, x1_, n1, h)   /* (D16R: every digit limb has its prefetch) */ \
                    }                                                                                        \
                    c_ = (long long)((unsigned long long)cr_ + (unsigned long long)tz_carry(k, dpc_, x1_));  \
                }                                                                                            \
                continue;                                                                                    \
            }                                                                                                \
            long long& cy = carry[2 * n1 + h];                                                               \
            const unsigned long long y = (unsigned long long)x + half;                                       \
            const long long d = (long long)(y & mask) - (long long)half;                                     \
            const long long cr = (long long)y >> k;                                                          \
            if (first && !writes) {                                                                          \
                cy = cr;                                                                                     \
            } else {                                                                                         \
                const unsigned long long y2 = (unsigned long long)d + (unsigned long long)cy + half;         \
                long long x1 = (long long)(y2 & mask) - (long long)half;                                     \
                cy = (long long)((unsigned long long)cr + (unsigned long long)((long long)y2 >> k));         \
                if ((SMALL || SGN) && g.post_neg && ng_) x1 = (long long)(0ull - (unsigned long long)x1);    \
                if (RSH) {                                                                                   \
                    if (writes) PZ_TAIL_RSH_STORE(x1)                                                        \
                } else if (writes) {                                                                         \
                    if (ACC32 && (g.acc32 & 16)) (reinterpret_cast<short*>(g.res) + (res_col - g.res))[(long long)j * res_ls + (long long)(c0 / CB) * (2 * M1 * CB) + (b_ov * CB + b_cv) + (SPLIT ? hs * R2 * CB : 0) + (h * M1 * CB + (SPLIT ? 2 : 1) * R2 * CB * n1)] = (short)x1; \
                    else if (ACC32 && (g.acc32 & 2)) (reinterpret_cast<int*>(g.res) + (res_col - g.res))[(long long)j * res_ls + idx] = (int)x1; \
                    else if (ROWMAJOR) st_stream(res_col + (long long)j * res_ls + idx, x1);                 \
                    else res_col[(long long)j * res_ls + idx] = x1;                                          \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }
        // NZ = 1 (the diagonal launches of a tensoring): the same-base steps of vec_znx_normalize with a bit offset in f64, as the plain carry
        // chain below - every value an exact integer below 2^51, digits of at most 31 bits.  Per coefficient (normalize.rs:50-144,
        // znx/normalization.rs:107-221 with lsh):  d = digit_{k-lsh}(x), cr = carry_{k-lsh}(x);  dpc = d 2^lsh + c;  x1 = digit_k(dpc);
        // c <- cr + carry_k(dpc), with digit_w(y) = y - floor((y + 2^(w-1)) 2^-w) 2^w - the value of the reference's shift pairs - about 12 f64
        // operations where the integer form spends ~45 (64-bit shifts at quarter rate).  Built earlier in round 4 inside the one NZ
        // instantiation, where it spilled (88 -> 140 B of scratch, tails 4.4 -> 5.0 ms); the instantiation without the mode-5 prefetch has the
        // registers for it.  The integer steps stay as the fallback for values beyond 2^51 (the carries are converted once, for good).
        // (Measured and dropped: the per-coefficient tests of the store forms - raw / carry-only / plain / minus the diagonal digits / the
        //  NzCombine modes: ~10 scalar branches in front of every store, 4 000 in the instantiation - hoisted into one choice per limb with
        //  straight-line bodies.  The compiler then keeps every chain of a thread in flight: 290 - 550 B of scratch whatever the
        //  sched_barriers; the branches are what bounds its live ranges here.)
        // (round 6: the pairwise forms that read the 16-bit side copies ride on the same f64 steps - the digit minus the two diagonal digits leaves
        //  through the store; on the integer steps the N = 2^16 pairwise launch ran at 2.2 TB/s of its own bytes once it wrote 16-bit digits only)
        const bool nzf = (NZ == 1 || (D16R && PZ_TAIL_D16R_F64)) && (NZD || g.nz != 0);
        if (nzf && !icarry && big < 2251799813685247.0) {
            if (j >= g.nz_a_end) {
                const int kk_ = k - g.nz_lsh;
                const double halfkk = (double)(1ull << (kk_ - 1)), twokk = 2.0 * halfkk, invkk = 1.0 / twokk, lshmul = (double)(1ull << g.nz_lsh);
#pragma unroll
                for (int n1 = 0; n1 < RE; ++n1) {
                    const int j1 = b_ov + R2 * PZ_TAIL_N1(n1);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const long long idx = (long long)j1 * g.m2 + c0 + b_cv + (h ? m : 0);
                        const double r = round_half_away(h ? v[n1].y : v[n1].x);
                        const double q = floor((r + halfkk) * invkk);
                        const double dd = __builtin_fma(-q, twokk, r);
                        const double dpc = __builtin_fma(dd, lshmul, __longlong_as_double(carry[2 * n1 + h]));
                        const double q2 = floor((dpc + halfd) * invk);
                        carry[2 * n1 + h] = __double_as_longlong(q + q2);
                        if (j < g.nz_a_start) {
                            const long long x1_ = (long long)(int)__builtin_fma(-q2, twok, dpc);
                            if constexpr (D16R) {
                                const int xv_ = (int)x1_ - (d5 ? d5a[NZ == 2 ? 2 * n1 + h : 0] : 0);
                                if (D16ONLY) g.d16w[PZ_TAIL_D16_AT(j - g.nz_a_start + g.nz_res_start, n1, h, idx)] = (short)xv_;
                                else tz_put(res_col + (long long)(j - g.nz_a_start + g.nz_res_start) * res_ls + idx, g.nz_mode, (long long)xv_);
                            } else
                            PZ_TAIL_NZ_STORE(j - g.nz_a_start + g.nz_res_start, idx, x1_, n1, h)
                        }
                    }
                }
            }
        } else
        if (FCARRY && !icarry && !(PZ_TAIL_IS_RAW || PZ_TAIL_IS_NZ) && big < 2251799813685247.0) {
#pragma unroll
            for (int n1 = 0; n1 < RE; ++n1) {
                const int j1 = b_ov + R2 * PZ_TAIL_N1(n1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const long long idx = (long long)j1 * g.m2 + c0 + b_cv + (h ? m : 0);
                    const double val = h ? v[n1].y : v[n1].x;
                    double r = round_half_away(val);
                    if (OP16) { const double op_ = (double)(int)(h ? op16[OP16 ? n1 : 0].y : op16[OP16 ? n1 : 0].x); r += g.small_neg ? -op_ : op_; }   // (an exact integer below 2^51 + 2^15)
                    bool ng_ = false;
                    if (SGN) {   // s(n) in front of the chain, and back on the digit (the integer path's steps, on the f64 chain)
                        ng_ = (((unsigned)idx * g.auto_mul) & (unsigned)(2 * n - 1)) >= (unsigned)n;
                        if (ng_ != (g.auto_neg != 0)) r = -r;
                    }
                    const double vv = r + __longlong_as_double(carry[2 * n1 + h]);
                    const double q = floor((vv + halfd) * invk);
                    carry[2 * n1 + h] = __double_as_longlong(q);
                    if (writes) {
                        long long x1 = (long long)(int)__builtin_fma(-q, twok, vv);
                        if (SGN && g.post_neg && ng_) x1 = -x1;
                        if constexpr (RSH) PZ_TAIL_RSH_STORE(x1)
                        else
                        if (ROWMAJOR) st_stream(res_col + (long long)j * res_ls + idx, x1); else res_col[(long long)j * res_ls + idx] = x1;
                    }
                }
            }
        } else {
            if (FCARRY && !icarry && !(PZ_TAIL_IS_RAW || (PZ_TAIL_IS_NZ && !nzf))) {   // leave the f64 chain: the carries become integers, for good
                icarry = true;
#pragma unroll
                for (int u = 0; u < 2 * RE; ++u) carry[u] = fast_i64_from_integral(__longlong_as_double(carry[u]));
            }
            if (big < 2251799813685247.0) {  // 2^51 - 1 (false for NaN too)
                PZ_TAIL_COEFFS(fast_i64_from_integral)
            } else {
                PZ_TAIL_COEFFS(sat_i64_from_integral)
            }
        }
#undef PZ_TAIL_COEFFS
#undef PZ_TAIL_RSH_STORE
        __syncthreads();
    }
    if (PZ_TAIL_IS_NZ) {   // the top res limbs are digits of the carry alone (middle_step_assign / final_step_assign on zero limbs, normalization.rs:132-157, 254-272)
#pragma unroll
        for (int n1 = 0; n1 < RE; ++n1) {
            const int j1 = b_o + R2 * PZ_TAIL_N1(n1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long long idx = (long long)j1 * g.m2 + c0 + b_c + (h ? m : 0);
                long long c_ = ((NZ == 1 || (D16R && PZ_TAIL_D16R_F64)) && !icarry) ? fast_i64_from_integral(__longlong_as_double(carry[2 * n1 + h])) : carry[2 * n1 + h];
                for (int jj = 0; jj < g.nz_res_end; ++jj) {
                    const long long x1_ = tz_digit(k, c_);
                    PZ_TAIL_NZ_STORE(g.nz_res_end - jj - 1, idx, x1_, n1, h)
                    if (jj != g.nz_res_end - 1) c_ = tz_carry(k, c_, x1_);
                }
            }
        }
    }
#undef PZ_TAIL_NZ_STORE
#undef PZ_TAIL_D16_AT
#undef PZ_TAIL_IS_NZ
#undef PZ_TAIL_IS_RAW
}

#undef PZ_TAIL_N1

}  // namespace pz
