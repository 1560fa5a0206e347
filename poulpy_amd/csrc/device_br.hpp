// device_br.hpp — CGGI blind rotation as ONE kernel per batch
// (poulpy-bin-fhe/src/blind_rotation/algorithms/cggi/algorithm.rs:265-368, execute_block_binary; :370-440, execute_standard).
//
// One workgroup owns CT LWE ciphertexts for the whole rotation: the accumulator GLWEs (i64 limbs) and ONE complex
// work buffer per ciphertext live in LDS from the first block to the last, so per block only the prepared key is read
// (from L2: every workgroup walks the same keys in the same order, and the CT ciphertexts of a workgroup share each
// key value) and nothing is written; HBM sees the LUT, the LWE exponents and the final GLWEs.  The reference's
// per-block op sequence
//     vec_znx_dft_apply x cols | vec_znx_dft_zero | block_size x (vmp_apply_dft_to_dft, svp_apply_dft_to_dft, add, sub)
//     | vec_znx_idft_apply, big_add_small_assign, big_normalize x cols
// becomes: pack+twist -> 3 Stockham passes (R0 x 8 x 8, natural order in and out) -> pointwise product with
// (DFT(X^a) - 1) folded in -> 3 inverse passes -> untwist, round, + acc, carry chain.  Every pass and the product
// work IN PLACE on the one buffer: all threads first read their inputs into registers, a barrier, then they write.
// Sizes: m = N/2 in {128, 256, 512}; rows of the product <= 8, output polynomials <= 8; LDS <= 160 KiB.
#pragma once
#include <type_traits>
#include "device_fft.hpp"

// (Round 5, measured and removed: m = 256 as two radix-16 passes, a wave owning four polynomials - a third fewer LDS stores, 40 % fewer LDS reads, half as many
//  busy waves: 154 100 -> 155 000 rotations/s at 1024 per call, 107 100 -> 103 600 at 256.)

namespace pz {

struct BrFusedArgs {
    long long* res;        // [batch] GLWE(cols, rsz), overwritten
    const long long* lut;  // VecZnx(1, lut_size)
    const long long* lwe;  // [batch][n_lwe + 1] = mod_switch_2n output: [b, a_1 .. a_n]
    const cplx* brk;       // n_lwe prepared GGSWs, P[(r*ncols + c)*m + q]
    const cplx* w2n;       // exp(2 pi i t / 4m), t < 4m
    long long key_stride;  // points between consecutive keys
    int n_lwe, blk, cols, rsz, dnum, bsz, lut_size, base2k, m, batch;
    unsigned long long* margin;   // rounding-margin probe (margin_note, device_fft.hpp); null = off
    int dbg_skip;          // (run-time on purpose: with the tests compiled out — PZ_DBG, device_fft.hpp — this kernel's register allocation changes and the
                           //  N = 1024 two-ciphertext variants spill: 112 000 -> 89 000 rotations/s, round 3)  timing diagnostic (wrong results): 1 no DFT passes, 2 no product, 4 no carry phase, 8 no pack
};

template <bool ACC32> struct AccT { typedef long long type; };
template <> struct AccT<true> { typedef int type; };

// one extra point per 16: radix passes read with stride m/R and write with stride p, both powers of two.
// (Round 5, measured and dropped: position i ^ ((i >> 3) & 7) without padding - by the bank model of tools/dbg/lds_bank_model.py every access of every pass
//  then takes the conflict-free number of LDS cycles, where this layout reads at 2 x and writes the first two passes at 2 x, and a polynomial takes m entries
//  instead of 17 m / 16.  Same rate (N = 512: 170 000 - 171 200 vs 171 200 - 173 000 rotations/s, N = 1024: 146 700 - 148 300 vs 145 800 - 147 600), but the XORed
//  offsets no longer fold into base + constant: 20 forms spill 12 - 148 B.  What did cost time were the twiddle reads, now from per-pass tables.)
__device__ __forceinline__ int br_pad(int i) { return i + (i >> 4); }

// Stockham autosort pass, in place (natural order in and out): sub-transforms of length p are done, this pass makes p*R.
//   u[r] = buf[i + r*m/R] * w^(+-k r), w = exp(2 pi i/(pR)), k = i mod p;  DFT_R;  buf[(i-k)*R + k + s*p] = u[s]
// The factors w^(k r) of a pass come from its own table T[(r - 1) p + k] (round 5): the lanes of a read differ in k only, so it touches consecutive
// 16-byte entries.  From one table exp(2 pi i t / m) indexed r k m/(pR) the reads of even r were 2- and 4-way bank conflicts (a third of the LDS's busy time).
// A thread owns up to JMAX butterflies (job = tid + jj*NT) and keeps them in registers across the barrier.
// 16-byte key value by buffer load: resource in SGPRs, scalar byte offset, 32-bit lane offset
typedef int br_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cplx br_key_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const br_v4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
    cplx d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}

template <int R> struct Log2 { static constexpr int v = 1 + Log2<R / 2>::v; };
template <> struct Log2<1> { static constexpr int v = 0; };

// Polynomials of the pass: njobs_poly of them, np1 per ciphertext, ciphertext slots P polynomials apart (np1 < P: the transform runs over fewer
// polynomials than a slot holds - both ciphertexts of a workgroup still share ONE pass and its two barriers; until round 5 they ran one after the
// other with a quarter of the threads busy at the reference's shape: 4 input rows in slots of 8).
// WOWN (m >= 256): a wave owns the same polynomials in all three passes of a transform - 64 radix-8 jobs are two polynomials at m = 256, one at
// m = 512, and the first pass (64 jobs per polynomial for R0 = 4 and 8 alike) takes its jobs in that order too (PAIR: polynomials 2w, 2w + 1) -
// so inside a transform nothing crosses waves: the LDS serves one wave's reads and writes in issue order, and the passes need no workgroup
// barrier, only the compiler kept from moving LDS accesses across the pass boundary.  The caller puts one barrier behind the last pass.
__device__ __forceinline__ void br_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int R, bool INV, int JMAX, int NT, bool WOWN = false, bool PAIR = false>
__device__ __forceinline__ void br_pass(cplx* buf, int njobs_poly, int np1, int P, int mp, int m, int lm, int p, const cplx* T, int tid) {
    // m, p and R are powers of two: positions come from shifts and masks (lm = log2 m), never from integer division
    const int lt = lm - Log2<R>::v;
    const int t = 1 << lt;
    const int njobs = njobs_poly << lt;
    cplx u[JMAX][R];
#pragma unroll
    for (int jj = 0; jj < JMAX; ++jj) {
        const int job = PAIR ? (((tid >> 6) * 2 + jj) << 6) + (tid & 63) : tid + jj * NT;
        if (job < njobs) {
            const int poly = job >> lt, i = job & (t - 1);
            const int k = i & (p - 1);
            const cplx* src = buf + (poly < np1 ? poly : poly - np1 + P) * mp;
#pragma unroll
            for (int r = 0; r < R; ++r) u[jj][r] = src[br_pad(i + r * t)];
            if (p > 1) {
                // (all R - 1 twiddle reads in front of the products: written read-and-multiply per r, each read sat behind lgkmcnt(0) - round 5 ISA)
                cplx twd[R];
#pragma unroll
                for (int r = 1; r < R; ++r) twd[r] = T[(r - 1) * p + k];   // w^(r k), w = exp(2 pi i / (p R)): the pass's own table, k contiguous
#pragma unroll
                for (int r = 1; r < R; ++r) u[jj][r] = cmul_t<INV>(u[jj][r], twd[r]);
            }
            Bfly<R, INV>::run(u[jj]);
        }
    }
    if (WOWN) br_wave_sync(); else __syncthreads();
#pragma unroll
    for (int jj = 0; jj < JMAX; ++jj) {
        const int job = PAIR ? (((tid >> 6) * 2 + jj) << 6) + (tid & 63) : tid + jj * NT;
        if (job < njobs) {
            const int poly = job >> lt, i = job & (t - 1);
            const int k = i & (p - 1);
            cplx* dst = buf + (poly < np1 ? poly : poly - np1 + P) * mp;
            const int j = (i - k) * R + k;
#pragma unroll
            for (int s = 0; s < R; ++s) dst[br_pad(j + s * p)] = u[jj][s];
        }
    }
    if (WOWN) br_wave_sync(); else __syncthreads();
}

// CT ciphertexts per workgroup, NT threads, output polynomials in groups of CG, PJ product jobs per thread
// (m * ceil(ncols/CG) <= PJ*NT), row_max <= MAXR.
// LDS: T2, T3 (m entries) | X[CT][P][mp] (cplx) | acc[CT][rsz][cols][n].  ACC32 stores the accumulators as 32-bit digits (base2k <= 31):
// after the first block they are normalized digits; the first block reads X^b * LUT (any i64) straight from global memory.
// STD = execute_standard (algorithm.rs:370-440, block size 1): per LWE coefficient  tmp = external_product(acc, BRK_i)  (product
// without the monomial factor, rounding, carry chain WITHOUT adding acc),  acc += (X^a_i - 1) * tmp  on the i64 limbs (a gather in
// LDS), and one in-place normalization of acc at the very end; needs a second accumulator-sized LDS array for tmp.
template <int R0, int CT, int NT, int PJ, int MAXR, int CG, bool ACC32, bool STD = false, bool PROBE = false>
__global__ void __launch_bounds__(NT, 2) k_br_fused(BrFusedArgs g) {
    static_assert(!(STD && ACC32), "the standard variant keeps un-normalized sums: 64-bit accumulators");
    typedef typename AccT<ACC32>::type acc_t;
    // radix-8 butterflies a thread may own per pass (CT*P*m/8 <= JM8*NT, host-checked): two only for m = 512 with CT = 2
    constexpr int JM8 = (CT == 2 && R0 == 8) ? 2 : 1;
    constexpr bool WOWN = R0 != 2;   // br_pass: waves own their polynomials through a transform (m = 128: four polynomials per radix-8 wave, two per first-pass wave)
    extern __shared__ cplx lds_br[];
    const int tid = threadIdx.x;
    const int m = g.m, n = 2 * m, cols = g.cols;
    const int lm = 31 - __builtin_clz((unsigned)m);  // m is a power of two and NT a multiple of it (host-checked)
    const int mp = br_pad(m);
    const int in_limbs = min(g.dnum, g.rsz);  // limbs of acc that enter the product (acc_dft has dnum limbs, the rest are zero)
    const int row_max = cols * in_limbs, ncols = cols * g.bsz;
    const int P = max(row_max, ncols);
    const int ct_polys = g.rsz * cols;        // accumulator polynomials per ciphertext
    cplx* T2 = lds_br;                        // twiddle tables of the second and third pass (m entries reserved, 63 R0 used)
    cplx* T3 = T2 + 7 * R0;
    cplx* X = T2 + m;
    acc_t* acc = reinterpret_cast<acc_t*>(X + CT * P * mp);
    acc_t* tmpd = acc + (STD ? CT * g.rsz * cols * n : 0);  // STD: digits of the current external product
    const int b0 = blockIdx.x * CT;
    const unsigned mask2 = 2u * (unsigned)n - 1u;

    // T2[(r - 1) R0 + k] = exp(2 pi i r k / (8 R0)), k < R0;  T3[(r - 1) 8 R0 + k] = exp(2 pi i r k / m), k < 8 R0  (7 R0 + 56 R0 <= m entries)
    for (int e = tid; e < 63 * R0; e += NT) {
        const bool second = e >= 7 * R0;
        const int ee = second ? e - 7 * R0 : e, pp = second ? 8 * R0 : R0;
        const int r = ee / pp + 1, kk = ee % pp;
        T2[e] = g.w2n[4 * (r * kk * (second ? 1 : 8))];   // w2n[4 t] = exp(2 pi i t / m); m = 64 R0
    }
    // NT is a multiple of m (host-checked): a thread's pack / carry jobs all have j = tid mod m, so its twist factor
    // exp(2 pi i j / 4m) is fetched once
    const cplx tw_j = g.w2n[tid % m];
    // X^b * LUT in column 0, zero elsewhere (:298-301): coefficient idx of (ct, limb, col)
    auto lut_rot = [&](int ct, int limb, int col, int idx) -> long long {
        if (col != 0 || limb >= g.lut_size) return 0;
        const int b = min(b0 + ct, g.batch - 1);
        const unsigned kk = (unsigned)((unsigned long long)g.lwe[(long long)b * (g.n_lwe + 1)] & (unsigned long long)mask2);
        const unsigned i0 = ((unsigned)idx - kk) & mask2;
        const long long s = g.lut[(long long)limb * n + (i0 & (unsigned)(n - 1))];
        return i0 >= (unsigned)n ? (long long)(0ull - (unsigned long long)s) : s;
    };
    if (!ACC32) {
        for (int e = tid; e < CT * ct_polys * n; e += NT) {
            const int j = e % n, pc = (e / n) % ct_polys, ct = e / (n * ct_polys);
            acc[e] = (acc_t)lut_rot(ct, pc / cols, pc % cols, j);
        }
    }
    __syncthreads();

    const int k = g.base2k;
    const unsigned long long half = 1ull << (k - 1), dmask = (1ull << k) - 1;
    const double inv_m = 1.0 / (double)m;
    const int ncg = (ncols + CG - 1) / CG;    // output polynomials are handled CG at a time
    const int njobs_prod = m * ncg;           // product jobs (q, column group); a job covers the CT ciphertexts

    // Monomial factors.  DFT(X^a_i)[q] needs a_i (one word of the LWE sample) and then an entry of the root table at a_i (4q + 1): two dependent
    // requests, which until round 5 were made per coefficient inside the product loop - two memory latencies in front of every coefficient's last
    // FMAs.  Now: lane l of a wave holds a_(blk0 + l) of the block (blk <= 64, host-checked), requested one block ahead; the factor of coefficient
    // i + 1 is requested while coefficient i is multiplied (a block's first one together with its key values: it is used behind their FMAs).
    const int lane = tid & 63;
    const unsigned qf = 4u * (unsigned)(tid & (m - 1)) + 1u;   // (NT is a multiple of m: a thread's product jobs share q)
    auto load_amounts = [&](unsigned (&dst)[CT], int first) {
        const int i = min(first + min(lane, g.blk - 1), g.n_lwe - 1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int b = min(b0 + ct, g.batch - 1);
            dst[ct] = (unsigned)((unsigned long long)g.lwe[(long long)b * (g.n_lwe + 1) + 1 + i] & (unsigned long long)mask2);
        }
    };
    auto load_factors = [&](cplx (&dst)[CT], const unsigned (&amounts)[CT], int l) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dst[ct] = g.w2n[((unsigned)__builtin_amdgcn_readlane((int)amounts[ct], l) * qf) & mask2];
    };
    // XPRE (m = 512): the factors of a block's first coefficient are requested at the end of the previous block's product phase (N = 1024: + 2.4 %; at
    // m = 256 their eight registers, live across every phase, cost more than the request in flight with the first key values: - 2.5 %)
    constexpr bool XPRE = R0 == 8;
    unsigned aiv[CT], aivn[CT];
    cplx xmn[XPRE ? CT : 1];
    load_amounts(aiv, 0);
    if (XPRE && !STD) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) xmn[XPRE ? ct : 0] = g.w2n[((unsigned)__builtin_amdgcn_readlane((int)aiv[ct], 0) * qf) & mask2];
    }
    // (Measured and dropped: the key values of a block's first coefficient requested at the end of the previous block's product phase, so that they
    //  arrive during the transforms and the carry chains - 64 registers live across every phase, affordable only with the thread-index arithmetic
    //  recomputed per phase: N = 512 154 600 -> 148 800 - 153 600 rotations/s.)

    for (int blk0 = 0; blk0 + g.blk <= g.n_lwe; blk0 += g.blk) {
        // Everything the phases derive from the thread index (butterfly positions, LDS offsets, key offsets) is loop-invariant;
        // hoisted out of this loop it would stay live across all phases and spill.  An opaque copy of the index per
        // iteration makes the compiler recompute those few integer operations where they are used.
        // Only where it would spill (two ciphertexts at m = 512): elsewhere keeping them in registers is faster (9-13 %).
        int tidv = tid;
        if (CT == 2 && R0 == 8) asm volatile("" : "+v"(tidv));
        const bool from_lut = ACC32 && blk0 == 0;  // the accumulator is still X^b * LUT in global memory
        // ---- pack + twist: X[ct][r][j] = (acc[r][j] + i acc[r][j+m]) * exp(2 pi i j / 4m),  r = limb*cols + col (:319-320)
        if (!(g.dbg_skip & 8))
        for (int pr = tidv >> lm; pr < CT * row_max; pr += NT >> lm) {  // (ciphertext, row) pairs; j = tid mod m is fixed
            const int j = tidv & (m - 1), ct = (CT == 2 && pr >= row_max) ? 1 : 0, r = pr - ct * row_max;
            const acc_t* a = acc + ((long long)ct * ct_polys + r) * n;
            const cplx z = from_lut ? make_double2((double)lut_rot(ct, r / cols, r % cols, j), (double)lut_rot(ct, r / cols, r % cols, j + m))
                                    : make_double2((double)a[j], (double)a[j + m]);
            X[(ct * P + r) * mp + br_pad(j)] = cmul(z, tw_j);
        }
        __syncthreads();
        // forward DFT of the CT*row_max input polynomials (they sit at poly index ct*P + r)
        if (!(g.dbg_skip & 1))
        {
            br_pass<R0, false, 2, NT, WOWN, WOWN && R0 == 4>(X, CT * row_max, row_max, P, mp, m, lm, 1, T2, tidv);
            br_pass<8, false, JM8, NT, WOWN>(X, CT * row_max, row_max, P, mp, m, lm, R0, T2, tidv);
            br_pass<8, false, JM8, NT, WOWN>(X, CT * row_max, row_max, P, mp, m, lm, R0 * 8, T3, tidv);
            if (WOWN) __syncthreads();
        }
        // ---- product, in place: X[ct][c][q] = sum_i (DFT(X^a_i)[q] - 1) * sum_r X[ct][r][q] * BRK_i[r][c][q]   (:321-337)
        if (!(g.dbg_skip & 2)) {
            load_amounts(aivn, blk0 + g.blk);   // the next block's rotation amounts
            cplx out[PJ][CT][CG];
            // EARLY (two jobs per thread, rows <= CG): the second job's column group starts at output polynomial CG >= row_max, which no thread
            // reads as an input - it runs first and stores at once, so only one job's sums are live across the barrier
            constexpr bool EARLY = PJ == 2 && MAXR <= CG;
#pragma unroll
            for (int pjr = 0; pjr < PJ; ++pjr) {
                const int pj = EARLY ? PJ - 1 - pjr : pjr;
                const int job = tidv + pj * NT;
                if (job < njobs_prod) {
                    // m is a multiple of 64, so a wave has one column group: keep it (and every key row pointer) in SGPRs,
                    // the loads then need one VGPR offset instead of a 64-bit VGPR pointer each
                    const int q = job & (m - 1), cg = __builtin_amdgcn_readfirstlane(job >> lm);
                    // input points of this frequency: in registers for the whole block, or (ALDS: two ciphertexts with more than
                    // 4 rows, where registers would spill) re-read from LDS at each use
                    constexpr bool ALDS = CT == 2 && MAXR > 4;
                    cplx a[ALDS ? 1 : CT][ALDS ? 1 : MAXR];
                    if (!ALDS) {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                            for (int r = 0; r < MAXR; ++r)
                                a[ALDS ? 0 : ct][ALDS ? 0 : r] = r < row_max ? X[(ct * P + r) * mp + br_pad(q)] : make_double2(0.0, 0.0);
                    }
#define PZ_BR_A(CT_, R_) (ALDS ? X[((CT_) * P + (R_)) * mp + br_pad(q)] : a[ALDS ? 0 : (CT_)][ALDS ? 0 : (R_)])
                    // coefficient by coefficient: the CG x row_max key values of this thread's column group are requested
                    // together (one exposed L2 latency per coefficient), then consumed
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int j = 0; j < CG; ++j) out[pj][ct][j] = make_double2(0.0, 0.0);
                    // (a second register set that prefetches coefficient i+1 during the arithmetic of i was tried: it spills
                    // at 256 VGPRs and is 5-12 % slower.  Round 3: the coefficient loop as one stream of (coefficient, column) steps with
                    // a ring of CG key slots requested CG - 1 steps ahead and the monomial factors one coefficient ahead - the
                    // scheme that pays in k_mid128r - was bit-exact and SLOWER here, 129 500 -> 122 800 rotations/s at N = 512 and
                    // 109 900 -> 107 500 at N = 1024 (52-76 bytes of scratch; profiles/r03_ab_br_pipe.txt): the latency of the
                    // coefficient's requests is already covered by the other waves)
                    cplx xn[CT];
                    if (!STD) {
                        if constexpr (XPRE) {
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) xn[ct] = xmn[XPRE ? ct : 0];
                        } else {
                            load_factors(xn, aiv, 0);   // (in flight with the first coefficient's key values; used behind its FMAs)
                        }
                    }
                    for (int i = blk0; i < blk0 + g.blk; ++i) {
                        const cplx* K = g.brk + (long long)i * g.key_stride;
                        const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, -1, 0x00027000);   // raw buffer, no bounds
                        cplx kv[CG][MAXR];
#pragma unroll
                        for (int j = 0; j < CG; ++j) {
                            const int c = min(cg * CG + j, ncols - 1);
#pragma unroll
                            for (int r = 0; r < MAXR; ++r)
                                // (buffer load: the coefficient's key as the resource, the row's byte offset in an SGPR, ONE 32-bit lane offset shared by all
                                //  CG x MAXR loads - as global loads each had its own 64-bit vector add in front; round 5 ISA)
                                kv[j][r] = (PZ_DBG(g.dbg_skip) & 16) ? make_double2(1.0, (double)(q + r))
                                         : br_key_load(krs, (unsigned)q * 16u, (unsigned)((min(r, row_max - 1) * ncols + c) * m) * 16u);
                        }
                        cplx xm[CT];
                        if (!STD) {
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) {
                                xm[ct] = xn[ct];
                                xm[ct].x -= 1.0;   // d = DFT(X^a_i)[q] - 1: the factor enters as four FMAs (round 5)
                            }
                            load_factors(xn, aiv, min(i + 1 - blk0, g.blk - 1));   // the next coefficient's (the last request of a block is unused)
                        }
                        // (Round 5, not measured: the key values in two halves of two columns, each half requested for the next coefficient right behind its
                        //  FMAs - the same 64 registers on paper, 212 B of scratch in the reference shape's form, 60 B with one ciphertext: as in round 3.)
                        // (Round 5, measured and dropped: the sums of 2 columns x 2 ciphertexts side by side - 8 FMA chains interleaved instead of the 2 the
                        //  compiler leaves from this order: no gain on any form, - 1 % at N = 1024; the other wave of the SIMD already fills the gaps.)
                        // GUARD_: row_max < MAXR (the template's row count is the next of 4 / 6 / 8).  With row_max == MAXR - the usual shapes -
                        // the per-row test is dropped: as a run-time test inside the unrolled row loop it costs a branch per (column, ciphertext,
                        // row), 64 branches and 167 scalar instructions per coefficient beside 188 floating-point ones (round 3 ISA)
#define PZ_BR_FMAS(GUARD_)                                                                                        \
    _Pragma("unroll") for (int j = 0; j < CG; ++j) {                                                             \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                      \
            cplx s = make_double2(0.0, 0.0);                                                                     \
            _Pragma("unroll") for (int r = 0; r < MAXR; ++r) {                                                   \
                if (!(GUARD_) || r < row_max) {                                                                  \
                    const cplx av = PZ_BR_A(ct, r);                                                              \
                    s.x = __builtin_fma(av.x, kv[j][r].x, s.x);                                                  \
                    s.x = __builtin_fma(-av.y, kv[j][r].y, s.x);                                                 \
                    s.y = __builtin_fma(av.x, kv[j][r].y, s.y);                                                  \
                    s.y = __builtin_fma(av.y, kv[j][r].x, s.y);                                                  \
                }                                                                                                \
            }                                                                                                    \
            if (STD) {                                                                                           \
                out[pj][ct][j].x += s.x;                                                                         \
                out[pj][ct][j].y += s.y;                                                                         \
            } else {                                                                                             \
                out[pj][ct][j].x = __builtin_fma(xm[ct].x, s.x, out[pj][ct][j].x);                               \
                out[pj][ct][j].x = __builtin_fma(-xm[ct].y, s.y, out[pj][ct][j].x);                              \
                out[pj][ct][j].y = __builtin_fma(xm[ct].x, s.y, out[pj][ct][j].y);                               \
                out[pj][ct][j].y = __builtin_fma(xm[ct].y, s.x, out[pj][ct][j].y);                               \
            }                                                                                                    \
        }                                                                                                        \
    }
                        // (both forms only where the tile re-reads its operands from LDS - two ciphertexts, more than 4 rows: there the test-free form
                        //  is worth +15 %; in the register-resident variants the second copy of the block spilled 164 bytes for no gain)
                        // ALDS, every row in use (the usual shapes): the operands of row r + 1 are read from LDS in front of row r's FMAs.  Written as in
                        // PZ_BR_FMAS the compiler put each read directly in front of its four FMAs behind lgkmcnt(0) - 30 of a coefficient's 36 LDS reads
                        // fully exposed (round 5 ISA).  Same chains, same order inside each chain.
#define PZ_BR_FMAS_LDS                                                                                            \
    {                                                                                                            \
        cplx cur[CT], nxt[CT];                                                                                   \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) cur[ct] = PZ_BR_A(ct, 0);                              \
        _Pragma("unroll") for (int j = 0; j < CG; ++j) {                                                         \
            cplx s[CT];                                                                                          \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) s[ct] = make_double2(0.0, 0.0);                    \
            _Pragma("unroll") for (int r = 0; r < MAXR; ++r) {                                                   \
                if (!(j == CG - 1 && r == MAXR - 1)) {                                                           \
                    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) nxt[ct] = PZ_BR_A(ct, (r + 1) % MAXR);     \
                }                                                                                                \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                              \
                    s[ct].x = __builtin_fma(cur[ct].x, kv[j][r].x, s[ct].x);                                     \
                    s[ct].x = __builtin_fma(-cur[ct].y, kv[j][r].y, s[ct].x);                                    \
                    s[ct].y = __builtin_fma(cur[ct].x, kv[j][r].y, s[ct].y);                                     \
                    s[ct].y = __builtin_fma(cur[ct].y, kv[j][r].x, s[ct].y);                                     \
                }                                                                                                \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) cur[ct] = nxt[ct];                             \
            }                                                                                                    \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                                  \
                out[pj][ct][j].x = __builtin_fma(xm[ct].x, s[ct].x, out[pj][ct][j].x);                           \
                out[pj][ct][j].x = __builtin_fma(-xm[ct].y, s[ct].y, out[pj][ct][j].x);                          \
                out[pj][ct][j].y = __builtin_fma(xm[ct].x, s[ct].y, out[pj][ct][j].y);                           \
                out[pj][ct][j].y = __builtin_fma(xm[ct].y, s[ct].x, out[pj][ct][j].y);                           \
            }                                                                                                    \
        }                                                                                                        \
    }
                        // (m = 256: the compiler keeps the re-read operands of the pipelined form in registers and spills 100 - 170 B)
                        if constexpr (ALDS && R0 == 8) { if (row_max == MAXR) { PZ_BR_FMAS_LDS } else { PZ_BR_FMAS(true) } }
                        else if constexpr (ALDS) { if (row_max == MAXR) { PZ_BR_FMAS(false) } else { PZ_BR_FMAS(true) } }
#undef PZ_BR_FMAS_LDS
                        else { PZ_BR_FMAS(false) }   // rows beyond row_max hold zeros (a) and a clamped key row: their FMAs add exact zeros
#undef PZ_BR_FMAS
                    }
                    if (EARLY && pj == 1) {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                            for (int j = 0; j < CG; ++j) {
                                const int c = cg * CG + j;
                                if (c < ncols) X[(ct * P + c) * mp + br_pad(q)] = out[pj][ct][j];
                            }
                    }
                }
            }
#undef PZ_BR_A
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) aiv[ct] = aivn[ct];
            if (XPRE && !STD) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xmn[XPRE ? ct : 0] = g.w2n[((unsigned)__builtin_amdgcn_readlane((int)aiv[ct], 0) * qf) & mask2];
            }
            __syncthreads();  // every input point has been read: the outputs may overwrite them
#pragma unroll
            for (int pj = 0; pj < (EARLY ? 1 : PJ); ++pj) {
                // (an opaque copy of the index: the store offsets below, hoisted out of the block loop, were what the register-bound forms spilled -
                //  reloaded here one by one, each behind vmcnt(0); round 5 ISA)
                int tst = tidv;
                if (!(CT == 2 && R0 == 8)) asm volatile("" : "+v"(tst));   // (there tidv is opaque already)
                const int job = tst + pj * NT;
                if (job < njobs_prod) {
                    const int q = job & (m - 1), cg = job >> lm;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int j = 0; j < CG; ++j) {
                            const int c = cg * CG + j;
                            if (c < ncols) X[(ct * P + c) * mp + br_pad(q)] = out[pj][ct][j];
                        }
                }
            }
            __syncthreads();
        }
        // inverse DFT of the CT*ncols output polynomials
        if (!(g.dbg_skip & 1))
        {
            br_pass<R0, true, 2, NT, WOWN, WOWN && R0 == 4>(X, CT * ncols, ncols, P, mp, m, lm, 1, T2, tidv);
            br_pass<8, true, JM8, NT, WOWN>(X, CT * ncols, ncols, P, mp, m, lm, R0, T2, tidv);
            br_pass<8, true, JM8, NT, WOWN>(X, CT * ncols, ncols, P, mp, m, lm, R0 * 8, T3, tidv);
            if (WOWN) __syncthreads();
        }
        // ---- untwist, round(x/m), + acc, carry chain from the last limb to limb 0 (:342-346); thread = (ct, column, j < m):
        //      coefficients j and j+m.  Same digit/carry arithmetic as the fused tail (device_fft.hpp, PZ_TAIL_COEFFS).
        // The (ciphertext, column) pairs of a thread run their chains side by side, CU at a time (4 when the thread has more than two pairs): a chain
        // is ~30 dependent integer operations per limb and coefficient, and the two coefficients of one pair alone leave the SIMD waiting on its
        // own results (N = 512: 147 900 -> 151 500 rotations/s with two, 153 400 with four; four where a thread has only two pairs: - 11 %).
        auto carry_phase = [&](auto cu_tag) {
        constexpr int CU = decltype(cu_tag)::value;
        for (int pc0 = tidv >> lm; pc0 < CT * cols; pc0 += CU * (NT >> lm)) {  // (ciphertext, column) pairs; j = tid mod m is fixed
            const int j = tidv & (m - 1);
            const cplx tw = tw_j;
            int ctv[CU], colv[CU];
            bool on[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int pc = pc0 + u * (NT >> lm);
                on[u] = pc < CT * cols;
                const int pcc = on[u] ? pc : pc0;
                ctv[u] = (CT == 2 && pcc >= cols) ? 1 : 0;
                colv[u] = pcc - ctv[u] * cols;
            }
            // (Round 5: the chain in f64 as in the fused tail - v = r + acc + carry, q = floor((v + 2^(k-1)) 2^-k), digit = v - q 2^k, 13 instructions
            //  where this integer form spends about 30 - was bit-exact and SLOWER: 137 600 -> 128 600 rotations/s at N = 512, no change at N = 1024.
            //  Its floor and the conversions are quarter-rate FP64 instructions; the 32-bit integer ones here are full rate.)
            long long cy[CU][2];
#pragma unroll
            for (int u = 0; u < CU; ++u) cy[u][0] = cy[u][1] = 0;
            // Per limb, in stages over all 2 CU coefficients of the thread: the LDS reads (values, accumulator digits) together, the conversions
            // behind ONE test (every value below 2^51: the three-instruction form), the chain steps as straight-line code behind the limb's uniform
            // tests.  (Until round 5 each coefficient walked through its own tests: ~10 scalar / exec branches per coefficient, every LDS read
            // directly in front of its use behind lgkmcnt(0).)
            for (int limb = g.bsz - 1; limb >= 0; --limb) {
                const bool writes = limb < g.rsz;
                const bool first = limb == g.bsz - 1;
                cplx v[CU];
#pragma unroll
                for (int u = 0; u < CU; ++u) v[u] = X[(ctv[u] * P + limb * cols + colv[u]) * mp + br_pad(j)];
                acc_t* ap[CU];
#pragma unroll
                for (int u = 0; u < CU; ++u) ap[u] = (STD ? tmpd : acc) + ((long long)ctv[u] * ct_polys + (long long)limb * cols + colv[u]) * n + j;
                long long prev[CU][2];
                if (writes && !STD) {
                    if (from_lut) {
#pragma unroll
                        for (int u = 0; u < CU; ++u)
#pragma unroll
                            for (int h = 0; h < 2; ++h) prev[u][h] = lut_rot(ctv[u], limb, colv[u], j + h * m);
                    } else {
#pragma unroll
                        for (int u = 0; u < CU; ++u)
#pragma unroll
                            for (int h = 0; h < 2; ++h) prev[u][h] = (long long)ap[u][h * m];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < CU; ++u) prev[u][0] = prev[u][1] = 0;
                }
                double rv[CU][2];
                double big = 0.0;
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    const cplx w = cmulc(v[u], tw);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const double val = (h ? w.y : w.x) * inv_m;
                        rv[u][h] = round_half_away(val);
                        if (PROBE) margin_note(g.margin, fabs(val - rv[u][h]));   // rounding-margin instantiation (br_forms.hpp)
                        big = fmax(big, fabs(rv[u][h]));   // (fmax drops a NaN: the test below sends it to the saturating form, which maps it to 0)
                        if (rv[u][h] != rv[u][h]) big = 1.0e300;
                    }
                }
                long long x[CU][2];
                // 3-instruction conversion when exact (|x| < 2^51), the saturating one otherwise (Rust `as i64`)
                if (big < 2251799813685247.0) {
#pragma unroll
                    for (int u = 0; u < CU; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h) x[u][h] = fast_i64_from_integral(rv[u][h]);
                } else {
#pragma unroll
                    for (int u = 0; u < CU; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            x[u][h] = fabs(rv[u][h]) < 2251799813685247.0 ? fast_i64_from_integral(rv[u][h]) : sat_i64_from_integral(rv[u][h]);
                }
                long long dg[CU][2], cr[CU][2];
#pragma unroll
                for (int u = 0; u < CU; ++u)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned long long y = (unsigned long long)x[u][h] + (unsigned long long)prev[u][h] + half;
                        dg[u][h] = (long long)(y & dmask) - (long long)half;
                        cr[u][h] = (long long)y >> k;
                    }
                if (first && !writes) {
#pragma unroll
                    for (int u = 0; u < CU; ++u) { cy[u][0] = cr[u][0]; cy[u][1] = cr[u][1]; }
                } else {
                    long long x1[CU][2];
#pragma unroll
                    for (int u = 0; u < CU; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned long long y2 = (unsigned long long)dg[u][h] + (unsigned long long)cy[u][h] + half;
                            x1[u][h] = (long long)(y2 & dmask) - (long long)half;
                            cy[u][h] = (long long)((unsigned long long)cr[u][h] + (unsigned long long)((long long)y2 >> k));
                        }
                    if (writes) {
#pragma unroll
                        for (int u = 0; u < CU; ++u)
                            if (on[u]) { ap[u][0] = (acc_t)x1[u][0]; ap[u][m] = (acc_t)x1[u][1]; }
                    }
                }
            }
            // limbs of acc beyond the precision of the big value are zero (normalize.rs:118-120)
#pragma unroll
            for (int u = 0; u < CU; ++u)
                for (int limb = g.bsz; limb < g.rsz; ++limb) {
                    acc_t* a = (STD ? tmpd : acc) + ((long long)ctv[u] * ct_polys + (long long)limb * cols + colv[u]) * n;
                    if (on[u]) { a[j] = 0; a[j + m] = 0; }
                }
        }
        };
        if (!(g.dbg_skip & 4)) {
            if (CT * cols > 2 * (NT >> lm)) carry_phase(std::integral_constant<int, 4>());
            else carry_phase(std::integral_constant<int, 2>());
        }
        __syncthreads();
        if (STD) {
            // acc += (X^a - 1) * tmp, limb by limb (glwe_mul_xp_minus_one_assign + glwe_add_assign, :426-432)
            for (int e = tidv; e < CT * ct_polys * n; e += NT) {
                const int jj = e & (n - 1), ct = e >= ct_polys * n ? 1 : 0;
                const int b = min(b0 + ct, g.batch - 1);
                const unsigned kk = (unsigned)((unsigned long long)g.lwe[(long long)b * (g.n_lwe + 1) + 1 + blk0] & (unsigned long long)mask2);
                const unsigned i0 = ((unsigned)jj - kk) & mask2;
                const long long src = (long long)tmpd[e - jj + (int)(i0 & (unsigned)(n - 1))];
                const long long rot = i0 >= (unsigned)n ? -src : src;
                acc[e] = (acc_t)((long long)acc[e] + rot - (long long)tmpd[e]);
            }
            __syncthreads();
        }
    }
    if (STD) {
        // one in-place normalization of acc at the end (vec_znx_normalize_assign, :437)
        for (int pc = tid >> lm; pc < CT * cols; pc += NT >> lm) {
            const int j = tid & (m - 1), ct = (CT == 2 && pc >= cols) ? 1 : 0, col = pc - ct * cols;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                long long cy = 0;
                for (int limb = g.rsz - 1; limb >= 0; --limb) {
                    acc_t* a = acc + ((long long)ct * ct_polys + (long long)limb * cols + col) * n;
                    const unsigned long long y = (unsigned long long)(long long)a[j + h * m] + half;
                    const long long d = (long long)(y & dmask) - (long long)half;
                    const long long cr = (long long)y >> k;
                    const unsigned long long y2 = (unsigned long long)d + (unsigned long long)cy + half;
                    a[j + h * m] = (acc_t)((long long)(y2 & dmask) - (long long)half);
                    cy = (long long)((unsigned long long)cr + (unsigned long long)((long long)y2 >> k));
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < CT * ct_polys * n; e += NT) {
        const int ct = e / (ct_polys * n);
        if (b0 + ct < g.batch) {
            // (no complete block at all: the result is the rotated LUT itself)
            const bool untouched = ACC32 && g.blk > g.n_lwe;
            const int pc = (e / n) % ct_polys;
            g.res[(long long)b0 * ct_polys * n + e] = untouched ? lut_rot(ct, pc / cols, pc % cols, e % n) : (long long)acc[e];
        }
    }
}

}  // namespace pz
