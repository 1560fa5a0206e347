// device_small.hpp — two-kernel GLWE product pipeline for N = 1024 / 2048 / 4096 (m = M1 x 128, M1 = 4 / 8 / 16), where a whole
// polynomial (<= 32 KiB of spectrum) fits in LDS.  The three-kernel pipeline moves 1.5 MiB per ciphertext at BASELINE configs[1] (N = 4096, 4 limbs: limbs in,
// T', T' again, T2', T2' again, limbs out); here the spectra cross HBM once:
//   k_small_fwd : i64 limbs -> full forward transform in LDS (column pass in registers, row pass as in the middle kernel) -> S
//   k_small_inv : one workgroup per (ciphertext, output column): pointwise product with the row-sliced key in registers (no LDS:
//                 the product needs no exchange), two output limbs at a time -> inverse row pass + inverse column pass in LDS ->
//                 round, (+ key-switch body), carry chain with the carries in registers -> balanced digits
// = 1.0 MiB per ciphertext (+ the second column's re-read of S, served by the XCD's L2: both columns of a ciphertext are placed on
// one XCD).  Same tables, same stage formulas as k_fwd_pass1 / k_mid128 / k_inv_tail, regrouped; dsize = 1, one base2k, <= 4 key limbs.
//   S  : S[poly][q1][q2]     (the tile layout of the middle kernel after its forward row pass)
//   Pp : P'[q1][r][c][q2]    (the row-sliced key of the three-kernel pipeline)
// Variants of k_small_inv (template flags): NOPROD - the spectra are given (blind rotation: the block step produced them), FWD - the
// forward transform of the new digits follows in the same workgroup (the next block's input), AU - the glwe_automorphism family
// (phi as an index / sign map in the carry-chain stage; optionally the trace's one-bit shift on the way out).
#pragma once
#include "device_fft.hpp"

namespace pz {

constexpr int kSmallM2 = 128, kSmallRS = 16 * 9;   // row stride as in k_mid128 (z[k1][o] at k1*9 + o)
constexpr int kSmallIdftRS = kSmallRS + 1;
// k_small_inv<.., NOPROD> (blind rotation's tail): the spectra arrive in the standard order and consecutive lanes drop them at M1 consecutive ROWS
// of the tile - with a row stride of 144 points (= 0 mod 64 banks) an M1-way bank conflict on every write (round 5 counters: 4.6 conflict cycles per
// LDS instruction, profiles/r05_br_pmc/n2048_pmc.txt).  Row stride = 144 + d with (row * d + column) mod 16 distinct over the 16 lanes of a
// 128-bit access: d = 4 (M1 = 4: 4 rows x 4 columns), 2 (8 rows x 2 columns), 1 (16 rows).  The in-row layout (k1 * 9 + o) is untouched.
constexpr int small_inv_rs(int m1, bool noprod) { return noprod ? kSmallRS + (m1 == 16 ? 1 : (m1 == 8 ? 2 : 4)) : kSmallRS; }
#ifndef PZ_SMALL_STAMP
#define PZ_SMALL_STAMP 0   // diagnostic build: per-phase s_memtime totals of k_small_inv, printed by three waves of three workgroups
#endif

#ifndef PZ_SMALL_PROBE
#define PZ_SMALL_PROBE 1   // (A/B: -DPZ_SMALL_PROBE=0 compiles the run-time rounding-margin probe out of the small-ring kernels)
#endif
struct SmallFwdArgs {
    const long long* src;
    PolyMap smap;
    cplx* S;
    int npolys;
    const cplx* tw1;     // [16] twist of the column pass
    const cplx* tw12t;   // [q1][j2]
    const cplx* wL2;     // exp(2 pi i t / 128)
    int natural;         // 1: spectrum written in the standard device order [q1 + M1 q2] (pointwise consumers holding standard keys: the
                         // blind rotation's block step) instead of S[q1][q2]; the 8 M1 threads of a store still cover 8 M1 consecutive points
    // per-op vec_znx_dft_apply / svp_apply_dft in ONE kernel (round 3; natural order only): polynomial p goes to S + map_off(dmap, p) / 2
    // instead of S + p m, multiplied pointwise by mul (an SvpPPol, standard order) when given
    int use_dmap;
    PolyMap dmap;
    const cplx* mul;
};

// 256 threads = 2 polynomials x 128 threads; LDS 2 x M1 rows x 144 points + wL2
template <int M1>
__global__ void __launch_bounds__(256, 2) k_small_fwd(SmallFwdArgs g) {
    constexpr int M2 = kSmallM2, RS = kSmallRS;
    constexpr long long m = (long long)M1 * kSmallM2;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x, pl = tid >> 7, t = tid & 127;
    cplx* wl = lds + 2 * M1 * RS;
    if (tid < M2) wl[tid] = g.wL2[tid];
    const int p = blockIdx.x * 2 + pl;
    const bool active = p < g.npolys;
    const long long* a = g.src + map_off(g.smap, active ? p : g.npolys - 1);
    cplx* buf = lds + pl * M1 * RS;
    // ---- column pass: thread t owns column j2 = t, M1 points over j1 (k_fwd_pass1 with one radix-M1 butterfly)
    {
        long long re[M1], im[M1];
#pragma unroll
        for (int j1 = 0; j1 < M1; ++j1) {
            re[j1] = ld_stream(a + j1 * M2 + t);
            im[j1] = ld_stream(a + m + j1 * M2 + t);
        }
        cplx v[M1];
#pragma unroll
        for (int j1 = 0; j1 < M1; ++j1) v[j1] = cmul(make_double2((double)re[j1], (double)im[j1]), g.tw1[j1]);
        Bfly<M1, false>::run(v);
#pragma unroll
        for (int q1 = 0; q1 < M1; ++q1) buf[q1 * RS + t] = cmul(v[q1], g.tw12t[q1 * M2 + t]);
    }
    __syncthreads();
    // ---- row pass: 8 lanes per row, exactly the forward pass of k_mid128 (8 * M1 of the polynomial's 128 threads)
    const int row = t >> 3, o = t & 7;
    if (row < M1) {
        cplx* rowbuf = buf + row * RS;
        cplx x[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) x[n1] = rowbuf[o + 8 * n1];
        row_sync();
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
        if (active) {
            cplx* dst = g.S + (g.use_dmap ? map_off(g.dmap, p) / 2 : (long long)p * m) + (g.natural ? row + M1 * o : row * M2 + o);
            const int qs = g.natural ? M1 : 1;
            if (g.mul) {   // (natural order) x ppol[q], q = row + M1 (o + 8h + 16 k2)
                const cplx* mp = g.mul + row + M1 * o;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) x[8 * h + k2] = cmul(x[8 * h + k2], mp[(8 * h + 16 * k2) * M1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) dst[(8 * h + 16 * k2) * qs] = x[8 * h + k2];   // S[q1][q2 = o + 8h + 16 k2]: re-read twice, kept cacheable
        }
    }
}

// standard device VmpPMat  P[p][q1 + M1*q2]  ->  P'[q1][p][q2]  for the ring degrees without a pipeline plan (one thread per point)
template <int M1>
__global__ void __launch_bounds__(256) k_small_permute(const cplx* __restrict__ P, cplx* __restrict__ Pp, int npolys) {
    constexpr int M2 = kSmallM2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over [q1][p][q2]
    const long long total = (long long)M1 * npolys * M2;
    if (idx >= total) return;
    const int q2 = (int)(idx % M2);
    const long long t = idx / M2;
    const int p = (int)(t % npolys), q1 = (int)(t / npolys);
    Pp[idx] = P[(long long)p * (M1 * M2) + q1 + M1 * q2];
}

// Per-op vec_znx_idft_apply in one kernel for N = 1024 / 2048 / 4096 (round 3): spectrum in the standard device order -> inverse row pass
// (k_mid128's) x conj tw12 -> inverse column pass, x tw1inv (1/m folded in), round half away, saturating i64 -> VecZnxBig.  The per-op
// path otherwise runs k_inv_pass2 + k_inv_pass1 with the spectrum's worth of T between them in HBM (32 B per coefficient instead of 16).
struct SmallIdftArgs {
    const cplx* S;       // spectra, polynomial p at S + map_off(smap, p) / 2 (standard order [q1 + M1 q2])
    PolyMap smap;
    long long* res;      // i64 coefficients, polynomial p at res + map_off(dmap, p)
    PolyMap dmap;
    int npolys;
    const cplx* tw12t;
    const cplx* wL2;
    const cplx* tw1inv;
    unsigned long long* margin;   // rounding-margin probe (margin_note, device_fft.hpp); null = off
};
// 256 threads = 2 polynomials x 128 threads; LDS 2 x M1 rows x 144 points + wL2
template <int M1>
__global__ void __launch_bounds__(256, 2) k_small_idft(SmallIdftArgs g) {
    // row stride 145 points (= 4 dwords mod 64 banks): the transposing drop below writes M1 consecutive rows at one column from
    // consecutive lanes - with the tile's usual 144 (= 0 mod 64) that is an M1-way bank conflict (N = 4096: 0.79 vs 0.46 ms per GiB)
    constexpr int M2 = kSmallM2, RS = kSmallIdftRS;
    constexpr long long m = (long long)M1 * kSmallM2;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x, pl = tid >> 7, t = tid & 127;
    cplx* wl = lds + 2 * M1 * RS;
    cplx* tw1i = wl + M2;   // untwist factors as LDS broadcasts (k_small_inv)
    if (tid < M2) wl[tid] = g.wL2[tid];
    else if (tid < M2 + M1) tw1i[tid - M2] = g.tw1inv[tid - M2];
    const int p = blockIdx.x * 2 + pl;
    const bool active = p < g.npolys;
    const cplx* src = g.S + map_off(g.smap, active ? p : g.npolys - 1) / 2;
    cplx* buf = lds + pl * M1 * RS;
    // consecutive threads read consecutive points q = q1 + M1 q2 and drop them at (q1, q2) of the tile; all M1 loads first (left to
    // itself the compiler waited for each one before its LDS store: 61 of the 63 loads of the N = 4096 kernel sat behind vmcnt(0))
    {
        cplx in[M1];
#pragma unroll
        for (int i = 0; i < M1; ++i) in[i] = src[t + 128 * i];
        // q = t + 128 i: q % M1 = t % M1, q / M1 = t / M1 + (128 / M1) i
        cplx* drop = buf + (t % M1) * RS + t / M1;
#pragma unroll
        for (int i = 0; i < M1; ++i) drop[(128 / M1) * i] = in[i];
    }
    __syncthreads();
    // inverse row pass: 8 lanes per row (8 M1 of the polynomial's 128 threads)
    const int row = t >> 3, o = t & 7;
    if (row < M1) {
        cplx* rowbuf = buf + row * RS;
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
                if (oo > 0) v = cmulc(v, wl[oo * k1]);   // (k1 = 0 reads W^0 = 1: exact)
                rowbuf[k1 * 9 + oo] = v;
            }
        row_sync();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 9 + o];
        Bfly<16, true>::run(u);
        row_sync();
        const cplx* tw = g.tw12t + row * M2 + o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two batches of 8 loads (k_small_inv)
            cplx t8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t8[i] = tw[8 * (8 * h + i)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) rowbuf[o + 8 * (8 * h + i)] = cmulc(u[8 * h + i], t8[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    // inverse column pass: thread t owns column j2 = t
    {
        cplx v[M1];
#pragma unroll
        for (int q1 = 0; q1 < M1; ++q1) v[q1] = buf[q1 * RS + t];
        Bfly<M1, true>::run(v);
        double big = 0.0;   // a SUM, so that a NaN or an infinity selects the saturating conversion (k_small_inv)
#pragma unroll
        for (int j1 = 0; j1 < M1; ++j1) big += fabs(v[j1].x) + fabs(v[j1].y);
        big *= 1.0 / (double)m;
        if (PZ_SMALL_PROBE && g.margin && active) {
            double worst = 0.0;
#pragma unroll
            for (int j1 = 0; j1 < M1; ++j1) {
                const cplx val = cmul(v[j1], tw1i[j1]);
                worst = fmax(worst, fmax(margin_dist(val.x), margin_dist(val.y)));
            }
            margin_note(g.margin, worst);
        }
        if (active) {
            long long* dst = g.res + map_off(g.dmap, p) + t;
            if (big < 2251799813685247.0) {
#pragma unroll
                for (int j1 = 0; j1 < M1; ++j1) {
                    const cplx val = cmul(v[j1], tw1i[j1]);
                    dst[j1 * M2] = fast_i64_from_integral(round_half_away(val.x));
                    dst[m + j1 * M2] = fast_i64_from_integral(round_half_away(val.y));
                }
            } else {
#pragma unroll
                for (int j1 = 0; j1 < M1; ++j1) {
                    const cplx val = cmul(v[j1], tw1i[j1]);
                    dst[j1 * M2] = sat_i64_from_integral(round_half_away(val.x));
                    dst[m + j1 * M2] = sat_i64_from_integral(round_half_away(val.y));
                }
            }
        }
    }
}

struct SmallInvArgs {
    const cplx* S;
    const cplx* Pp;
    long long* res;
    const long long* small;      // key-switch body operand (added to column body_col), may be null
    long long res_bs, small_bs;
    int batch, npi, nrows, ncols, cols_out, ksz;
    int res_cols, res_size, small_cols, small_size, base2k, body_col;
    const cplx* tw12t;
    const cplx* wL2;
    const cplx* tw1inv;          // [16] untwist with 1/m folded in
    int post_rsh;                // AU: the digits leave through vec_znx_rsh_assign by one bit (k_inv_tail<.., RSH>, device_fft.hpp); base2k <= 29
    unsigned au_p;               // AU: Galois element mod 2n (coefficient i goes to i * au_p mod 2n, negated beyond n)
    unsigned au_pinv;            // AU: its inverse mod 2n (set by launch_small_inv)
    int au_mode;                 // AU: 0 phi(normalize(big)), 1 normalize(phi(big) + a), 2 normalize(phi(big) - a), 3 normalize(a - phi(big));
                                 //     a = column `col` of `small` (the key-switch input itself), big includes the body (body_col)
    cplx* S_out;                 // FWD: spectra of the first fwd_limbs limbs of the NEW res column, standard order, [b][limb * cols_out + col]
    const cplx* tw1;             // FWD: twist of the forward column pass
    int fwd_limbs;               // FWD: <= min(KS, res_size)
    int dbg;                     // timing ablation (POULPY_DBG_SMALL_SKIP; results invalid): 1 no key loads, 2 no S loads, 4 no stores, 8 no LDS phases
    unsigned long long* margin;  // rounding-margin probe (margin_note, device_fft.hpp); null = off
    int acc32;                   // NOPROD (blind rotation's tail): bit 0 `small` holds 32-bit digits, bit 1 `res` takes 32-bit digits (bits 2 / 3: 16-bit digits) - same
                                 // element strides, half the bytes: between the blocks of a rotation the accumulator only ever holds normalized
                                 // digits (base2k <= 31), and this kernel moves them at HBM rate (api_br.hip)
};

// 64 M1 threads (1024 at N = 4096: 16 waves, the product phase is latency-bound with fewer); LDS holds the KS output polynomials of one
// (ciphertext, column): KS x M1 rows x 144 points + wL2 — 149.5 KiB at N = 4096, KS = 4 (one workgroup per CU), 75.7 KiB at N = 2048 (two),
// 38.9 KiB at N = 1024 (four).  Variants that kept two workgroups per CU — two limbs in LDS at a time, the other accumulators in registers, or one
// product pass per limb pair — either spilled (128 accumulator registers at 256 threads) or re-read S from HBM (0.26 - 0.36 ms per 1024
// ciphertexts against 0.35 ms for the whole three-kernel pipeline).  Also tried: a persistent workgroup in two roles of 512 threads (role A:
// the next item's product in registers, role B: this item's transforms in the tile, hand-over between barriers) — at the 128-VGPR cap of a
// 1024-thread workgroup role A's 64 accumulator registers leave no room for prefetch slots (spills, 0.93 ms at 4 limbs; 0.25 vs 0.17 ms
// at 3).  And (end of round 2): a persistent 512-thread workgroup (four product positions per thread, 256 registers) with the next item's
// product spread in eight row steps between the stages of the current item's transform - bit-exact, slower: 2.17 vs 3.05 M/s at 4 limbs,
// 3.8 vs 4.07 at 3, key switch 3.5 vs 5.2 (the transform stages are latency-bound and take about twice as long with 8 waves instead of
// 16, which costs more than the hidden loads; 116 - 276 bytes of scratch at 3 - 4 limbs).  And (round 3): the same kernel as TWO 512-thread
// workgroups per CU, each with two limbs in the tile at a time (75 KiB), four product positions per thread, the limb groups through the tile from the
// last limb up with the carries and the waiting accumulators in registers - bit-exact, two workgroups resident, and 20 - 40 % slower (3 limbs 1.55 ->
// 1.88 ms per 10 launches, key switch 1.25 -> 1.64, 4 limbs 2.31 -> 3.27; profiles/r03_small_inv_stamps.txt): 8 waves keep half as many key / spectrum
// requests in flight per workgroup at the 128-register cap, and each transform stage runs on 4 waves.  KS = key limbs (g.ksz).  NOPROD: the spectra of the (ciphertext, column) are given, in the standard device order [q1 + M1 q2]:
// S[b][l * cols_out + col] (npi = ksz * cols_out), no key (the blind rotation's block step produced them).  FWD (blind rotation: the
// result is the accumulator the next block transforms): the digits go back into the tile as doubles and the forward transform of
// k_small_fwd runs on them before the workgroup ends - the next block's k_small_fwd launch and its read of the accumulator are saved.
// AU (glwe_automorphism family, automorphism/glwe_ct.rs:51-275): a thread owns OUTPUT positions, takes the big value's coefficient
// i = position * p^-1 mod 2n from the tile (LDS gather), negates it where phi = X -> X^p wraps, adds / subtracts the operand at its
// position, runs the carry chain and stores the digits - coalesced (rounds 1 - 2 owned the SOURCE coefficient and scattered 8-byte
// stores; round 3: glwe_trace at N = 4096 +24 %).  The key-switch body is added at the source positions by the column pass (coalesced
// loads).  In place (res == a): those loads happen before the barrier in front of the carry phase, i.e. before any thread stores; the
// operand at a thread's own positions is loaded before its first store.
template <int M1, int KS, bool NOPROD = false, bool FWD = false, bool AU = false>
__global__ void __launch_bounds__(64 * M1) k_small_inv(SmallInvArgs g) {
    constexpr int NT = 64 * M1;          // 2 product positions per thread (m = 128 M1 points)
    constexpr int M2 = kSmallM2, RS = small_inv_rs(M1, NOPROD), L = KS;
    constexpr long long m = (long long)M1 * kSmallM2, n = 2 * m;
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x;
    // both columns of a ciphertext on one XCD (consecutive workgroup ids go to consecutive XCDs): the second one finds S in its L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / g.cols_out) * 8 + xcd, col = slot % g.cols_out;
    if (b >= g.batch) return;
    cplx* wl = lds + KS * M1 * RS;
    cplx* tw1i = wl + M2;   // the untwist factors, read as LDS broadcasts by the column pass (as loads from g.tw1inv they were vector loads, one L2 latency each)
    if (tid < M2) wl[tid] = g.wL2[tid];
    else if (tid < M2 + M1) tw1i[tid - M2] = g.tw1inv[tid - M2];
    const int k = g.base2k;
    const int row_max = min(g.nrows, g.npi);
#if PZ_SMALL_STAMP
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_t;
#define PZ_SSTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define PZ_SSTAMP(i)
#endif
    // ---------------- product, one pass over S: acc[l][j] = sum_r S[r][pos_j] * P'[q1_j][r][l * cols_out + col][q2_j], pos_j = tid + NT j ----------------
    const int pq2 = tid & 127, pq1 = tid >> 7;   // q1 = pq1 + (M1 / 2) j
    cplx acc[KS][2];
#pragma unroll
    for (int l = 0; l < KS; ++l)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[l][j] = make_double2(0.0, 0.0);
    if constexpr (NOPROD) {
        // consecutive threads read consecutive points q = q1 + M1 q2 and drop them at (q1, q2) of the tile
        const cplx* Sb = g.S + ((long long)b * g.npi + col) * m + tid;
#pragma unroll
        for (int l = 0; l < KS; ++l)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[l][j] = Sb[(long long)l * g.cols_out * m + NT * j];
#pragma unroll
        for (int l = 0; l < KS; ++l)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = tid + NT * j;
                lds[(l * M1 + (q % M1)) * RS + q / M1] = acc[l][j];
            }
    } else {
        const cplx* Sb = g.S + (long long)b * g.npi * m + tid;
        const long long qstride = (long long)(M1 / 2) * g.nrows * g.ncols * M2;   // q1 advances by M1 / 2 per j
        const long long prow = (long long)g.ncols * M2;
        const long long lstride = (long long)g.cols_out * M2;
        const cplx* kp = g.Pp + ((long long)pq1 * g.nrows * g.ncols + col) * M2 + pq2;
        // one position per group; the next group's loads travel while this one is consumed (two register slots in ping-pong)
        cplx aA, kA[KS], aB, kB[KS];
#define PZ_SMALL_LOAD(A_, K_, R_, J_)                                                              \
    {                                                                                               \
        if (!(PZ_DBG(g.dbg) & 2)) A_ = Sb[(long long)(R_) * m + NT * (J_)];                                 \
        if (!(PZ_DBG(g.dbg) & 1)) { _Pragma("unroll") for (int l = 0; l < KS; ++l) K_[l] = kp[(long long)(R_) * prow + l * lstride + (J_) * qstride]; } \
    }
#define PZ_SMALL_USE(A_, K_, J_)                                                                   \
    {                                                                                               \
        _Pragma("unroll") for (int l = 0; l < KS; ++l) {                                            \
            cplx& c_ = acc[l][(J_)];                                                                \
            c_.x = __builtin_fma(A_.x, K_[l].x, c_.x);                                              \
            c_.x = __builtin_fma(-A_.y, K_[l].y, c_.x);                                             \
            c_.y = __builtin_fma(A_.x, K_[l].y, c_.y);                                              \
            c_.y = __builtin_fma(A_.y, K_[l].x, c_.y);                                              \
        }                                                                                           \
    }
        if (PZ_DBG(g.dbg) & 3) {
            aA = aB = make_double2(1.0, 2.0);
#pragma unroll
            for (int l = 0; l < KS; ++l) kA[l] = kB[l] = make_double2(0.5, 0.25);
        }
        // (measured and dropped, round 3: warming L2 with the ciphertext's spectra by one dword load per 128-byte line at the top of the
        // kernel - 7 - 10 % slower at every ring degree, profiles/r03_small_inv_stamps.txt)
        // (static priority for the second-dispatched half of the waves, which finishes this phase late - 28 k vs 17 k cycles: no gain, round 3)
        // the sched_barriers keep the two slots in ping-pong: without them the machine scheduler gathers the ten loads of a row at the top of
        // the iteration and the wave drains to vmcnt(0) once per row (round 3: seen in the ISA, product phase 29 k cycles per item at N = 4096)
        PZ_SMALL_LOAD(aA, kA, 0, 0)
        for (int r = 0; r < row_max; ++r) {
            const int rn = min(r + 1, row_max - 1);   // (the last prefetch is simply unused)
            PZ_SMALL_LOAD(aB, kB, r, 1)
            __builtin_amdgcn_sched_barrier(0);
            PZ_SMALL_USE(aA, kA, 0)
            __builtin_amdgcn_sched_barrier(0);
            PZ_SMALL_LOAD(aA, kA, rn, 0)
            __builtin_amdgcn_sched_barrier(0);
            PZ_SMALL_USE(aB, kB, 1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef PZ_SMALL_LOAD
#undef PZ_SMALL_USE
    }
    PZ_SSTAMP(0)
    if constexpr (!NOPROD) {
#pragma unroll
        for (int l = 0; l < KS; ++l)
#pragma unroll
            for (int j = 0; j < 2; ++j) lds[(l * M1 + pq1 + (M1 / 2) * j) * RS + pq2] = acc[l][j];
    }
    PZ_SSTAMP(1)
    __syncthreads();
    PZ_SSTAMP(2)
    // ---------------- inverse row pass (k_mid128) of the KS polynomials, x conj tw12, back into the tile as T2[q1][j2] ----------------
    {
        const int rp = tid / (8 * M1), rrow = (tid % (8 * M1)) >> 3, ro = tid & 7;   // 8 M1 threads per polynomial
        if (rp < KS) {
            cplx* rowbuf = lds + (rp * M1 + rrow) * RS;
            cplx u[16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) u[8 * h + k2] = rowbuf[ro + 8 * h + 16 * k2];
#if PZ_SMALL_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); PZ_SSTAMP(8)
#endif
            Bfly<8, true>::run(u);
            Bfly<8, true>::run(u + 8);
#if PZ_SMALL_STAMP
            __builtin_amdgcn_sched_barrier(0); PZ_SSTAMP(9)
#endif
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) {
                    cplx v = u[8 * h + oo];
                    const int k1 = ro + 8 * h;
                    if (k1 > 0 && oo > 0) v = cmulc(v, wl[oo * k1]);
                    rowbuf[k1 * 9 + oo] = v;
                }
            row_sync();
#if PZ_SMALL_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); PZ_SSTAMP(10)
#endif
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 9 + ro];
#if PZ_SMALL_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); PZ_SSTAMP(11)
#endif
            Bfly<16, true>::run(u);
#if PZ_SMALL_STAMP
            __builtin_amdgcn_sched_barrier(0); PZ_SSTAMP(12)
#endif
            row_sync();
            // the 16 inter-pass twiddles in two batches of 8 (left to itself the compiler loads them one by one behind vmcnt(0): 16 L2 latencies)
            const cplx* tw = g.tw12t + rrow * M2 + ro;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                cplx t8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t8[i] = tw[8 * (8 * h + i)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) rowbuf[ro + 8 * (8 * h + i)] = cmulc(u[8 * h + i], t8[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    PZ_SSTAMP(3)
    __syncthreads();
    PZ_SSTAMP(6)
    // coordinates of the carry phase (thread = (column j2, component, j1 group), 4 coefficients per limb) and the key-switch body / per-column
    // operand of those coefficients, all limbs, requested HERE, between the row and the column pass (in front of the row pass the 8 KS registers spill / cost the N = 2048 kernel its second workgroup per CU): in the blind rotation's tail this is the previous accumulator, fresh from HBM -
    // requested at the top of the carry phase it cost that phase 4 - 8 k cycles of plain waiting (round 3 stamps)
    constexpr int JG = M1 / 4;   // thread groups over j1 (NT / 256): this thread's outputs are j1 = JG e + jq, e < 4
    const int cj2 = tid & 127, ch = (tid >> 7) & 1, jq = tid >> 8;   // component 0: coefficients j < m, 1: j >= m
    const long long* small_col =
        (g.small && (col == g.body_col || g.body_col < 0))   // body_col < 0: every column gets its own column of `small`
            ? g.small + (long long)b * g.small_bs + (g.body_col < 0 ? (long long)col * n : 0) + (ch ? m : 0) + cj2 + (long long)jq * M2
            : nullptr;
    const long long small_ls = (long long)g.small_cols * n;
    long long smv[AU ? 1 : KS][4];
    if constexpr (!AU) {
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) smv[j][e] = 0;
        if (NOPROD && (g.acc32 & 4) && small_col && g.small_size > 0) {   // 16-bit digits at the same element offsets (base2k <= 15, round 6)
            const short* sc16 = reinterpret_cast<const short*>(g.small) + (small_col - g.small);
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) smv[j][e] = (long long)sc16[(long long)min(j, g.small_size - 1) * small_ls + JG * e * M2];
        } else
        if (NOPROD && (g.acc32 & 1) && small_col && g.small_size > 0) {   // 32-bit digits at the same element offsets
            const int* sc32 = reinterpret_cast<const int*>(g.small) + (small_col - g.small);
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) smv[j][e] = (long long)sc32[(long long)min(j, g.small_size - 1) * small_ls + JG * e * M2];
        } else if (small_col && g.small_size > 0) {
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) smv[j][e] = small_col[(long long)min(j, g.small_size - 1) * small_ls + JG * e * M2];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---------------- inverse column pass + rounding, NT / 128 limbs at a time: thread = (limb, column j2); the 2 M1 integers go back
    // into the tile in place of the column's M1 complex values (same bytes) ----------------
#pragma unroll
    for (int cl0 = 0; cl0 < KS; cl0 += NT / 128) {
        const int cl = cl0 + (tid >> 7), cj = tid & 127;
        if (cl < KS) {
            cplx v[M1];
#pragma unroll
            for (int q1 = 0; q1 < M1; ++q1) v[q1] = lds[(cl * M1 + q1) * RS + cj];
            Bfly<M1, true>::run(v);
            // |component of v * tw1inv| <= (|v.x| + |v.y|) / m <= big: below 2^51 the 3-instruction conversion is exact.  A SUM, so that a
            // NaN or an infinity anywhere reaches `big` (fmax would drop a NaN) and selects the saturating conversion
            double big = 0.0;
#pragma unroll
            for (int j1 = 0; j1 < M1; ++j1) big += fabs(v[j1].x) + fabs(v[j1].y);
            big *= 1.0 / (double)m;
            longlong2* out = reinterpret_cast<longlong2*>(lds);
            // AU: the key-switch body joins the big value here, at the SOURCE positions this thread holds (coalesced loads, four j1 at a time);
            // the carry phase then owns output positions and has nothing left to gather from global memory.  In place: read before the barrier
            const long long* bsrc = nullptr;
            if constexpr (AU) {
                if (g.small && (col == g.body_col || g.body_col < 0) && cl < g.small_size)
                    bsrc = g.small + (long long)b * g.small_bs + (g.body_col < 0 ? (long long)col * n : 0) + (long long)cl * g.small_cols * n + cj;
            }
            if (PZ_SMALL_PROBE && g.margin) {   // rounding-margin probe: the values PZ_SMALL_ROUND rounds
                double worst = 0.0;
#pragma unroll
                for (int j1 = 0; j1 < M1; ++j1) {
                    const cplx val = cmul(v[j1], tw1i[j1]);
                    worst = fmax(worst, fmax(margin_dist(val.x), margin_dist(val.y)));
                }
                margin_note(g.margin, worst);
            }
#define PZ_SMALL_ROUND(CONVERT)                                                                               \
    _Pragma("unroll") for (int jb = 0; jb < M1; jb += 4) {                                                   \
        long long bx_[4] = {0, 0, 0, 0}, by_[4] = {0, 0, 0, 0};                                              \
        if (AU && bsrc) {                                                                                    \
            _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { bx_[t_] = bsrc[(jb + t_) * M2]; by_[t_] = bsrc[m + (jb + t_) * M2]; } \
        }                                                                                                    \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                   \
            const int j1 = jb + t_;                                                                          \
            const cplx val = cmul(v[j1], tw1i[j1]);                                                          \
            out[(cl * M1 + j1) * RS + cj] = make_longlong2((long long)((unsigned long long)CONVERT(round_half_away(val.x)) + (unsigned long long)bx_[t_]), \
                                                           (long long)((unsigned long long)CONVERT(round_half_away(val.y)) + (unsigned long long)by_[t_])); \
        }                                                                                                    \
    }
            if (big < 2251799813685247.0) {   // 2^51 - 1 (false for NaN too)
                PZ_SMALL_ROUND(fast_i64_from_integral)
            } else {
                PZ_SMALL_ROUND(sat_i64_from_integral)
            }
#undef PZ_SMALL_ROUND
        }
    }
    PZ_SSTAMP(4)
    __syncthreads();
    PZ_SSTAMP(7)
    // ---------------- (+ key-switch body), carry chain from the last limb up, stores: thread = (column j2, component, j1 parity), 8 coefficients ----------------
    long long carry[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) carry[u] = 0;
    long long* res_col = g.res + (long long)b * g.res_bs + (long long)col * n + (ch ? m : 0) + cj2 + (long long)jq * M2;
    const long long res_ls = (long long)g.res_cols * n;
    // AU: the thread owns four OUTPUT positions (consecutive lanes, consecutive positions: stores and operand loads coalesce) and fetches
    // their source coefficients i = position * p^-1 mod 2n (negated where that product is >= n) from the tile - an LDS gather instead of
    // the 8-byte global scatters of round 2 (the key-switch body, which sits at the SOURCE positions, joined the tile in the column pass)
    long long opos[4];
    bool oneg[4];
    int lsrc[4];   // index of the source coefficient's limb-0 value in the tile (as long long)
    long long auv[AU ? KS : 1][4];   // operand of this thread's positions, all limbs, loaded in one batch (in place: before this thread's stores)
    const long long* au_col = nullptr;
    if constexpr (AU) {
        const unsigned n2m = 2u * (unsigned)n - 1u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned jo = (unsigned)((ch ? m : 0) + (long long)(JG * e + jq) * M2 + cj2);   // output position
            const unsigned i0 = (jo * g.au_pinv) & n2m;
            oneg[e] = i0 >= (unsigned)n;
            const unsigned i = i0 & ((unsigned)n - 1u), ich = i >= (unsigned)m ? 1u : 0u, ii = i - ich * (unsigned)m;
            opos[e] = (long long)jo;
            lsrc[e] = 2 * (int)((ii >> 7) * RS + (ii & 127u)) + (int)ich;
        }
        res_col = g.res + (long long)b * g.res_bs + (long long)col * n;
        if (g.au_mode != 0 && g.small) au_col = g.small + (long long)b * g.small_bs + (long long)col * n;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) auv[j][e] = 0;
        if (au_col && g.small_size > 0) {
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) auv[j][e] = au_col[(long long)min(j, g.small_size - 1) * small_ls + opos[e]];
        }
#pragma unroll
        for (int j = 0; j < KS; ++j)
            if (j >= g.small_size)
#pragma unroll
                for (int e = 0; e < 4; ++e) auv[j][e] = 0;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { opos[e] = JG * e * M2; oneg[e] = false; lsrc[e] = 2 * ((JG * e + jq) * RS + cj2) + ch; }
    }
    // (the operand's loads were issued in front of the row pass) limbs beyond its size: the last one was read again, masked here
    if constexpr (!AU) {
#pragma unroll
        for (int j = 0; j < KS; ++j)
            if (!(small_col && j < g.small_size))
#pragma unroll
                for (int e = 0; e < 4; ++e) smv[j][e] = 0;
    }
    // limbs of res beyond the precision of the big value are zero (normalize.rs:118-120); shifted stores: limb L receives the bit shifted
    // out of limb L - 1
#if PZ_SMALL_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); PZ_SSTAMP(13)   // operand / body loads have arrived
#endif
    const bool rsh = AU && g.post_rsh;
    int cy2[4] = {0, 0, 0, 0};
    const bool res32 = NOPROD && (g.acc32 & 2), res16 = NOPROD && (g.acc32 & 8);
    int* res_col32 = reinterpret_cast<int*>(g.res) + (res_col - g.res);
    short* res_col16 = reinterpret_cast<short*>(g.res) + (res_col - g.res);
    for (int j = L + (rsh ? 1 : 0); j < g.res_size; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (res16) res_col16[(long long)j * res_ls + opos[e]] = 0;
            else if (res32) res_col32[(long long)j * res_ls + opos[e]] = 0;
            else res_col[(long long)j * res_ls + opos[e]] = 0;
        }
    const unsigned long long half = 1ull << (k - 1), mask = (1ull << k) - 1;
    const long long* xin = reinterpret_cast<const long long*>(lds);
    // One limb at a time: the chain steps of the thread's four coefficients in straight-line code, then the stores / the FWD tile writes
    // behind ONE uniform test each.  (Rounds 1 - 3 tested "carry only" - a last limb that is not stored - and `writes` per coefficient:
    // a branch per (limb, coefficient) in the unrolled chain; the carry-only case needs no test at all, the chain starts from carry 0, for
    // which the two-step form gives the same carry.)
#pragma unroll
    for (int j = L - 1; j >= 0; --j) {
        const bool writes = j < g.res_size;
        long long x1v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            long long x = xin[lsrc[e] + 2 * j * M1 * RS];
            if constexpr (AU) {
                unsigned long long ux = (unsigned long long)x;   // (the body was added at the source position, in the column pass)
                if (g.au_mode != 0) {   // phi on the big value, then +- a over the common limbs (limbs beyond a: + 0, - 0, 0 - big)
                    if (oneg[e]) ux = 0ull - ux;
                    const unsigned long long aj = (unsigned long long)auv[j][e];
                    ux = g.au_mode == 1 ? ux + aj : (g.au_mode == 2 ? ux - aj : aj - ux);
                }
                x = (long long)ux;
            } else x = (long long)((unsigned long long)x + (unsigned long long)smv[j][e]);
            long long& cy = carry[e];
            const unsigned long long y = (unsigned long long)x + half;
            const long long d = (long long)(y & mask) - (long long)half;
            const long long cr = (long long)y >> k;
            const unsigned long long y2 = (unsigned long long)d + (unsigned long long)cy + half;
            x1v[e] = (long long)(y2 & mask) - (long long)half;
            cy = (long long)((unsigned long long)cr + (unsigned long long)((long long)y2 >> k));
        }
        if (writes) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long x1 = x1v[e];
                if (!(PZ_DBG(g.dbg) & 4) || x1 == 0x7fffffffffffLL) {
                    // AU mode 0: phi acts on the normalized digits (glwe_ct.rs:69-71)
                    const long long xs = (AU && g.au_mode == 0 && oneg[e]) ? (long long)(0ull - (unsigned long long)x1) : x1;
                    if constexpr (AU) {   // (plain stores: in place the operand's lines are re-read by the next trace step)
                        if (rsh) {
                            const int xd = (int)xs, d1 = -(xd & 1), cr1 = (xd - d1) >> 1;
                            if (j == g.res_size - 1) {
                                cy2[e] = cr1;
                            } else {
                                const int dpc = d1 * (1 << (k - 1)) + cy2[e];
                                const int nv = sx_digit(k, dpc);
                                cy2[e] = cr1 + sx_carry(k, dpc, nv);
                                res_col[(long long)(j + 1) * res_ls + opos[e]] = (long long)nv;
                            }
                            if (j == 0) res_col[opos[e]] = (long long)sx_digit(k, cy2[e]);
                        } else {
                            res_col[(long long)j * res_ls + opos[e]] = xs;
                        }
                    } else if (res16) res_col16[(long long)j * res_ls + opos[e]] = (short)xs;
                    else if (res32) res_col32[(long long)j * res_ls + opos[e]] = (int)xs;
                    else st_stream(res_col + (long long)j * res_ls + opos[e], xs);
                }
            }
        }
        if constexpr (FWD) {   // this thread's own slots (read above): component ch of z[limb j][j1][j2]
            if (j < g.fwd_limbs) {
#pragma unroll
                for (int e = 0; e < 4; ++e) reinterpret_cast<double*>(lds)[2 * ((j * M1 + JG * e + jq) * RS + cj2) + ch] = (double)x1v[e];
            }
        }
    }
    PZ_SSTAMP(5)
#if PZ_SMALL_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PZ_SSTAMP(14)   // stores acknowledged
#endif
#if PZ_SMALL_STAMP
    if ((tid & 63) == 0 && blockIdx.x == 257)
        printf("SSTAMP wg %d wave %2d t0 %llu total %llu | product %llu accwr %llu bar1 %llu rowpass [rd %llu b8 %llu tw+wr %llu rd %llu b16 %llu tw+wr %llu] bar2 %llu colpass %llu bar3 %llu carry [opnd %llu chain %llu storeack %llu]\n", (int)blockIdx.x, tid >> 6,
               (unsigned long long)(st_t0 & 0xffffffull), (unsigned long long)(st_t - st_t0), st_acc[0], st_acc[1], st_acc[2], st_acc[8], st_acc[9], st_acc[10], st_acc[11], st_acc[12], st_acc[3], st_acc[6], st_acc[4], st_acc[7], st_acc[13], st_acc[5], st_acc[14]);
#endif
#undef PZ_SSTAMP
    if constexpr (FWD) {
        __syncthreads();
        // ---------------- forward column pass (k_small_fwd), NT / 128 limbs at a time: thread = (limb, column j2) ----------------
        for (int cl0 = 0; cl0 < g.fwd_limbs; cl0 += NT / 128) {
            const int cl = cl0 + (tid >> 7), cj = tid & 127;
            if (cl < g.fwd_limbs) {
                cplx v[M1];
#pragma unroll
                for (int j1 = 0; j1 < M1; ++j1) v[j1] = cmul(lds[(cl * M1 + j1) * RS + cj], g.tw1[j1]);
                Bfly<M1, false>::run(v);
#pragma unroll
                for (int q1 = 0; q1 < M1; ++q1) lds[(cl * M1 + q1) * RS + cj] = cmul(v[q1], g.tw12t[q1 * M2 + cj]);
            }
        }
        __syncthreads();
        // ---------------- forward row pass (k_small_fwd / k_mid128), 8 lanes per row; standard spectrum order out ----------------
        const int rp = tid / (8 * M1), rrow = (tid % (8 * M1)) >> 3, ro = tid & 7;
        if (rp < g.fwd_limbs) {
            cplx* rowbuf = lds + (rp * M1 + rrow) * RS;
            cplx x[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) x[n1] = rowbuf[ro + 8 * n1];
            row_sync();
            Bfly<16, false>::run(x);
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                cplx v = x[k1];
                if (k1 > 0) v = cmul(v, wl[ro * k1]);
                rowbuf[k1 * 9 + ro] = v;
            }
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) x[8 * h + oo] = rowbuf[(ro + 8 * h) * 9 + oo];
            Bfly<8, false>::run(x);
            Bfly<8, false>::run(x + 8);
            cplx* dst = g.S_out + ((long long)b * g.fwd_limbs * g.cols_out + (long long)rp * g.cols_out + col) * m + rrow + M1 * ro;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) dst[(8 * h + 16 * k2) * M1] = x[8 * h + k2];
        }
    }
}


}  // namespace pz
