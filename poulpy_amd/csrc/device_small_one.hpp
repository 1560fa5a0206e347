// device_small_one.hpp — ONE kernel per GLWE product for N = 1024 / 2048 (round 6): forward transforms, product, inverse
// transforms and carry chains of a ciphertext in one workgroup - the spectra never reach HBM (the two-kernel pipeline of device_small.hpp writes
// them once and reads them once per output column; the three-kernel pipeline crosses HBM three times).
//
// Why these rings and not N = 4096 (the capacity argument, all three resources of a CU - DESIGN.md 4.3):
//   * LDS: the 8 input polynomials of a rank-1, 4-limb ciphertext are 8 x M1 rows x 144 points x 16 B = 73.7 KiB (N = 1024), 147 KiB (N = 2048) and
//     295 KiB at N = 4096 against 160 KiB.  In two groups of four the product needs its accumulators in registers:
//   * registers: 8 output polynomials x 2048 points = 16 complex sums per thread at 1024 threads (64 of the 128 registers a thread has at 16 waves per
//     CU) beside a radix-16 column pass that holds 16 complex values + 32 incoming i64 (~100 registers): it spills; at 512 threads the sums alone
//     are 128 registers;
//   * key stream: one ciphertext per workgroup means every key value is fetched per ciphertext - 2 MiB from L2 at the ~50 B/clk a CU sustains =
//     42 k cycles per ciphertext, as long as its whole HBM traffic takes (512 KiB at 1/256 of 5.5 TB/s = 51 k cycles); the three-kernel pipeline's
//     tile (8 ciphertexts x 8 polynomials x one frequency row) fetches each key value once per EIGHT ciphertexts.  Sharing a key value between
//     ciphertexts inside one workgroup multiplies the accumulator registers by the ciphertext count.
// At N <= 2048 none of the three binds: the inputs of a ciphertext fit the tile, the sums are 8 - 16 complex values per thread, the key is
// 0.5 - 1 MiB per ciphertext.
//
// Workgroup = 512 threads = one ciphertext (rank 1: 2 output columns, <= 4 key limbs, <= 8 input polynomials, one base2k, dsize 1):
//   A  forward column pass   thread = (polynomial, column j2): i64 -> f64, twist, radix-M1 butterfly, x tw12 -> tile[p][q1][j2]     (k_small_fwd)
//   B  forward row pass      8 lanes per row, spectrum S[p][q1][q2] in place                                                       (k_mid128)
//   C  product               thread = M1 / 4 frequency points; sum_r S[r] x P'[q1][r][c][q2] for all 2 KS output polynomials in registers
//   both output columns: D inverse row pass, E inverse column pass + rounding, F carry chain from the last limb up + stores         (k_small_inv)
// Same tables, same stage formulas as the two-kernel pipeline; the results are the same i64 limbs (tests/test_gpu_parity.py, pool tests).
#pragma once
#include "device_small.hpp"
#ifndef PZ_SMALL_ONE_STAMP
#define PZ_SMALL_ONE_STAMP 0
#endif

namespace pz {

struct SmallOneArgs {
    const long long* src;        // input limbs; polynomial p of ciphertext b at src + map_off(smap, b * npi + p)
    PolyMap smap;
    const cplx* Pp;              // P'[q1][r][c][q2]
    long long* res;
    const long long* small;      // key-switch body operand (added to column body_col; body_col < 0: every column its own), may be null
    long long res_bs, small_bs;
    int batch, npi, nrows, ncols, ksz;
    int res_cols, res_size, small_cols, small_size, base2k, body_col;
    const cplx* tw1;             // [M1] twist of the forward column pass
    const cplx* tw12t;           // [q1][j2]
    const cplx* wL2;             // exp(2 pi i t / 128)
    const cplx* tw1inv;          // [M1] untwist with 1/m folded in
    unsigned long long* margin;  // rounding-margin probe (margin_note); null = off
};

// (Measured and dropped, round 6: input groups of 4 polynomials through a 32-row tile at N = 2048 - 73.7 KiB, two workgroups per CU, the product's sums
//  waiting in registers between the groups: at the 128-register cap of 16 waves per CU the kernel spills 76 - 640 B per lane, and the group loop alone
//  costs the N = 1024 forms 12 - 112 B.  One group, the whole ciphertext in the tile.)
// (Also measured and dropped: the i64 loads of both column-pass sweeps requested before the first butterfly, and the key values of the product's first row
//  requested in front of the forward row pass - N = 1024: 16.55 -> 16.5 M external products/s, key switch 20.7 -> 19.9 M/s, 2 limbs 30.9 -> 29.5 M/s; N = 2048
//  7.9 -> 7.8 M/s: with two workgroups per CU the other workgroup already fills those waits, and the extra live registers cost; profiles/r06_ab_small_one.txt)
template <int M1, int KS>
__global__ void __launch_bounds__(512, (M1 == 4 ? 4 : 2)) k_small_one(SmallOneArgs g) {   // (waves per SIMD: two workgroups per CU at N = 1024, one at 2048)
    constexpr int NT = 512, M2 = kSmallM2, RS = kSmallRS, CO = 2, NPO = CO * KS, PP = M1 / 4;
    constexpr long long m = (long long)M1 * M2, n = 2 * m;
    static_assert(M1 == 4 || M1 == 8, "one-kernel product: N = 1024 / 2048 (see the capacity argument above)");
    extern __shared__ cplx lds[];   // tile: 8 polynomials x M1 rows x RS | wL2[128] | tw1inv[M1]
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    cplx* wl = lds + 8 * M1 * RS;
    cplx* tw1i = wl + M2;
    if (tid < M2) wl[tid] = g.wL2[tid];
    else if (tid < M2 + M1) tw1i[tid - M2] = g.tw1inv[tid - M2];
    const int k = g.base2k;
    const int row_max = min(g.nrows, g.npi);
#if PZ_SMALL_ONE_STAMP   // diagnostic build: per-phase s_memtime totals of one workgroup (tools/dbg/small_one_stamps.sh; NOTEBOOK.md 15.9)
    unsigned long long so_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long so_t = __builtin_amdgcn_s_memtime();
    const unsigned long long so_t0 = so_t;
#define PZ_OSTAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); so_acc[i] += t_ - so_t; so_t = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PZ_OSTAMP(i)
#endif
    // ---------------- A: forward column pass (k_small_fwd): 4 polynomials per sweep ----------------
    {
        const int t = tid & 127;
        for (int p = tid >> 7; p < g.npi; p += NT / 128) {
            const long long* a = g.src + map_off(g.smap, b * g.npi + p);
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
            for (int q1 = 0; q1 < M1; ++q1) lds[(p * M1 + q1) * RS + t] = cmul(v[q1], g.tw12t[q1 * M2 + t]);
        }
    }
    PZ_OSTAMP(0)
    __syncthreads();
    PZ_OSTAMP(1)
    // ---------------- B: forward row pass, 8 lanes per row; the spectrum stays in the tile as S[p][q1][q2] ----------------
    {
        const int row = tid >> 3, o = tid & 7;
        if (row < g.npi * M1) {
            cplx* rowbuf = lds + row * RS;
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
            row_sync();   // every lane of the row has read its exchange values: the row can take the spectrum
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 8 * h + 16 * k2] = x[8 * h + k2];
        }
    }
    PZ_OSTAMP(2)
    __syncthreads();
    PZ_OSTAMP(3)
    // ---------------- C: product, all NPO output polynomials of the ciphertext in registers: acc[c][j] = sum_r S[r][pos_j] * P'[q1_j][r][c][q2] ----------------
    const int pq2 = tid & 127, pq1 = tid >> 7;   // q1_j = pq1 + 4 j
    cplx acc[NPO][PP];
#pragma unroll
    for (int c = 0; c < NPO; ++c)
#pragma unroll
        for (int j = 0; j < PP; ++j) acc[c][j] = make_double2(0.0, 0.0);
    {
        const long long qstride = (long long)4 * g.nrows * g.ncols * M2;   // q1 advances by 4 per j
        const long long prow = (long long)g.ncols * M2;
        const cplx* kp = g.Pp + ((long long)pq1 * g.nrows * g.ncols) * M2 + pq2;
        const cplx* ap = lds + pq1 * RS + pq2;
        // two register slots in ping-pong (k_small_inv): the next group's key values travel while this one is consumed
        cplx aA[PP], kA[NPO][PP], aB[PP], kB[NPO][PP];
#define PZ_ONE_LOAD(A_, K_, R_)                                                                      \
    {                                                                                               \
        _Pragma("unroll") for (int j = 0; j < PP; ++j) {                                            \
            A_[j] = ap[((R_) * M1 + 4 * j) * RS];                                                   \
            _Pragma("unroll") for (int c = 0; c < NPO; ++c) K_[c][j] = kp[(long long)(R_) * prow + c * M2 + j * qstride]; \
        }                                                                                           \
    }
#define PZ_ONE_USE(A_, K_)                                                                           \
    {                                                                                               \
        _Pragma("unroll") for (int j = 0; j < PP; ++j)                                              \
            _Pragma("unroll") for (int c = 0; c < NPO; ++c) {                                       \
                cplx& c_ = acc[c][j];                                                               \
                c_.x = __builtin_fma(A_[j].x, K_[c][j].x, c_.x);                                    \
                c_.x = __builtin_fma(-A_[j].y, K_[c][j].y, c_.x);                                   \
                c_.y = __builtin_fma(A_[j].x, K_[c][j].y, c_.y);                                    \
                c_.y = __builtin_fma(A_[j].y, K_[c][j].x, c_.y);                                    \
            }                                                                                       \
    }
        PZ_ONE_LOAD(aA, kA, 0)
        for (int r = 0; r < row_max; r += 2) {
            const int r1 = min(r + 1, row_max - 1), r2 = min(r + 2, row_max - 1);   // (the last prefetches are simply unused)
            PZ_ONE_LOAD(aB, kB, r1)
            __builtin_amdgcn_sched_barrier(0);
            PZ_ONE_USE(aA, kA)
            __builtin_amdgcn_sched_barrier(0);
            PZ_ONE_LOAD(aA, kA, r2)
            __builtin_amdgcn_sched_barrier(0);
            if (r + 1 < row_max) PZ_ONE_USE(aB, kB)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef PZ_ONE_LOAD
#undef PZ_ONE_USE
    }
    // ---------------- BOTH output columns through the tile at once (2 KS polynomials x M1 rows <= the tile's 8 M1): D inverse row pass, E inverse column pass +
    // rounding, F carry chain + stores (k_small_inv's stages).  (First version: one column at a time - eight barriers and a quarter to a half of the threads
    // busy in D and F; this form: four barriers.)  Tile slot of output polynomial (limb l, column col) = col * KS + l ----------------
    constexpr int JG = M1 / 4;         // thread groups over j1: a chain thread's outputs are j1 = JG e + jq, e < 4
    constexpr int CT_N = 64 * M1;      // chain threads per column
    constexpr int CPR = NT / CT_N;     // columns per chain round: 2 at N = 1024, 1 at N = 2048
    const unsigned long long half = 1ull << (k - 1), mask = (1ull << k) - 1;
    PZ_OSTAMP(4)
    __syncthreads();   // every spectrum value has been read
#pragma unroll
    for (int c = 0; c < NPO; ++c)
#pragma unroll
        for (int j = 0; j < PP; ++j) lds[(((c % CO) * KS + c / CO) * M1 + pq1 + 4 * j) * RS + pq2] = acc[c][j];
    __syncthreads();
    PZ_OSTAMP(5)
    // ---- D: inverse row pass of the 2 KS polynomials, x conj tw12, back into the tile as T2[q1][j2]
    {
        const int rp = tid / (8 * M1), rrow = (tid % (8 * M1)) >> 3, ro = tid & 7;   // 8 M1 threads per polynomial: 16 / 8 polynomials per sweep
        if (rp < NPO) {
            cplx* rowbuf = lds + (rp * M1 + rrow) * RS;
            cplx u[16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) u[8 * h + k2] = rowbuf[ro + 8 * h + 16 * k2];
            Bfly<8, true>::run(u);
            Bfly<8, true>::run(u + 8);
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
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 9 + ro];
            Bfly<16, true>::run(u);
            row_sync();
            const cplx* tw = g.tw12t + rrow * M2 + ro;
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // the 16 inter-pass twiddles in two batches of 8 (k_small_inv)
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
    PZ_OSTAMP(6)
    __syncthreads();
    PZ_OSTAMP(7)
    // ---- E: inverse column pass + rounding: thread = (polynomial slot, column j2), 4 slots per sweep; the 2 M1 integers take the place of the column's M1 complex values
#pragma unroll
    for (int cl0 = 0; cl0 < NPO; cl0 += NT / 128) {
        const int cl = cl0 + (tid >> 7), cj = tid & 127;
        if (cl < NPO) {
            cplx v[M1];
#pragma unroll
            for (int q1 = 0; q1 < M1; ++q1) v[q1] = lds[(cl * M1 + q1) * RS + cj];
            Bfly<M1, true>::run(v);
            double big = 0.0;   // a SUM: a NaN or an infinity anywhere selects the saturating conversion (k_small_inv)
#pragma unroll
            for (int j1 = 0; j1 < M1; ++j1) big += fabs(v[j1].x) + fabs(v[j1].y);
            big *= 1.0 / (double)m;
            longlong2* out = reinterpret_cast<longlong2*>(lds);
            if (PZ_SMALL_PROBE && g.margin) {   // rounding-margin probe: the values rounded below
                double worst = 0.0;
#pragma unroll
                for (int j1 = 0; j1 < M1; ++j1) {
                    const cplx val = cmul(v[j1], tw1i[j1]);
                    worst = fmax(worst, fmax(margin_dist(val.x), margin_dist(val.y)));
                }
                margin_note(g.margin, worst);
            }
            if (big < 2251799813685247.0) {   // 2^51 - 1 (false for NaN too)
#pragma unroll
                for (int j1 = 0; j1 < M1; ++j1) {
                    const cplx val = cmul(v[j1], tw1i[j1]);
                    out[(cl * M1 + j1) * RS + cj] = make_longlong2(fast_i64_from_integral(round_half_away(val.x)), fast_i64_from_integral(round_half_away(val.y)));
                }
            } else {
#pragma unroll
                for (int j1 = 0; j1 < M1; ++j1) {
                    const cplx val = cmul(v[j1], tw1i[j1]);
                    out[(cl * M1 + j1) * RS + cj] = make_longlong2(sat_i64_from_integral(round_half_away(val.x)), sat_i64_from_integral(round_half_away(val.y)));
                }
            }
        }
    }
    PZ_OSTAMP(8)
    __syncthreads();
    PZ_OSTAMP(9)
    // ---- F: (+ key-switch body), carry chain from the last limb up, stores: thread = (column of the ciphertext, column j2, component, j1 group), 4 coefficients
    // per limb; no barrier between the rounds (the chains only read the tile)
    {
        const int tl = tid % CT_N;
        const int cj2 = tl & 127, ch = (tl >> 7) & 1, jq = tl >> 8;   // component 0: coefficients j < m, 1: j >= m
        const long long* xin = reinterpret_cast<const long long*>(lds);
#pragma unroll
        for (int c0 = 0; c0 < CO; c0 += CPR) {
            const int col = c0 + tid / CT_N;
            const long long* small_col =
                (g.small && (col == g.body_col || g.body_col < 0))
                    ? g.small + (long long)b * g.small_bs + (g.body_col < 0 ? (long long)col * n : 0) + (ch ? m : 0) + cj2 + (long long)jq * M2
                    : nullptr;
            const long long small_ls = (long long)g.small_cols * n;
            long long smv[KS][4];
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) smv[j][e] = 0;
            if (small_col && g.small_size > 0) {   // the body operand of this thread's coefficients, all limbs, in one batch
#pragma unroll
                for (int j = 0; j < KS; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) smv[j][e] = small_col[(long long)min(j, g.small_size - 1) * small_ls + JG * e * M2];
            }
            __builtin_amdgcn_sched_barrier(0);
            long long carry[4] = {0, 0, 0, 0};
            long long* res_col = g.res + (long long)b * g.res_bs + (long long)col * n + (ch ? m : 0) + cj2 + (long long)jq * M2;
            const long long res_ls = (long long)g.res_cols * n;
            // limbs of res beyond the precision of the big value are zero (normalize.rs:118-120)
            for (int j = KS; j < g.res_size; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) res_col[(long long)j * res_ls + JG * e * M2] = 0;
#pragma unroll
            for (int j = KS - 1; j >= 0; --j) {
                const bool has_body = small_col && j < g.small_size;
                long long x1v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    long long x = xin[2 * (((col * KS + j) * M1 + JG * e + jq) * RS + cj2) + ch];
                    if (has_body) x = (long long)((unsigned long long)x + (unsigned long long)smv[j][e]);
                    long long& cy = carry[e];
                    const unsigned long long y = (unsigned long long)x + half;
                    const long long d = (long long)(y & mask) - (long long)half;
                    const long long cr = (long long)y >> k;
                    const unsigned long long y2 = (unsigned long long)d + (unsigned long long)cy + half;
                    x1v[e] = (long long)(y2 & mask) - (long long)half;
                    cy = (long long)((unsigned long long)cr + (unsigned long long)((long long)y2 >> k));
                }
                if (j < g.res_size) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) st_stream(res_col + (long long)j * res_ls + JG * e * M2, x1v[e]);
                }
            }
        }
    }
#if PZ_SMALL_ONE_STAMP
    PZ_OSTAMP(10)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PZ_OSTAMP(11)
    if ((tid & 63) == 0 && (blockIdx.x == 300 || blockIdx.x == 3000))
        printf("OSTAMP wg %d wave %d total %llu | A colpass(loads+bfly) %llu bar %llu | B rowpass %llu bar %llu | C product(key from L2) %llu | bar+acc->tile %llu | D inv rowpass %llu bar %llu | E inv colpass %llu bar %llu | F chains+stores %llu storeack %llu\n",
               (int)blockIdx.x, tid >> 6, (unsigned long long)(so_t - so_t0), so_acc[0], so_acc[1], so_acc[2], so_acc[3], so_acc[4], so_acc[5], so_acc[6], so_acc[7], so_acc[8], so_acc[9], so_acc[10], so_acc[11]);
#endif
#undef PZ_OSTAMP
}

}  // namespace pz