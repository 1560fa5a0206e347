// device_mid.hpp — fused middle of the GLWE product pipeline for m2 = 256:
//   forward row pass (length-256 DFT over j2)  ->  vector-matrix product with the prepared key
//   ->  inverse row pass (x conj tw12), for CT ciphertexts and one frequency row q1 per workgroup.
// The spectra (VecZnxDft a_dft / res_dft of the reference's op sequence,
// poulpy-core/src/external_product/glwe.rs:231-234,270) never leave the CU: a tile of
// CT x 16 polynomials x 256 points lives in LDS (<= 136 KiB), which removes two of the five
// HBM round trips of the unfused pipeline.
//
//   T   : T'[b*npi + r][q1][j2]   rows of 256 contiguous points (k_fwd_pass1<ROWMAJOR>)
//   T2  : T2'[b*npo + c][q1][j2]  (consumed by k_inv_tail<ROWMAJOR>)
//   P   : P'[q1][r][c][q2]        key permuted so that a row's slice is contiguous along q2
// r = limb_in*cols_in + col_in, c = limb_out*cols_out + col_out (the flat VMP indices).
// Threads: 512 = (row of the tile: 32) x (16 lanes per row); each thread always holds 16 points.
// The key slice of a row (npi*npo*256 points, 1 MiB at 16x16) is shared by every ciphertext tile of
// that row; the 1-D grid is decoded so that those workgroups run back to back on ONE XCD and the
// slice is served by its L2 after the first fetch.
#pragma once
#include "device_fft.hpp"

namespace pz {

#ifndef PZ_MID_RS
#define PZ_MID_RS 144
#endif
constexpr int kMidRS = PZ_MID_RS;
#ifndef PZ_MID_BR_POST
#define PZ_MID_BR_POST 1   // blind-rotation block step on k_mid128 (nested form): monomial factor on the coefficient's sum (0: on every input value, rounds 3 - 4)
#endif
#ifndef PZ_MID_BR_AVPF
#define PZ_MID_BR_AVPF 1   // blind-rotation block step on k_mid128: operands of the next row read from LDS one row ahead (0: read at their use)
#endif
#ifndef PZ_MID_STAMP
#define PZ_MID_STAMP 0   // diagnostic build: per-phase s_memtime totals of k_mid128, printed by a few waves (tools/dbg/mid_stamps.sh)
#endif
#ifndef PZ_MIDR_HALFKEY
#define PZ_MIDR_HALFKEY 0   // timing ablation (results invalid): every second key row of k_mid128r's product is never requested - what the product phase
#endif                      // would cost with eight ciphertexts per key fetch, before any of that scheme's own costs (NOTEBOOK.md 13.2)
#ifndef PZ_MIDR_SADDR
#define PZ_MIDR_SADDR 1     // k_mid128r product: uniform key-row base + 32-bit lane offsets, lane base + scalar row offset in LDS (0: 64-bit lane pointers, A/B)
#endif
#ifndef PZ_MIDR_ILV
#define PZ_MIDR_ILV 1       // k_mid128r: inverse row pass of tile t and forward row pass of tile t + 1 interleaved (0: one after the other, rounds 3 - 4a; NOTEBOOK.md 13.2)
#endif
#ifndef PZ_MIDR_KR
#define PZ_MIDR_KR 6     // key-row slots of k_mid128r (rows requested KR - 1 ahead; build-time for A/B runs)
#endif
   // row stride of the k_mid128 tile (see there; build-time for A/B runs)

struct MidArgs {
    const cplx* T;
    cplx* T2;
    const cplx* P;
    int npi, npo;       // polynomials per ciphertext in / out (<= 16 each)
    int nrows, ncols;   // key matrix: rows*cols_in x cols_out*size
    int row_max;        // min(nrows, npi)
    int ncomp;          // output polynomials that have a key column; the rest are zero
    int batch, m1, n_ct;
    const cplx* wL2;    // exp(2*pi*i*t/256)
    const cplx* tw12t;  // [q1][j2]
    cplx* dummy;        // >= 512*256 points of scratch: where rows without an output polynomial store
    int groups;         // row groups per XCD (see k_mid)
    int stagger, stagger_mod;  // start delay of workgroup w: ((w >> 3) % stagger_mod) * stagger * s_sleep(127)
    // k_mid128<.., PERM = true>: the product is written to spectrum position q_out = perm_mul * q_in + perm_add (mod m), so that
    // the inverse transform of the result is phi(big) for X -> X^p with p = 1 mod 4:  DFT(phi(a))[q] = DFT(a)[p q + (p-1)/4],
    // perm_mul = p^-1, perm_add = -p^-1 (p-1)/4.  Rows map to rows (q1_out depends on q1 only), q2 moves inside the row.
    // Galois elements p = 3 mod 4 (round 3): the evaluation points of the folded transform are the 2N-th roots with exponent 1 mod 4, and
    // p (4q + 1) = 3 mod 4 names the CONJUGATE point of -p (4q + 1) = 1 mod 4, so DFT(phi(a))[q] = conj(DFT(a)[-p q - (p + 1)/4]):
    // the same affine row-to-row map with perm_mul = (-p)^-1, perm_add = (-p)^-1 (p + 1)/4, and the product is conjugated as it is
    // written (perm_ysign = -1.0; +1.0 otherwise).  X -> X^-1, the first step of glwe_trace, is the pure conjugation (identity map).
    unsigned perm_mul, perm_add;
    double perm_ysign;
    int log_m1;
    // k_mid128<.., DS = true> (dsize > 1, poulpy-core external_product/glwe.rs:235-267, keyswitching/glwe.rs:332-379): the limbs of `a`
    // are digits of dsize groups; input polynomial ds_in[t] (its slot in the tile) multiplies key row ds_row[t] shifted by ds_coff[t]
    // columns and only reaches the first ds_cb[t] output polynomials (the reference's per-digit limb_offset and its dropped limbs):
    //   res[c] = sum_t a[ds_in[t]] * P[ds_row[t]][c + ds_coff[t]]   for c < ds_cb[t]
    int ds_n;
    unsigned char ds_in[32], ds_row[32], ds_coff[32], ds_cb[32];
    // k_mid128<.., BR = true> (CGGI block step, poulpy-bin-fhe blind_rotation/algorithms/cggi/algorithm.rs:319-337): P holds the br_blk
    // GGSWs of one LWE block as br_blk * br_rm key rows; term t = (i, r) multiplies input slot r by key row t and by the monomial factor
    //   DFT(X^a_i - 1)[q] = w2n[a_i (4q + 1) mod 2n] - 1,   a_i = br_lwe[b * br_lwe_bs + 1 + br_i0 + i]
    // of its ciphertext (what the reference does with svp_apply_dft_to_dft on x_pow_a[a_i] and two vec_znx_dft additions, :331-335)
    const long long* br_lwe;
    long long br_lwe_bs;
    int br_i0, br_blk, br_rm;
    const cplx* w2n;
    int dbg;   // timing ablation of k_mid128 (POULPY_DBG_MID_SKIP; results invalid): 1 no product FMAs, 2 no key loads, 4 no T' loads,
               // 8 no T2' stores, 16 no row DFTs
};

// Persistent: gridDim.x = 8*W workgroups (one per CU); workgroup (xcd = bid & 7, w = bid >> 3) walks the tiles
// (q1 = 8*k + xcd, ciphertext tile) of "its" XCD with stride W, so the W workgroups of an XCD sweep the
// ciphertext tiles of one frequency row together and share its key slice through that XCD's L2.  The loads
// of the next tile are issued before the inverse row pass of the current one.
template <int CT>
__global__ void __launch_bounds__(CT * 256)
k_mid(MidArgs g) {
    constexpr int M2 = 256;
    constexpr int NC = 16 / CT;  // output polynomials per thread in the product phase (CT*256 threads cover 256 points x 16)
    constexpr int RS = 17 * 16;  // padded row stride (points)
    extern __shared__ cplx lds[];  // CT*16 rows x RS | wL2[256] | tw12t row [256]
    const int tid = threadIdx.x;
    const long long m = (long long)g.m1 * M2;
    const int row = tid >> 4, o = tid & 15;
    const int ctl = row >> 4, rr = row & 15;  // ciphertext within the tile, polynomial slot
    cplx* rowbuf = lds + row * RS;
    // twiddles are read from LDS: a global gather in front of dependent arithmetic costs a full memory latency
    // per use at this occupancy (one workgroup per CU) and its vmcnt wait would also drain the tile stores
    cplx* wl = lds + CT * 16 * RS;
    cplx* twrow = wl + M2;

    // tile enumeration: XCD x owns the rows q1 = 8*k + x; its W workgroups are split into G groups, group gi sweeps
    // the rows k = gi (mod G) with stride W/G over (row, ciphertext tile).  G > 1 keeps several key slices in flight
    // per XCD (still L2 resident) and lowers the number of workgroups hammering the same lines at once.
    const bool xcd_map = (g.m1 & 7) == 0 && (gridDim.x & 7) == 0;
    const int xcd = xcd_map ? (blockIdx.x & 7) : 0;
    const int wx = xcd_map ? (blockIdx.x >> 3) : blockIdx.x;
    const int Wx = xcd_map ? (gridDim.x >> 3) : gridDim.x;
    const int rows_x = xcd_map ? g.m1 / 8 : g.m1;
    const int G = (g.groups > 0 && Wx % g.groups == 0 && rows_x % g.groups == 0) ? g.groups : 1;
    const int W = Wx / G, gi = wx / W, w = wx % W;
    const int ntiles = (rows_x / G) * g.n_ct;
    if (w >= ntiles) return;
    if (tid < M2) wl[tid] = g.wL2[tid];
    __syncthreads();
    // phase stagger (see launch_mid): workgroups start up to (stagger_mod-1) x stagger x 8128 clocks apart
    if (g.stagger > 0) {
        const int k = (blockIdx.x >> 3) % g.stagger_mod;
        for (int i = 0; i < k * g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // Loop shape (software pipeline, one tile per iteration):
    //   prologue : loads(t0); forward row DFT(t0) -> S
    //   iteration: product(t) ; issue loads(t+1) ; inverse row DFT(t) ; 16 stores(t) ; barrier ;
    //              forward row DFT(t+1) -> S
    // Loads and stores are UNCONDITIONAL (absent rows read row 0 of a valid polynomial and write to the
    // scratch rows behind T2), so that exactly 16 stores are younger than the 16 loads when the next forward
    // pass needs them: the compiler can then wait with vmcnt(16) and the stores of tile t drain under the
    // forward pass and the product of tile t+1.
    cplx x[16];
    auto tile_q1 = [&](int L) { const int k = (L / g.n_ct) * G + gi; return xcd_map ? k * 8 + xcd : k; };
    auto src_ptr = [&](int L) {
        const int Lc = min(L, ntiles - 1);
        const int b_ = min((Lc % g.n_ct) * CT + ctl, g.batch - 1);
        const int r_ = min(rr, g.npi - 1);
        return g.T + ((long long)b_ * g.npi + r_) * m + (long long)tile_q1(Lc) * M2 + o;
    };
    auto in_active = [&](int L) { return L < ntiles && (L % g.n_ct) * CT + ctl < g.batch && rr < g.npi; };

#define PZ_MID_FWD(ACTIVE)                                                                        \
    {                                                                                             \
        if (!(ACTIVE)) {                                                                          \
            _Pragma("unroll") for (int n1 = 0; n1 < 16; ++n1) x[n1] = make_double2(0.0, 0.0);     \
        }                                                                                         \
        Bfly<16, false>::run(x);                                                                  \
        _Pragma("unroll") for (int k1 = 0; k1 < 16; ++k1) {                                       \
            cplx v = x[k1];                                                                       \
            if (k1 > 0) v = cmul(v, wl[o * k1]);                                                  \
            rowbuf[k1 * 17 + o] = v;                                                              \
        }                                                                                         \
        row_sync();                                                                               \
        _Pragma("unroll") for (int oo = 0; oo < 16; ++oo) x[oo] = rowbuf[o * 17 + oo];            \
        Bfly<16, false>::run(x);                                                                  \
        row_sync();                                                                               \
        _Pragma("unroll") for (int k2 = 0; k2 < 16; ++k2) rowbuf[o + 16 * k2] = x[k2];            \
        lds_barrier();                                                                            \
    }

    cplx twn = make_double2(0.0, 0.0);  // this thread's entry of the next tile's inter-pass twiddle row
    {
        const cplx* src = src_ptr(w);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) x[n1] = src[16 * n1];
        twn = g.tw12t[(long long)tile_q1(w) * M2 + (tid & 255)];
    }
    if (tid < M2) twrow[tid] = twn;
    // first key row of the coming product: requested before the forward pass so that its L2 latency is hidden
    const int vq2 = tid & 255, vcg = tid >> 8;
    // The workgroups of an XCD sweep the same key slice at the same time; each starts the row loop at a different
    // row so that they do not all hit the same L2 channels at once (sum order differs per workgroup, far inside the
    // rounding margin; results stay deterministic for a given launch geometry)
    const int rot = g.row_max > 0 ? (w % g.row_max) : 0;
    cplx pn[NC];
#define PZ_MID_P0(LT)                                                                                  \
    {                                                                                                  \
        const long long base_ = (long long)tile_q1(min((LT), ntiles - 1)) * g.nrows * g.ncols;         \
        _Pragma("unroll") for (int j = 0; j < NC; ++j)                                                 \
            pn[j] = g.P[(base_ + (long long)rot * g.ncols + min(vcg * NC + j, g.ncomp - 1)) * M2 + vq2];  \
    }
    PZ_MID_P0(w)
    PZ_MID_FWD(in_active(w))

    for (int L = w; L < ntiles; L += W) {
        const int q1 = tile_q1(L);
        const int b = (L % g.n_ct) * CT + ctl;

        // ---------------- product: res[b][c][q] = sum_r a[b][r][q] * P[r][c][q] ----------------
        {
            const int q2 = tid & 255, cg = tid >> 8;  // NC output polynomials per 256-thread slice of the workgroup
            cplx acc[CT][NC];
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[i][j] = make_double2(0.0, 0.0);
            const cplx* pp[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = min(cg * NC + j, g.ncomp - 1);
                pp[j] = g.P + ((long long)q1 * g.nrows * g.ncols + c) * M2 + q2;
            }
            const long long prow = (long long)g.ncols * M2;
            // key row r+1 travels while row r is consumed (row 0 was requested before the forward pass; deeper
            // prefetch spills at 256 VGPRs and did not pay)
            // Two register slots in ping-pong (pn = the row about to be used, pb = the one after): written so that no
            // register copy sits between a load and its use — a copy makes the compiler wait for the load at the end of
            // the iteration that issued it, which exposed the full L2 latency on every row.
            cplx pb[NC];
#define PZ_LOADROW(DST, IT)                                                                     \
    {                                                                                           \
        int r_ = (IT) + rot;                                                                    \
        r_ -= (r_ >= g.row_max) ? g.row_max : 0;                                                \
        r_ -= (r_ >= g.row_max) ? g.row_max : 0;                                                \
        const long long off_ = (long long)r_ * prow;                                            \
        _Pragma("unroll") for (int j = 0; j < NC; ++j) DST[j] = pp[j][off_];                    \
    }
#define PZ_USEROW(SRC, IT)                                                                      \
    {                                                                                           \
        int r_ = (IT) + rot;                                                                    \
        r_ -= (r_ >= g.row_max) ? g.row_max : 0;                                                \
        _Pragma("unroll") for (int i = 0; i < CT; ++i) {                                        \
            const cplx av = lds[(i * 16 + r_) * RS + q2];                                       \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) {                                    \
                acc[i][j].x = __builtin_fma(av.x, SRC[j].x, acc[i][j].x);                       \
                acc[i][j].x = __builtin_fma(-av.y, SRC[j].y, acc[i][j].x);                      \
                acc[i][j].y = __builtin_fma(av.x, SRC[j].y, acc[i][j].y);                       \
                acc[i][j].y = __builtin_fma(av.y, SRC[j].x, acc[i][j].y);                       \
            }                                                                                   \
        }                                                                                       \
    }
            int it = 0;
            for (; it + 1 < g.row_max; it += 2) {
                PZ_LOADROW(pb, it + 1)
                PZ_USEROW(pn, it)
                PZ_LOADROW(pn, it + 2)   // wraps to a valid row at the end; that load is simply unused
                PZ_USEROW(pb, it + 1)
            }
            if (it < g.row_max) PZ_USEROW(pn, it)
#undef PZ_LOADROW
#undef PZ_USEROW
            lds_barrier();  // every a value has been read: the tile can be overwritten with the products
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int c = cg * NC + j;
                    lds[(i * 16 + c) * RS + q2] = (c < g.ncomp) ? acc[i][j] : make_double2(0.0, 0.0);
                }
            lds_barrier();
        }

        // next tile's inputs start travelling now (always 16 loads)
        {
            const cplx* src = src_ptr(L + W);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) x[n1] = src[16 * n1];
            twn = g.tw12t[(long long)tile_q1(min(L + W, ntiles - 1)) * M2 + (tid & 255)];
        }

        // ---------------- inverse row DFT of the output polynomials, always 16 stores ----------------
        {
            cplx u[16];
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) u[k2] = rowbuf[o + 16 * k2];  // k1 = o
            Bfly<16, true>::run(u);
            row_sync();
#pragma unroll
            for (int oo = 0; oo < 16; ++oo) {
                cplx v = u[oo];
                if (o > 0 && oo > 0) v = cmulc(v, wl[oo * o]);
                rowbuf[o * 17 + oo] = v;  // z[k1 = o][oo]
            }
            row_sync();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 17 + o];
            Bfly<16, true>::run(u);
            const bool active = b < g.batch && rr < g.npo;
            cplx* dst = active ? g.T2 + ((long long)b * g.npo + rr) * m + (long long)q1 * M2 + o
                               : g.dummy + ((long long)blockIdx.x * (CT * 16) + row) * M2 + o;   // this workgroup's own scratch rows (see k_mid128)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) dst[16 * n1] = cmulc(u[n1], twrow[o + 16 * n1]);
        }
        lds_barrier();  // the tile is rewritten by the forward pass below
        if (tid < M2) twrow[tid] = twn;  // read again only after the barriers of the next product phase
        PZ_MID_P0(L + W)
        PZ_MID_FWD(in_active(L + W))
    }
#undef PZ_MID_FWD
#undef PZ_MID_P0
}

// =================================================================================
// Same fused middle for plans with m2 = 128 (m1 = 256): rows of 128 points are owned by 8 lanes
// (16 points each: one radix-16 butterfly, then two radix-8 butterflies), so a tile of FOUR ciphertexts
// x 16 polynomials fits in LDS (144 KiB) and every key value fetched serves four ciphertexts instead of
// two.  512 threads = 64 rows x 8 lanes; product phase: 128 points x 4 groups of 4 outputs.
// Measured (round 1): middle kernel 16-19 % faster than the m2 = 256 form (the key slice is streamed from L2 half as
// often); default at N = 2^16 since the radix 16 x 16 tail of the m1 = 256 column passes no longer spills
// (device_fft.hpp, SPLIT).  POULPY_DBG_SPLIT=t selects the 128 x 256 split (k_mid<2>) instead.  (Two ciphertexts per tile
// with two workgroups per CU was no faster: NOTEBOOK.md 4.)
// =================================================================================
// NP = polynomial slots per ciphertext: 16; 8 for shapes with <= 8 polynomials in and out such as rank 1 with 4 limbs (twice the
// ciphertexts per tile and per key fetch); 32 for rank 2-3 or 16 limbs (two ciphertexts per tile).  The 4 thread groups of the product phase split into GC column groups x GT
// ciphertext groups so that a thread always owns 16 accumulators (4 ciphertexts x 4 outputs, or 2 x 8).
// NCO (round 4): outputs per thread when the tile's 8 slots carry SIX output polynomials (blind rotation at rank 1 with 3 key limbs): 3, so that
// the two column groups cover exactly the 6 columns - with the default 4 a quarter of the product's multiply-adds and key loads went to two
// columns that do not exist (clamped loads, results discarded): product phase 48 k of the tile's 70 k cycles at N = 2^14 (r04_br_probe.txt).
template <int CT, int NP = 16, bool PERM = false, bool DS = false, bool BR = false, bool SKIPW = false, int BRNEST = 0, int NCO = 0>
__global__ void __launch_bounds__(CT * NP * 8)
k_mid128(MidArgs g) {
    static_assert(!SKIPW || (NP > 8 && !BR && !DS), "wave skipping: 16- and 32-slot tiles of the plain product only");
    constexpr int M2 = 128;
    constexpr int NT = CT * NP * 8;
    constexpr int NCG = NT / M2;       // thread groups in the product phase
    // outputs per thread (NP = 32: two ciphertexts x 8 outputs, as in k_mid<2>).  8 for the 16-slot tile too (2 ciphertexts x 8 outputs per
    // thread: half the operand reads from LDS, every key value fetched by two threads) was measured slower: 5.08 -> 5.39 ms
    constexpr int NC = NCO ? NCO : (NP == 32 ? 8 : 4);
    constexpr int GC = NCO == 3 ? 2 : NP / NC;   // column groups (NCO = 3: 2 x 3 columns of an 8-slot tile)
    constexpr int GT = NCG / GC;       // ciphertext groups
    constexpr int CTt = CT / GT;       // ciphertexts per thread
    static_assert(GC * GT == NCG && CTt * GT == CT && NT == 512, "k_mid128 tile shape");
    // padded row stride (points): z[k1][o] at k1*9 + o needs 143.  With 144 (= 0 mod 16) the rows alias in the 16-lane groups of
    // ds_read_b128 and three of the seven read passes of a tile are 2-way bank conflicts (SQ_LDS_BANK_CONFLICT = 16 % of the LDS cycles);
    // 152 (= 8 mod 16, -DPZ_MID_RS=152) makes them conflict-free and was measured identical (51.2 vs 51.2 ms per 10 launches): those
    // passes are not on the critical path.  144 it stays (16 KiB of LDS less).
    constexpr int RS = kMidRS;
    extern __shared__ cplx lds[];      // CT*16 rows x RS | wL2[128] | tw12t row [128]
    const int tid0 = threadIdx.x;
    // ablation mask: compile-time zero (PZ_DBG) except in the BR variant, whose register allocation is better WITH the run-time tests
    // (without them: 36 bytes of scratch and N = 2^14 blind rotation 6 240 -> 5 715/s, round 3)
    const int dbgv = (BR && BRNEST == 0) ? g.dbg : PZ_DBG(g.dbg);
    const long long m = (long long)g.m1 * M2;
    cplx* wl = lds + CT * NP * RS;
    // Lane coordinates are re-derived from an OPAQUE copy of the thread index at the top of every phase (round 3).  Derived once, the
    // compiler hoists every address that depends on them out of the tile loop - ~60 loop-invariant registers (LDS offsets of the
    // exchange passes, twiddle addresses, row pointers) that sit beside the 64 accumulators and the 64 prefetched T' values and
    // spill; recomputing them costs a few integer instructions per phase.
#define PZ_MID_LANE                                                                               \
    const int tid = pz_opaque(tid0);                                                              \
    const int row = tid >> 3, o = tid & 7;                                                        \
    const int ctl = row / NP, rr = row % NP;                                                      \
    cplx* const rowbuf = lds + row * RS;                                                          \
    const int vq2 = tid & (M2 - 1), vcg = (tid / M2) % GC, vtg = (tid / M2) / GC;                 \
    (void)ctl; (void)rr; (void)rowbuf; (void)vq2; (void)vcg; (void)vtg; (void)o;
    cplx* twrow = wl + M2;
    unsigned* abuf = reinterpret_cast<unsigned*>(twrow + M2);   // BR: a_i mod 2n of the tile's ciphertexts, [ct][16]

    const bool xcd_map = (g.m1 & 7) == 0 && (gridDim.x & 7) == 0;
    const int xcd = xcd_map ? (blockIdx.x & 7) : 0;
    const int w = xcd_map ? (blockIdx.x >> 3) : blockIdx.x;
    const int W = xcd_map ? (gridDim.x >> 3) : gridDim.x;
    const int rows_x = xcd_map ? g.m1 / 8 : g.m1;
    const int ntiles = rows_x * g.n_ct;
    if (w >= ntiles) return;
    if (tid0 < M2) wl[tid0] = g.wL2[tid0];
    __syncthreads();
    if (g.stagger > 0 && g.stagger_mod > 0) {
        const int k = (blockIdx.x / 256) % g.stagger_mod;
        for (int i = 0; i < k * g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    cplx x[16];
    auto tile_q1 = [&](int L) { const int k = L / g.n_ct; return xcd_map ? k * 8 + xcd : k; };
    auto out_q1 = [&](int q1_) { return PERM ? (int)((g.perm_mul * (unsigned)q1_ + g.perm_add) & (unsigned)(g.m1 - 1)) : q1_; };
    auto src_ptr = [&](int L, int ctl, int rr, int o) {
        const int Lc = min(L, ntiles - 1);
        const int b_ = min((Lc % g.n_ct) * CT + ctl, g.batch - 1);
        const int r_ = min(rr, g.npi - 1);
        return g.T + ((long long)b_ * g.npi + r_) * m + (long long)tile_q1(Lc) * M2 + o;
    };
    auto in_active = [&](int L, int ctl, int rr) { return L < ntiles && (L % g.n_ct) * CT + ctl < g.batch && rr < g.npi; };
    // A wave owns 8 consecutive polynomial slots of one ciphertext.  Waves whose slots all lie beyond the input (resp. output)
    // polynomials skip the forward (resp. inverse) row pass, its loads and its stores altogether — e.g. the upper half of every
    // ciphertext's 16 slots in a key switch (8 polynomials in): the product never reads those rows.  (Wave-uniform branches.)
    // (only where a ciphertext has more than 8 slots, and not in the BR / DS variants, whose register allocation the extra branches push
    //  over the 256-VGPR cap)
    // SKIPW is a template parameter (round 3): the branches around the tile's loads and stores make the compiler's s_waitcnt insertion
    // assume the worst case at every join — in the 16-polynomial external product, which never skips a wave, the forward pass's
    // prefetched T' loads were waited for with vmcnt(1) at the top of the inverse pass instead of after it.  launch_mid picks the
    // variant with the branches only for shapes that have idle waves (npi or npo <= NP - 8).
    const int rr0 = (tid0 >> 3) % NP;
    const bool wave_in = !SKIPW || (rr0 & ~7) < g.npi, wave_out = !SKIPW || (rr0 & ~7) < g.npo;

    // forward row DFT: x[n1] = row[o + 8*n1] -> radix 16 over n1 (k1) -> x W128^(o*k1) -> z[k1][o];
    // then this lane takes k1 = o and o+8: radix 8 over o -> S[row][q2 = k1 + 16*k2]
#define PZ_MID_FWD(LT)                                                                            \
    {                                                                                             \
      if (wave_in) {                                                                              \
        PZ_MID_LANE                                                                               \
        if (!in_active((LT), ctl, rr)) {                                                                        \
            _Pragma("unroll") for (int n1 = 0; n1 < 16; ++n1) x[n1] = make_double2(0.0, 0.0);     \
        }                                                                                         \
        if (!(dbgv & 16)) Bfly<16, false>::run(x);                                               \
        _Pragma("unroll") for (int k1 = 0; k1 < 16; ++k1) {                                       \
            cplx v = x[k1];                                                                       \
            if (k1 > 0) v = cmul(v, wl[o * k1]);                                                  \
            rowbuf[k1 * 9 + o] = v;                                                               \
        }                                                                                         \
        row_sync();                                                                               \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                           \
            _Pragma("unroll") for (int oo = 0; oo < 8; ++oo) x[8 * h + oo] = rowbuf[(o + 8 * h) * 9 + oo]; \
        }                                                                                         \
        if (!(dbgv & 16)) { Bfly<8, false>::run(x); Bfly<8, false>::run(x + 8); }                \
        row_sync();                                                                               \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                           \
            _Pragma("unroll") for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 8 * h + 16 * k2] = x[8 * h + k2]; \
        }                                                                                         \
      }                                                                                           \
        lds_barrier();                                                                            \
    }

    cplx twn = make_double2(0.0, 0.0);
    // the T' rows and the inter-pass twiddle of tile LT start travelling (always 16 + 1 loads)
#define PZ_MID_XLOAD(LT)                                                                               \
    {                                                                                                  \
        PZ_MID_LANE                                                                                    \
        const cplx* src_ = src_ptr((LT), ctl, rr, o);                                                  \
        if (wave_in) {                                                                                 \
            _Pragma("unroll") for (int n1 = 0; n1 < 16; ++n1)                                          \
                x[n1] = (dbgv & 4) ? make_double2(1.0, (double)n1) : ld_stream(src_ + 8 * n1); \
        }                                                                                              \
        twn = g.tw12t[(long long)out_q1(tile_q1(min((LT), ntiles - 1))) * M2 + (tid & (M2 - 1))];      \
    }
    PZ_MID_XLOAD(w)
    if (tid0 < M2) twrow[tid0] = twn;
#if PZ_MID_STAMP
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_t;
    int st_tiles = 0;
#define PZ_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define PZ_STAMP(i)
#endif
    const int nrow = DS ? g.ds_n : g.row_max;          // product terms per output (BR: br_blk * br_rm, in order)
    if constexpr (!BR) __builtin_assume(nrow >= 1);                       // (launch_mid checks it) no path around the product loop: see SKIPW
    const int rot = (!BR && nrow > 0) ? (w % nrow) : 0;
    const unsigned w2n_mask = 4u * (unsigned)(g.m1 * M2) - 1u;   // 2n - 1
    // BR: the tile's exponents to LDS (read by the product phase after the forward pass's barrier)
#define PZ_MID_ABUF(LT)                                                                              \
    if constexpr (BR) {                                                                                \
        const int tid = tid0;                                                                          \
        if (tid < CT * 16) {                                                                           \
            const int ct_ = tid >> 4, i_ = tid & 15;                                                   \
            const int Lc_ = min((LT), ntiles - 1);                                                     \
            const int b_ = (Lc_ % g.n_ct) * CT + ct_;                                                  \
            unsigned a_ = 0u;                                                                          \
            if (b_ < g.batch && i_ < g.br_blk)                                                         \
                a_ = (unsigned)((unsigned long long)g.br_lwe[(long long)b_ * g.br_lwe_bs + 1 + g.br_i0 + i_] & (unsigned long long)w2n_mask); \
            abuf[tid] = a_;                                                                            \
        }                                                                                              \
    }
    cplx pn[NC], pb[NC];   // key-row slots in ping-pong
#define PZ_MID_P0(LT)                                                                                  \
    {                                                                                                  \
        PZ_MID_LANE                                                                                    \
        const long long base_ = (long long)tile_q1(min((LT), ntiles - 1)) * g.nrows * g.ncols;         \
        if (!(dbgv & 2)) { _Pragma("unroll") for (int j = 0; j < NC; ++j) {                           \
            const int c_ = DS ? min(vcg * NC + j, max((int)g.ds_cb[rot], 1) - 1) + (int)g.ds_coff[rot] : min(vcg * NC + j, g.ncomp - 1); \
            pn[j] = g.P[(base_ + (long long)(DS ? (int)g.ds_row[rot] : rot) * g.ncols + c_) * M2 + vq2]; } } \
    }
    PZ_MID_P0(w)
    PZ_MID_ABUF(w)
    PZ_MID_FWD(w)
    PZ_STAMP(7)

    for (int L = w; L < ntiles; L += W) {
        const int q1 = tile_q1(L);
        {
            PZ_MID_LANE
            const int q2 = vq2, cg = vcg;
            // BR: monomial factors of the current (f) and the next (fn, raw table values) block coefficient for this thread's ciphertexts
            const unsigned tq = 4u * ((unsigned)q1 + ((unsigned)q2 << g.log_m1)) + 1u;
            cplx f[BR ? CTt : 1], fn[BR ? CTt : 1], fl[BR ? CTt : 1];   // fl: the factor after next, in flight (see PZ_BR_USE)
            int br_slot = 0, br_i = 0;
#define PZ_MID_LOADF(DST, II)                                                                   \
    {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < CTt; ++i)                                         \
            DST[i] = g.w2n[(abuf[(vtg * CTt + i) * 16 + min((II), 15)] * tq) & w2n_mask];       \
    }
            if constexpr (BR) {
                PZ_MID_LOADF(f, 0)
                PZ_MID_LOADF(fn, 1)
                if constexpr (BRNEST != 0) PZ_MID_LOADF(fl, 2)
#pragma unroll
                for (int i = 0; i < CTt; ++i) f[i].x -= 1.0;
            }
            cplx acc[CTt][NC];
#pragma unroll
            for (int i = 0; i < CTt; ++i)
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[i][j] = make_double2(0.0, 0.0);
            const cplx* pp[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = DS ? 0 : min(cg * NC + j, g.ncomp - 1);   // DS: the column is part of the per-row offset
                pp[j] = g.P + ((long long)q1 * g.nrows * g.ncols + c) * M2 + q2;
            }
            const long long prow = (long long)g.ncols * M2;
#define PZ_LOADROW(DST, IT)                                                                     \
    {                                                                                           \
        int r_ = (IT) + rot;                                                                    \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                          \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                          \
        if (DS) {                                                                               \
            const int cb_ = max((int)g.ds_cb[r_], 1) - 1, co_ = (int)g.ds_coff[r_];             \
            const long long ro_ = (long long)g.ds_row[r_] * prow;                               \
            if (!(dbgv & 2)) { _Pragma("unroll") for (int j = 0; j < NC; ++j)                  \
                DST[j] = pp[j][ro_ + (long long)(min(cg * NC + j, cb_) + co_) * M2]; }          \
        } else {                                                                                \
            const long long off_ = (long long)r_ * prow;                                        \
            if (!(dbgv & 2)) { _Pragma("unroll") for (int j = 0; j < NC; ++j) DST[j] = pp[j][off_]; }  \
        }                                                                                       \
        if constexpr (BR) __builtin_amdgcn_sched_barrier(0);                                    \
    }
#define PZ_USEROW(SRC, IT)                                                                      \
    {                                                                                           \
        int r_ = (IT) + rot;                                                                    \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                          \
        if (DS) {                                                                               \
            const int cbv_ = (int)g.ds_cb[r_];                                                  \
            _Pragma("unroll") for (int j = 0; j < NC; ++j)                                      \
                if (cg * NC + j >= cbv_) SRC[j] = make_double2(0.0, 0.0);                       \
            r_ = (int)g.ds_in[r_];                                                              \
        }                                                                                       \
        if constexpr (BR) r_ = br_slot;                                                         \
        if (!(dbgv & 1)) _Pragma("unroll") for (int i = 0; i < CTt; ++i) {                      \
            cplx av = lds[((vtg * CTt + i) * NP + r_) * RS + q2];                               \
            if constexpr (BR) av = cmul(av, f[i]);                                              \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) {                                    \
                acc[i][j].x = __builtin_fma(av.x, SRC[j].x, acc[i][j].x);                       \
                acc[i][j].x = __builtin_fma(-av.y, SRC[j].y, acc[i][j].x);                      \
                acc[i][j].y = __builtin_fma(av.x, SRC[j].y, acc[i][j].y);                       \
                acc[i][j].y = __builtin_fma(av.y, SRC[j].x, acc[i][j].y);                       \
            }                                                                                   \
        }                                                                                       \
        if constexpr (BR) {                                                                     \
            if (++br_slot == g.br_rm) {        /* next block coefficient: fn becomes f, the one after starts travelling */ \
                br_slot = 0;                                                                    \
                ++br_i;                                                                         \
                _Pragma("unroll") for (int i = 0; i < CTt; ++i) f[i] = make_double2(fn[i].x - 1.0, fn[i].y); \
                PZ_MID_LOADF(fn, br_i + 1)                                                      \
            }                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                  \
        }                                                                                       \
    }
            int it = 0;
            // (BR: deeper key prefetch — rings of three / four row slots — spills at the 256-VGPR cap and measured slower: 30.1 vs 26.8 ms)
            if constexpr (NC == 8) {
                // 8 key values per thread and row: the second register slot of the ping-pong is what pushes this shape over the 256-VGPR
                // cap (132-164 bytes of scratch, 57.9 -> 50.3 ms per 10 launches at 16 limbs without it); one slot, the next row requested
                // right after the current one has been consumed
                for (; it + 1 < nrow; ++it) {
                    PZ_USEROW(pn, it)
                    PZ_LOADROW(pn, it + 1)
                }
                PZ_USEROW(pn, it)
            } else if constexpr (BR) {
                // (round 3, late) the stamps show this phase at 62 k of the tile's 86 k cycles at N = 2^14: 56 rows of 84 floating-point
                // instructions each.  What did NOT help on the flat row loop: a ring of four key-row slots (27.8 -> 28.0 ms per 82 blocks),
                // the operands read from LDS one row ahead (27.6).  What the ISA showed: 115 VALU + 50 SALU instructions and six branches
                // per row (run-time ablation tests, the row-index wrap with a 64-bit multiply, the factor rotation as a branch whose
                // register homes made the allocator wait for the fresh factor load at once).  BRNEST (an even number of rows per
                // coefficient): coefficient loop around a branch-free row-pair loop, running key offsets, factors rotated in
                // straight-line code and requested two coefficients ahead, compile-time ablation mask: 87 VALU + 9 SALU per row, no
                // branch - 28.35 -> 25.25 ms (N = 2^14: 5 530 -> 5 930 rotations/s, N = 4096: 25 300 -> 27 300).  Four key slots on top: +1 %, not kept.
#if PZ_MID_BR_AVPF
                cplx avA[CTt], avB[CTt];
            // (round 4, as in k_mid128r's product: lane base + scalar row offset for the operands, uniform key-row base + 32-bit lane offsets
            //  for the key - no 64-bit vector address arithmetic beside the row's 64 floating-point instructions)
            const cplx* const avp = lds + vtg * CTt * NP * RS + q2;
            const char* const kbase = (const char*)(g.P + (long long)q1 * g.nrows * g.ncols * M2);
            unsigned koff[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) koff[j] = (unsigned)((DS ? 0 : min(cg * NC + j, g.ncomp - 1)) * M2 + q2) * 16u;
#define PZ_BR_AV(DST, SLOT_)                                                                     \
    {                                                                                            \
        const cplx* ar_ = avp + (SLOT_) * RS;                                                    \
        _Pragma("unroll") for (int i = 0; i < CTt; ++i) DST[i] = ar_[i * NP * RS];               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
            // PZ_MID_BR_POST (round 5, nested form): the monomial factor d = DFT(X^a) - 1 multiplies the coefficient's SUM over its key rows (NC
            // complex FMAs per ciphertext and coefficient) instead of every input value in front of its row's FMAs (br_rm complex products)
            constexpr bool POSTF = PZ_MID_BR_POST && BRNEST != 0 && NCO == 3;   // (the 4-column forms spill with the extra sums: 96 - 100 B)
            cplx sacc[POSTF ? CTt : 1][POSTF ? NC : 1];
#define PZ_BR_FMA(SRC, AV)                                                                       \
    {                                                                                            \
        if (!(dbgv & 1)) _Pragma("unroll") for (int i = 0; i < CTt; ++i) {                       \
            const cplx av = POSTF ? AV[i] : cmul(AV[i], f[i]);                                   \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) {                                     \
                cplx& d_ = POSTF ? sacc[POSTF ? i : 0][POSTF ? j : 0] : acc[i][j];               \
                d_.x = __builtin_fma(av.x, SRC[j].x, d_.x);                                      \
                d_.x = __builtin_fma(-av.y, SRC[j].y, d_.x);                                     \
                d_.y = __builtin_fma(av.x, SRC[j].y, d_.y);                                      \
                d_.y = __builtin_fma(av.y, SRC[j].x, d_.y);                                      \
            }                                                                                    \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
                if constexpr (BRNEST != 0) {   // launch_mid: g.br_rm even
                    // coefficient by coefficient (an even number of rows each, so that the two key slots keep their parity): the factors
                    // rotate in straight-line code at the top of a coefficient - current <- next (requested two coefficients ago) <- the one in
                    // flight, which is requested here.  As a branch inside the flat row loop (below, odd row counts) the rotation made the
                    // register allocator copy the in-flight factor behind vmcnt(0) on EVERY row (round 3 ISA).
                    PZ_BR_AV(avA, 0)
                    long long ko = 0;   // offset of key row `it`
#define PZ_BR_KROW(DST, OFF_)                                                                    \
    {                                                                                            \
        const char* rp_ = kbase + (OFF_) * 16;                                                   \
        if (!(dbgv & 2)) { _Pragma("unroll") for (int j = 0; j < NC; ++j) DST[j] = *(const cplx*)(rp_ + koff[j]); }  \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
                    for (int ci = 0; ci < g.br_blk; ++ci) {
                        if (ci > 0) {
                            if constexpr (POSTF) {   // the finished coefficient's sum enters the accumulators with its factor
#pragma unroll
                                for (int i = 0; i < CTt; ++i)
#pragma unroll
                                    for (int j = 0; j < NC; ++j) {
                                        const cplx sv = sacc[POSTF ? i : 0][POSTF ? j : 0];
                                        acc[i][j].x = __builtin_fma(f[i].x, sv.x, acc[i][j].x);
                                        acc[i][j].x = __builtin_fma(-f[i].y, sv.y, acc[i][j].x);
                                        acc[i][j].y = __builtin_fma(f[i].x, sv.y, acc[i][j].y);
                                        acc[i][j].y = __builtin_fma(f[i].y, sv.x, acc[i][j].y);
                                    }
                            }
#pragma unroll
                            for (int i = 0; i < CTt; ++i) { f[i] = fn[i]; f[i].x -= 1.0; fn[i] = fl[i]; }
                            PZ_MID_LOADF(fl, ci + 2)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (POSTF) {
#pragma unroll
                            for (int i = 0; i < CTt; ++i)
#pragma unroll
                                for (int j = 0; j < NC; ++j) sacc[POSTF ? i : 0][POSTF ? j : 0] = make_double2(0.0, 0.0);
                        }
                        for (int sl = 0; sl < g.br_rm; sl += 2, it += 2, ko += 2 * prow) {
                            // key rows at running offsets (rows come in order here: no wrap arithmetic, no 64-bit multiply per row); the
                            // request past the last row re-reads it and is unused
                            PZ_BR_KROW(pb, ko + prow)
                            PZ_BR_AV(avB, sl + 1)
                            PZ_BR_FMA(pn, avA)
                            PZ_BR_KROW(pn, (it + 2 < nrow) ? ko + 2 * prow : ko + prow)
                            PZ_BR_AV(avA, (sl + 2 == g.br_rm) ? 0 : sl + 2)
                            PZ_BR_FMA(pb, avB)
                        }
                    }
                    if constexpr (POSTF) {   // the last coefficient's sum
#pragma unroll
                        for (int i = 0; i < CTt; ++i)
#pragma unroll
                            for (int j = 0; j < NC; ++j) {
                                const cplx sv = sacc[POSTF ? i : 0][POSTF ? j : 0];
                                acc[i][j].x = __builtin_fma(f[i].x, sv.x, acc[i][j].x);
                                acc[i][j].x = __builtin_fma(-f[i].y, sv.y, acc[i][j].x);
                                acc[i][j].y = __builtin_fma(f[i].x, sv.y, acc[i][j].y);
                                acc[i][j].y = __builtin_fma(f[i].y, sv.x, acc[i][j].y);
                            }
                    }
                } else {
                    // an odd number of rows per coefficient: the flat row loop of rounds 1 - 3 (two slots in ping-pong; the request past the
                    // end wraps to a valid row and is unused)
                    for (; it + 1 < nrow; it += 2) {
                        PZ_LOADROW(pb, it + 1)
                        PZ_USEROW(pn, it)
                        PZ_LOADROW(pn, it + 2)
                        PZ_USEROW(pb, it + 1)
                    }
                    if (it < nrow) PZ_USEROW(pn, it)
                }
#undef PZ_BR_FMA
#undef PZ_BR_AV
#undef PZ_BR_KROW
#else
                // two slots in ping-pong; the request past the end wraps to a valid row and is unused (with the peeled loop below this
                // variant spills)
                for (; it + 1 < nrow; it += 2) {
                    PZ_LOADROW(pb, it + 1)
                    PZ_USEROW(pn, it)
                    PZ_LOADROW(pn, it + 2)
                    PZ_USEROW(pb, it + 1)
                }
                if (it < nrow) PZ_USEROW(pn, it)
#endif
            } else {
                // two slots in ping-pong; no row is requested past the end
                for (; it + 3 < nrow; it += 2) {
                    PZ_LOADROW(pb, it + 1)
                    PZ_USEROW(pn, it)
                    PZ_LOADROW(pn, it + 2)
                    PZ_USEROW(pb, it + 1)
                }
                const int left = nrow - it;   // 1..3 rows
                if (left == 3) {
                    PZ_LOADROW(pb, it + 1)
                    PZ_USEROW(pn, it)
                    PZ_LOADROW(pn, it + 2)
                    PZ_USEROW(pb, it + 1)
                    PZ_USEROW(pn, it + 2)
                } else if (left == 2) {
                    PZ_LOADROW(pb, it + 1)
                    PZ_USEROW(pn, it)
                    PZ_USEROW(pb, it + 1)
                } else {
                    PZ_USEROW(pn, it)
                }
            }
#undef PZ_LOADROW
#undef PZ_USEROW
#undef PZ_MID_LOADF
            PZ_STAMP(0)
            lds_barrier();
            PZ_STAMP(1)
#pragma unroll
            for (int i = 0; i < CTt; ++i) {
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int c = cg * NC + j;
                    const int q2o = PERM ? (int)((((g.perm_mul * (unsigned)q1 + g.perm_add) >> g.log_m1) + g.perm_mul * (unsigned)q2) & (unsigned)(M2 - 1)) : q2;
                    lds[((vtg * CTt + i) * NP + c) * RS + q2o] = (c < g.ncomp) ? acc[i][j] : make_double2(0.0, 0.0);
                }
            }
            lds_barrier();
            PZ_STAMP(2)
        }
        PZ_MID_XLOAD(L + W)
        PZ_STAMP(8)
        // inverse row DFT: this lane owns k1 = o and o+8: radix 8 over k2 -> z[k1][oo] x conj W128^(oo*k1);
        // then lane o gathers z[k1][o] over k1: radix 16 -> row[o + 8*n1]
        if (wave_out) {
            PZ_MID_LANE
            const int b = (L % g.n_ct) * CT + ctl;
            cplx u[16];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) u[8 * h + k2] = rowbuf[o + 8 * h + 16 * k2];
            if constexpr (PERM) {   // Galois elements 3 mod 4: the conjugate of the permuted product (MidArgs::perm_ysign; +1.0 otherwise)
#pragma unroll
                for (int t = 0; t < 16; ++t) u[t].y *= g.perm_ysign;
            }
            if (!(dbgv & 16)) { Bfly<8, true>::run(u); Bfly<8, true>::run(u + 8); }
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
            if (!(dbgv & 16)) Bfly<16, true>::run(u);
            const bool active = b < g.batch && rr < g.npo;
            // rows without an output polynomial store to a scratch row of their OWN workgroup (one 2 KiB row per tile row): with a
            // shared scratch every workgroup of the chip wrote the same lines, which cost more than the real stores (measured on the
            // blind-rotation block step, 6 of 8 slots active: middle kernel 37.8 -> 28 ms per 82 blocks)
            cplx* dst = active ? g.T2 + ((long long)b * g.npo + rr) * m + (long long)out_q1(q1) * M2 + o
                               : g.dummy + ((long long)blockIdx.x * (NT / 8) + row) * M2 + o;
#if PZ_MID_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            PZ_STAMP(9)
#endif
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) if (!(dbgv & 8) || u[n1].x == 1.2345e300) st_stream(dst + 8 * n1, cmulc(u[n1], twrow[o + 8 * n1]));
        }
        PZ_STAMP(3)
        lds_barrier();
        PZ_STAMP(4)
        if (tid0 < M2) twrow[tid0] = twn;
        PZ_MID_P0(L + W)
        PZ_MID_ABUF(L + W)
#if PZ_MID_STAMP
        PZ_STAMP(5)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the prefetched T' rows (the 4 key loads of P0 are younger)
        PZ_STAMP(6)
        ++st_tiles;
#endif
        PZ_MID_FWD(L + W)
        PZ_STAMP(7)
    }
#if PZ_MID_STAMP
    if ((tid0 & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 9 || blockIdx.x == 130 || blockIdx.x == 255))
        printf("STAMP wg %d wave %d tiles %d total %llu | product %llu bar1 %llu accwr %llu xload %llu invc %llu stores %llu bar4 %llu p0 %llu xwait %llu fwd %llu\n",
               (int)blockIdx.x, tid0 >> 6, st_tiles, (unsigned long long)(st_t - st_t0), st_acc[0], st_acc[1], st_acc[2], st_acc[8], st_acc[9], st_acc[3], st_acc[4],
               st_acc[5], st_acc[6], st_acc[7]);
#endif
#undef PZ_STAMP
#undef PZ_MID_FWD
#undef PZ_MID_P0
#undef PZ_MID_ABUF
#undef PZ_MID_XLOAD
#undef PZ_MID_LANE
}

// =================================================================================
// k_mid128r (round 3): the plain product of k_mid128 (no digits, no blind-rotation factors, no idle waves, a multiple of 4 product
// rows) with the tile's GLOBAL-MEMORY instructions spread through the LDS / VALU phases.  Why — s_memtime stamps of k_mid128
// (profiles/r03_mid_stamps_*.txt, cycles per tile of 4 ciphertexts, 34.4 k in all): the product phase is bound by the key stream
// through the CU's vector-memory path (512 KiB per tile at the ~50 B/clk an L2-served stream reaches: 10.7 k), and ISSUING the next
// tile's 16 + 1 loads took a wave 1.9 - 4.7 k cycles, its 16 stores 1.3 - 1.7 k, the 12 key loads in front of the forward pass 1.1 - 2 k:
// a vector-memory instruction blocks its wave until the path accepts it, all eight waves reach the same burst together, and
// meanwhile LDS and VALU idle — as the memory path idles during the 15 k cycles of butterflies and LDS exchanges.  Here
//   * the next tile's T' loads go out in four groups between the steps of the inverse row pass,
//   * the tile's 16 stores are held back (64 registers) and go out in four groups between the steps of the NEXT forward row pass,
//   * the first key rows of the next product are requested between the last steps of that forward pass, KR - 1 of them,
//   * the inter-pass twiddle row is double-buffered, so no workgroup barrier separates the inverse pass from the forward pass
//     (both only touch the wave's own rows) and the waves drift apart instead of meeting the same resource at the same time.
// The arithmetic, its order and therefore every output bit are those of k_mid128.
// =================================================================================
// IN = false: the code of a wave whose 8 polynomial slots carry no input (HALFIN shapes, e.g. the 8 input polynomials of a key switch in
// a 16-slot tile): no T' loads, no forward pass; it still takes part in the product, runs the inverse pass of its 8 output rows and
// issues its share of the stores and key requests in the same order.  Both codes execute the same barriers.  They are two
// instantiations of this function rather than branches inside one loop because a branch around the loads makes the compiler's
// s_waitcnt insertion assume the worst case at every join (k_mid128, SKIPW).
// C2 (round 6, the 16-limb key switch family: 16 polynomials in, 32 = 16 limbs x 2 columns out): the tile of the 32-slot form holds 2 ciphertexts, so
// every key value a thread loads serves 2 ciphertexts and the product phase is paced by the L2 -> CU key stream (1 MiB per tile at ~50 B/clk:
// 21 k of the tile's 36 k cycles, profiles/r05_roofline.md "l2_stream").  Here the tile is 4 ciphertexts x 16 slots and the two output COLUMNS are
// two consecutive virtual tiles V = 2 k + col of the same inputs: slot j of pass `col` is key column / output polynomial 2 j + col.  Each key
// value serves 4 ciphertexts (half the key stream per ciphertext); the price is a second load (L2: the rows were read a tile ago) and forward
// row transform of the 16 input rows.
template <int CT, int NP, bool PERM, int NR, int KR, bool HALFIN, bool IN, bool DS = false, bool C2 = false>
__device__ __forceinline__ void mid128r_body(const MidArgs& g) {
    static_assert(!C2 || (!DS && !HALFIN && NP == 16), "two-column passes: the plain 16-slot tile");
    constexpr int M2 = 128, NT = CT * NP * 8;   // 512 threads (64 rows), or 256 (32 rows: two workgroups per CU, round-4 experiment)
    constexpr int NC = (NP * M2 / NT) > 4 ? (NP * M2 / NT) : 4;   // outputs per thread (32-slot tile: two ciphertexts x 8 outputs, as in k_mid128)
    constexpr int GC = NP / NC, GT = (NT / M2) / GC, CTt = CT / GT;
    static_assert((NT == 512 || NT == 256) && GC * GT == NT / M2 && CTt * GT == CT && KR >= 3 && KR <= 7 && KR <= NR, "k_mid128r tile shape");
    constexpr int RS = kMidRS;
    extern __shared__ cplx lds[];      // CT*NP rows x RS | wL2[128] | tw12t rows [2][128]
    const int tid0 = threadIdx.x;
    const long long m = (long long)g.m1 * M2;
    cplx* wl = lds + CT * NP * RS;
    cplx* twrow2 = wl + M2;
#define PZ_MID_LANE                                                                               \
    const int tid = pz_opaque(tid0);                                                              \
    const int row = tid >> 3, o = tid & 7;                                                        \
    const int ctl = row / NP, rr = row % NP;                                                      \
    cplx* const rowbuf = lds + row * RS;                                                          \
    const int vq2 = tid & (M2 - 1), vcg = (tid / M2) % GC, vtg = (tid / M2) / GC;                 \
    (void)ctl; (void)rr; (void)rowbuf; (void)vq2; (void)vcg; (void)vtg; (void)o;

    const bool xcd_map = (g.m1 & 7) == 0 && (gridDim.x & 7) == 0;
    const int xcd = xcd_map ? (blockIdx.x & 7) : 0;
    const int w = xcd_map ? (blockIdx.x >> 3) : blockIdx.x;
    const int W = xcd_map ? (gridDim.x >> 3) : gridDim.x;
    const int rows_x = xcd_map ? g.m1 / 8 : g.m1;
    const int ntiles = rows_x * g.n_ct;
    if (w >= ntiles) return;
    if (tid0 < M2) wl[tid0] = g.wL2[tid0];
    __syncthreads();
    // the threads that move the inter-pass twiddle row (128 entries): waves 0 and 1, or waves 0 and 2 where wave 1 carries no input
    const int tw_e = (HALFIN && NP == 16) ? ((tid0 >> 6) == 0 ? tid0 : ((tid0 >> 6) == 2 ? tid0 - 64 : -1)) : (tid0 < M2 ? tid0 : -1);
    // experiments (POULPY_DBG_MID_STAGGER = n, POULPY_DBG_MID_STAGGER_MOD = mode bits): n x 128 cycles of delay for the second-dispatched
    // half of the waves at the top of every inverse pass (mode bit 2: for the first half instead); mode bit 0: static priority 1 for the
    // second half, bit 1: for the first half
    const bool young = tid0 >= 256;
    if (g.stagger_mod & 1) { if (young) __builtin_amdgcn_s_setprio(1); }
    if (g.stagger_mod & 2) { if (!young) __builtin_amdgcn_s_setprio(1); }
    const int nsleep = ((g.stagger_mod & 4) ? !young : young) ? g.stagger : 0;

    // virtual tiles: V = L without C2 (stride W); with C2 V = 2 k + col for L = w + k W
    auto vL = [&](int V) { return C2 ? w + (V >> 1) * W : V; };
    auto vcol = [&](int V) { return C2 ? (V & 1) : 0; };
    auto vnext = [&](int V) { return C2 ? V + 1 : V + W; };
    const int V0 = C2 ? 0 : w;
    auto kcol = [&](int slot, int col) { return C2 ? 2 * slot + col : min(slot, g.ncomp - 1); };   // key column / output polynomial of a tile slot
    auto tile_q1 = [&](int L) { const int k = L / g.n_ct; return xcd_map ? k * 8 + xcd : k; };
    auto out_q1 = [&](int q1_) { return PERM ? (int)((g.perm_mul * (unsigned)q1_ + g.perm_add) & (unsigned)(g.m1 - 1)) : q1_; };
    auto src_ptr = [&](int L, int ctl, int rr, int o) {
        const int Lc = min(L, ntiles - 1);
        const int b_ = min((Lc % g.n_ct) * CT + ctl, g.batch - 1);
        const int r_ = min(rr, g.npi - 1);
        return g.T + ((long long)b_ * g.npi + r_) * m + (long long)tile_q1(Lc) * M2 + o;
    };
    auto in_active = [&](int L, int ctl, int rr) { return L < ntiles && (L % g.n_ct) * CT + ctl < g.batch && rr < g.npi; };
    constexpr int nrow = NR;           // product rows (launch_mid: g.row_max == NR; DS: product terms, g.ds_n == NR)
    // (DS: no rotation of the term order between workgroups - with a run-time first term every lookup in the ds_* tables is a scalar
    //  load of its own in front of the row it steers; in program order the compiler fetches the tables once)
    const int rot = DS ? 0 : w % nrow;

    cplx x[16];     // the next tile's T' values (in flight during the inverse pass), then the forward pass's working set
    cplx u[16];     // the inverse pass's working set, then the tile's results until the forward pass has stored them
    cplx* dst = nullptr;
    cplx twn = make_double2(0.0, 0.0);
    cplx kr[KR][NC];   // key-row ring: row i of a tile lives in slot i % KR; slots 0 .. KR-2 are requested before the product starts
#if PZ_MIDR_HALFKEY
#pragma unroll
    for (int i = 0; i < KR; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) kr[i][j] = make_double2(1.0, 0.0);
#endif

    // ---- pieces ----
#define PZ_XGROUP(SRC, G4)   /* four of the next tile's 16 T' loads */                                   \
    { _Pragma("unroll") for (int n1 = 4 * (G4); n1 < 4 * (G4) + 4; ++n1) x[n1] = ld_stream((SRC) + 8 * n1); }
#define PZ_SGROUP(G4)        /* four of the previous tile's 16 stores */                                \
    { _Pragma("unroll") for (int n1 = 4 * (G4); n1 < 4 * (G4) + 4; ++n1) st_stream(dst + 8 * n1, u[n1]); }
#define PZ_KGROUP(LT, SLOT)  /* key row SLOT (< KR - 1) of tile LT's product */                         \
    if (!(PZ_MIDR_HALFKEY && ((SLOT) & 1))) {                                                          \
        const long long base_ = (long long)tile_q1(min(vL(LT), ntiles - 1)) * g.nrows * g.ncols;     \
        const int kc_ = vcol(LT); (void)kc_;                                                           \
        int r_ = rot + (SLOT);                                                                         \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                                 \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                                 \
        const int krow_ = DS ? (int)g.ds_row[r_] : r_;                                                 \
        if constexpr (!PZ_MIDR_SADDR) {                                                                \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) {                                           \
                const int c_ = DS ? min(vcg * NC + j, max((int)g.ds_cb[r_], 1) - 1) + (int)g.ds_coff[r_] : kcol(vcg * NC + j, kc_); \
                kr[SLOT][j] = g.P[(base_ + (long long)krow_ * g.ncols + c_) * M2 + vq2];               \
            }                                                                                          \
        } else if constexpr (DS) {                                                                     \
            const int cb_ = max((int)g.ds_cb[r_], 1) - 1;                                              \
            const char* rp_ = (const char*)(g.P + (base_ + (long long)krow_ * g.ncols + (int)g.ds_coff[r_]) * M2); \
            _Pragma("unroll") for (int j = 0; j < NC; ++j)                                             \
                kr[SLOT][j] = *(const cplx*)(rp_ + (unsigned)(min(vcg * NC + j, cb_) * M2 + vq2) * 16u); \
        } else {   /* uniform row base + one 32-bit lane offset per column: no 64-bit vector address arithmetic (see the product loop) */ \
            const char* rp_ = (const char*)(g.P + (base_ + (long long)krow_ * g.ncols) * M2);          \
            _Pragma("unroll") for (int j = 0; j < NC; ++j)                                             \
                kr[SLOT][j] = *(const cplx*)(rp_ + (unsigned)(kcol(vcg * NC + j, kc_) * M2 + vq2) * 16u); \
        }                                                                                              \
    }
    // forward row DFT of x (see k_mid128) with the held-back stores (STORES) and the next product's first key rows in its gaps
#define PZ_MIDR_FWD(LT, STORES)                                                                   \
    {                                                                                             \
        PZ_MID_LANE                                                                               \
        if (STORES) { PZ_SGROUP(0) __builtin_amdgcn_sched_barrier(0); }                           \
        if (PZ_MID_STAMP && STORES) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); PZ_STAMP(4) } \
        if constexpr (IN) {                                                                       \
        if (!in_active(vL(LT), ctl, rr)) {                                                        \
            _Pragma("unroll") for (int n1 = 0; n1 < 16; ++n1) x[n1] = make_double2(0.0, 0.0);     \
        }                                                                                         \
        Bfly<16, false>::run(x);                                                                  \
        }                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (STORES) { PZ_SGROUP(1) __builtin_amdgcn_sched_barrier(0); }                           \
        /* twiddles W128^(o k1) in two batches of reads ahead of their multiplies: read one by one, each of the 15 sits behind its own  */ \
        /* LDS latency (the compiler does not batch them by itself)                                                                   */ \
        if constexpr (IN) _Pragma("unroll") for (int hb = 0; hb < 2; ++hb) {                      \
            cplx tw_[8];                                                                          \
            _Pragma("unroll") for (int k1 = 8 * hb; k1 < 8 * hb + 8; ++k1) tw_[k1 - 8 * hb] = wl[o * k1]; \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            _Pragma("unroll") for (int k1 = 8 * hb; k1 < 8 * hb + 8; ++k1) {                      \
                cplx v = x[k1];                                                                   \
                if (k1 > 0) v = cmul(v, tw_[k1 - 8 * hb]);                                        \
                rowbuf[k1 * 9 + o] = v;                                                           \
            }                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
        if (STORES) { PZ_SGROUP(2) __builtin_amdgcn_sched_barrier(0); }                           \
        if constexpr (IN) {                                                                       \
        row_sync();                                                                               \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                           \
            _Pragma("unroll") for (int oo = 0; oo < 8; ++oo) x[8 * h + oo] = rowbuf[(o + 8 * h) * 9 + oo]; \
        }                                                                                         \
        }                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (STORES) { PZ_SGROUP(3) __builtin_amdgcn_sched_barrier(0); }                           \
        PZ_KGROUP(LT, 0)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if constexpr (IN) Bfly<8, false>::run(x);                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        PZ_KGROUP(LT, 1)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if constexpr (IN) Bfly<8, false>::run(x + 8);                                             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if constexpr (KR > 3) PZ_KGROUP(LT, 2)                                                    \
        if constexpr (KR > 5) PZ_KGROUP(LT, 4)                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if constexpr (IN) {                                                                       \
        row_sync();                                                                               \
        _Pragma("unroll") for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 16 * k2] = x[k2];             \
        }                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if constexpr (KR > 4) PZ_KGROUP(LT, 3)                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if constexpr (IN) { _Pragma("unroll") for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 8 + 16 * k2] = x[8 + k2]; } \
        if constexpr (KR > 6) PZ_KGROUP(LT, 5)                                                    \
        if (PZ_MID_STAMP && STORES) { PZ_STAMP(5) }                                               \
        lds_barrier();                                                                            \
    }

    // ---- prologue: first tile ----
    if constexpr (IN) {
        PZ_MID_LANE
        const cplx* src_ = src_ptr(vL(V0), ctl, rr, o);
        PZ_XGROUP(src_, 0) PZ_XGROUP(src_, 1) PZ_XGROUP(src_, 2) PZ_XGROUP(src_, 3)
        if (tw_e >= 0) twrow2[tw_e] = g.tw12t[(long long)out_q1(tile_q1(vL(V0))) * M2 + tw_e];
    }
    int par = 0;
#if PZ_MID_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = 0;
    int st_tiles = 0;
#define PZ_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define PZ_STAMP(i)
#endif
    PZ_MIDR_FWD(V0, 0)
#if PZ_MID_STAMP
    st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_t;
#endif

    for (int V = V0; vL(V) < ntiles; V = vnext(V), par ^= 1) {
        const int L = vL(V), Vn = vnext(V), Ln = vL(Vn), col = vcol(V);
        (void)col; (void)Ln;
        const int q1 = tile_q1(L);
        // ---------------- product: res[b][c][q] = sum_r a[b][r][q] * P[r][c][q] ----------------
        {
            PZ_MID_LANE
            const int q2 = vq2, cg = vcg;
            cplx acc[CTt][NC];
#pragma unroll
            for (int i = 0; i < CTt; ++i)
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[i][j] = make_double2(0.0, 0.0);
            const cplx* pp[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) pp[j] = g.P + ((long long)q1 * g.nrows * g.ncols + (DS ? 0 : kcol(cg * NC + j, col))) * M2 + q2;   // DS: the column is part of the per-term offset
            const long long prow = (long long)g.ncols * M2;
            // plain product (round 4): the key row's address is a UNIFORM base (scalar registers, moved from row to row by scalar adds) plus
            // one 32-bit lane offset per column, and the operand's LDS address a lane base plus a scalar row offset.  With a 64-bit lane
            // pointer per column every load cost a 64-bit vector add, and the operand index a 64-bit multiply-add (quarter rate): 9 address
            // instructions per row beside its 64 FMAs, on the unit that paces this phase.
            const char* const kbase = (const char*)(g.P + (long long)q1 * g.nrows * g.ncols * M2);
            unsigned koff[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) koff[j] = (unsigned)(kcol(cg * NC + j, col) * M2 + q2) * 16u;
            const cplx* const avp = lds + vtg * CTt * NP * RS + q2;
            cplx av[2][CTt];
#define PZ_LOADROW(DST, IT)                                                                     \
    if (!(PZ_MIDR_HALFKEY && ((IT) & 1))) {                                                     \
        int r_ = (IT) + rot;                                                                    \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                          \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                          \
        if constexpr (DS && !PZ_MIDR_SADDR) {                                                   \
            const int cb_ = max((int)g.ds_cb[r_], 1) - 1, co_ = (int)g.ds_coff[r_];             \
            const long long ro_ = (long long)g.ds_row[r_] * prow;                               \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) DST[j] = pp[j][ro_ + (long long)(min(cg * NC + j, cb_) + co_) * M2]; \
        } else if constexpr (DS) {   /* uniform base of the term's key row and column window, lane offset of the (clamped) column */ \
            const int cb_ = max((int)g.ds_cb[r_], 1) - 1;                                       \
            const char* rp_ = kbase + ((long long)g.ds_row[r_] * prow + (long long)g.ds_coff[r_] * M2) * 16; \
            _Pragma("unroll") for (int j = 0; j < NC; ++j)                                      \
                DST[j] = *(const cplx*)(rp_ + (unsigned)(min(cg * NC + j, cb_) * M2 + q2) * 16u); \
        } else if constexpr (!PZ_MIDR_SADDR) {                                                  \
            const long long off_ = (long long)r_ * prow;                                        \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) DST[j] = pp[j][off_];                \
        } else {                                                                                \
            const char* rp_ = kbase + (long long)r_ * prow * 16;                                \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) DST[j] = *(const cplx*)(rp_ + koff[j]); \
        }                                                                                       \
    }
#define PZ_AVLOAD(DST, IT)                                                                      \
    {                                                                                           \
        int r_ = (IT) + rot;                                                                    \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                          \
        r_ -= (r_ >= nrow) ? nrow : 0;                                                          \
        const int slot_ = DS ? (int)g.ds_in[r_] : r_;   /* DS: term r_ multiplies input polynomial ds_in[r_] */ \
        if constexpr (!PZ_MIDR_SADDR) {                                                         \
            _Pragma("unroll") for (int i = 0; i < CTt; ++i) DST[i] = lds[((vtg * CTt + i) * NP + slot_) * RS + q2]; \
        } else {                                                                                \
            const cplx* ar_ = avp + slot_ * RS;                                                 \
            _Pragma("unroll") for (int i = 0; i < CTt; ++i) DST[i] = ar_[i * NP * RS];          \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);   /* the machine scheduler otherwise sinks these reads down to their first use */ \
    }
#define PZ_FMAROW(AV, SRC, IT)                                                                  \
    {                                                                                           \
        if constexpr (DS) {   /* the term only reaches the first ds_cb outputs */               \
            int r_ = (IT) + rot;                                                                \
            r_ -= (r_ >= nrow) ? nrow : 0;                                                      \
            r_ -= (r_ >= nrow) ? nrow : 0;                                                      \
            const int cbv_ = (int)g.ds_cb[r_];                                                  \
            _Pragma("unroll") for (int j = 0; j < NC; ++j)                                      \
                if (cg * NC + j >= cbv_) SRC[j] = make_double2(0.0, 0.0);                       \
        }                                                                                       \
        _Pragma("unroll") for (int i = 0; i < CTt; ++i) {                                       \
            _Pragma("unroll") for (int j = 0; j < NC; ++j) {                                    \
                acc[i][j].x = __builtin_fma(AV[i].x, SRC[j].x, acc[i][j].x);                    \
                acc[i][j].x = __builtin_fma(-AV[i].y, SRC[j].y, acc[i][j].x);                   \
                acc[i][j].y = __builtin_fma(AV[i].x, SRC[j].y, acc[i][j].y);                    \
                acc[i][j].y = __builtin_fma(AV[i].y, SRC[j].x, acc[i][j].y);                    \
            }                                                                                   \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    }
            // row `it` is in slot it % KR; the row loop is fully unrolled (NR is a template parameter) so that the slots are compile-time
            // registers.  Rows are requested KR - 1 ahead; none is requested past the end (the first KR - 1 rows of the next tile are
            // requested by its forward pass).
            PZ_AVLOAD(av[0], 0)
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                if (it + KR - 1 < NR) PZ_LOADROW(kr[(it + KR - 1) % KR], it + KR - 1)
                if (it + 1 < NR) PZ_AVLOAD(av[(it + 1) & 1], it + 1)
                PZ_FMAROW(av[it & 1], kr[it % KR], it)
            }
#undef PZ_LOADROW
#undef PZ_AVLOAD
#undef PZ_FMAROW
            PZ_STAMP(0)
            lds_barrier();  // every a value has been read: the tile can be overwritten with the products
            PZ_STAMP(1)
            {
                // one lane address, constant offsets per (ciphertext, column); the zero-fill of columns beyond ncomp only where there are any
                // (a 64-bit multiply-add and four selects per value otherwise: 96 instructions for these 16 stores)
                const int q2o = PERM ? (int)((((g.perm_mul * (unsigned)q1 + g.perm_add) >> g.log_m1) + g.perm_mul * (unsigned)q2) & (unsigned)(M2 - 1)) : q2;
                cplx* const wb = lds + (vtg * CTt * NP + cg * NC) * RS + q2o;
                if (C2 || g.ncomp >= NP) {
#pragma unroll
                    for (int i = 0; i < CTt; ++i)
#pragma unroll
                        for (int j = 0; j < NC; ++j) wb[(i * NP + j) * RS] = acc[i][j];
                } else {
#pragma unroll
                    for (int i = 0; i < CTt; ++i)
#pragma unroll
                        for (int j = 0; j < NC; ++j) wb[(i * NP + j) * RS] = (cg * NC + j < g.ncomp) ? acc[i][j] : make_double2(0.0, 0.0);
                }
            }
            lds_barrier();
            PZ_STAMP(2)
        }
#if PZ_MIDR_ILV
        // ---------------- inverse row DFT of this tile and forward row DFT of the next one, INTERLEAVED (round 4) ----------------
        // The two transforms work on different registers (u / x) and share only the wave's own rows of the tile, which the inverse pass
        // is done with once it has read its second exchange back.  From there on every LDS batch of the forward pass is followed by a
        // piece of the inverse pass's remaining arithmetic (last butterfly, inter-pass twiddles) instead of a wait: within the row passes
        // VALU, LDS and the memory path used to take turns (their busy times ADD UP to the passes' 16 k cycles per tile).
        // Same arithmetic in the same order per value: bit-identical.  Middle kernel 4.52 -> 4.38 ms per 1024 (profiles/r04_ab_midr_interleave.txt).
        // Measured and dropped on the way (same file): the next tile's T' loads issued earlier - behind the last key rows of the product, or
        // between the accumulator stores of the write-back - so that the forward pass's first butterfly could also fill the inverse pass's
        // FIRST exchange wait: both lose the gain again (4.50 - 4.52 ms; the loads then compete with the key stream / delay the write-back).
        for (int i = 0; i < nsleep; ++i) __builtin_amdgcn_s_sleep(2);
        {
            PZ_MID_LANE
            const int b = (L % g.n_ct) * CT + ctl;
            const cplx* src_ = src_ptr(Ln, ctl, rr, o);
            const cplx* twr = twrow2 + par * M2;
            if constexpr (IN) twn = g.tw12t[(long long)out_q1(tile_q1(min(Ln, ntiles - 1))) * M2 + max(tw_e, 0)];
            if constexpr (IN) { PZ_XGROUP(src_, 0) PZ_XGROUP(src_, 1) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) u[8 * h + k2] = rowbuf[o + 8 * h + 16 * k2];
            if constexpr (PERM) {
#pragma unroll
                for (int t = 0; t < 16; ++t) u[t].y *= g.perm_ysign;
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) PZ_XGROUP(src_, 2)
            __builtin_amdgcn_sched_barrier(0);
            Bfly<8, true>::run(u);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) PZ_XGROUP(src_, 3)
            __builtin_amdgcn_sched_barrier(0);
            Bfly<8, true>::run(u + 8);
            row_sync();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k1 = o + 8 * h;
                cplx tw_[8];
#pragma unroll
                for (int oo = 1; oo < 8; ++oo) tw_[oo] = wl[oo * k1];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) {
                    cplx v = u[8 * h + oo];
                    if (oo > 0) v = cmulc(v, tw_[oo]);
                    rowbuf[k1 * 9 + oo] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            row_sync();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 9 + o];
            __builtin_amdgcn_sched_barrier(0);
            // ---- the wave's rows are free from here: the forward pass of the next tile starts while u is still being finished ----
            if constexpr (IN) {
                if (!in_active(Ln, ctl, rr)) {
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) x[n1] = make_double2(0.0, 0.0);
                }
                Bfly<16, false>::run(x);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    cplx tw_[8];
#pragma unroll
                    for (int k1 = 8 * hb; k1 < 8 * hb + 8; ++k1) tw_[k1 - 8 * hb] = wl[o * k1];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k1 = 8 * hb; k1 < 8 * hb + 8; ++k1) {
                        cplx v = x[k1];
                        if (k1 > 0) v = cmul(v, tw_[k1 - 8 * hb]);
                        rowbuf[k1 * 9 + o] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // (forward exchange writes in flight) last butterfly of the inverse pass
            Bfly<16, true>::run(u);
            __builtin_amdgcn_sched_barrier(0);
            cplx twa[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) twa[n1] = twr[o + 8 * n1];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) {
                row_sync();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int oo = 0; oo < 8; ++oo) x[8 * h + oo] = rowbuf[(o + 8 * h) * 9 + oo];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (forward exchange reads in flight) inter-pass twiddles of the inverse pass, first half; the second half's roots behind them
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) u[n1] = cmulc(u[n1], twa[n1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) twa[n1] = twr[o + 8 * (n1 + 8)];
            __builtin_amdgcn_sched_barrier(0);
            const bool active = b < g.batch && rr < (C2 ? NP : g.npo);
            dst = active ? g.T2 + ((long long)b * g.npo + (C2 ? 2 * rr + col : rr)) * m + (long long)out_q1(q1) * M2 + o
                         : g.dummy + ((long long)blockIdx.x * (NT / 8) + row) * M2 + o;
            PZ_SGROUP(0)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) Bfly<8, false>::run(x);
            __builtin_amdgcn_sched_barrier(0);
            PZ_SGROUP(1)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) u[n1 + 8] = cmulc(u[n1 + 8], twa[n1]);
            if constexpr (IN) { if (tw_e >= 0) twrow2[(par ^ 1) * M2 + tw_e] = twn; }
            __builtin_amdgcn_sched_barrier(0);
            PZ_KGROUP(Vn, 0)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) Bfly<8, false>::run(x + 8);
            __builtin_amdgcn_sched_barrier(0);
            PZ_SGROUP(2)
            PZ_KGROUP(Vn, 1)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) {
                row_sync();
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 16 * k2] = x[k2];
            }
            __builtin_amdgcn_sched_barrier(0);
            PZ_SGROUP(3)
            if constexpr (KR > 3) PZ_KGROUP(Vn, 2)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) rowbuf[o + 8 + 16 * k2] = x[8 + k2];
            }
            if constexpr (KR > 4) PZ_KGROUP(Vn, 3)
            if constexpr (KR > 5) PZ_KGROUP(Vn, 4)
            if constexpr (KR > 6) PZ_KGROUP(Vn, 5)
            PZ_STAMP(3)
            lds_barrier();
        }
#else
        // ---------------- inverse row DFT of the wave's own 8 rows; the next tile's loads go out in its gaps ----------------
        for (int i = 0; i < nsleep; ++i) __builtin_amdgcn_s_sleep(2);
        {
            PZ_MID_LANE
            const int b = (L % g.n_ct) * CT + ctl;
            const cplx* src_ = src_ptr(Ln, ctl, rr, o);
            const cplx* twr = twrow2 + par * M2;
            if constexpr (IN) twn = g.tw12t[(long long)out_q1(tile_q1(min(Ln, ntiles - 1))) * M2 + max(tw_e, 0)];
            if constexpr (IN) PZ_XGROUP(src_, 0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) u[8 * h + k2] = rowbuf[o + 8 * h + 16 * k2];
            if constexpr (PERM) {   // Galois elements 3 mod 4: the conjugate of the permuted product (MidArgs::perm_ysign; +1.0 otherwise)
#pragma unroll
                for (int t = 0; t < 16; ++t) u[t].y *= g.perm_ysign;
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) PZ_XGROUP(src_, 1)
            __builtin_amdgcn_sched_barrier(0);
            Bfly<8, true>::run(u);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (IN) PZ_XGROUP(src_, 2)
            __builtin_amdgcn_sched_barrier(0);
            Bfly<8, true>::run(u + 8);
            row_sync();
            // conj W128^(oo k1): read in two batches ahead of their multiplies, for every lane (k1 = 0 reads W^0 = 1: the product is
            // exact), so that no lane-dependent branch cuts the batch
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k1 = o + 8 * h;
                cplx tw_[8];
#pragma unroll
                for (int oo = 1; oo < 8; ++oo) tw_[oo] = wl[oo * k1];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int oo = 0; oo < 8; ++oo) {
                    cplx v = u[8 * h + oo];
                    if (oo > 0) v = cmulc(v, tw_[oo]);
                    rowbuf[k1 * 9 + oo] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (IN) PZ_XGROUP(src_, 3)
            __builtin_amdgcn_sched_barrier(0);
            row_sync();
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) u[k1] = rowbuf[k1 * 9 + o];
            Bfly<16, true>::run(u);
            __builtin_amdgcn_sched_barrier(0);
            // inter-pass twiddles in two batches of 8 reads (all 16 at once, beside u[] and the 64 registers of x[] in flight, spill)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                cplx tw_[8];
#pragma unroll
                for (int n1 = 8 * hb; n1 < 8 * hb + 8; ++n1) tw_[n1 - 8 * hb] = twr[o + 8 * n1];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n1 = 8 * hb; n1 < 8 * hb + 8; ++n1) u[n1] = cmulc(u[n1], tw_[n1 - 8 * hb]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the next tile's twiddle row (requested FIRST among this pass's loads, so that waiting for it here leaves the 16 T' loads in
            // flight) goes to the row the next inverse pass reads
            if constexpr (IN) { if (tw_e >= 0) twrow2[(par ^ 1) * M2 + tw_e] = twn; }
            const bool active = b < g.batch && rr < (C2 ? NP : g.npo);
            dst = active ? g.T2 + ((long long)b * g.npo + (C2 ? 2 * rr + col : rr)) * m + (long long)out_q1(q1) * M2 + o
                         : g.dummy + ((long long)blockIdx.x * (NT / 8) + row) * M2 + o;
        }
        PZ_STAMP(3)
        // no workgroup barrier: the forward pass below only rewrites this wave's own rows, and it writes the OTHER twiddle row
        PZ_MIDR_FWD(Vn, 1)
#endif
        PZ_STAMP(6)
#if PZ_MID_STAMP
        ++st_tiles;
#endif
    }
#if PZ_MID_STAMP
    if ((tid0 & 63) == 0 && (blockIdx.x == 0 || blockIdx.x == 9 || blockIdx.x == 130 || blockIdx.x == 255))
        printf("STAMP wg %d wave %d tiles %d total %llu | product %llu bar1 %llu accwr %llu inv %llu xwait %llu fwd %llu bar6 %llu\n",
               (int)blockIdx.x, tid0 >> 6, st_tiles, (unsigned long long)(st_t - st_t0), st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4],
               st_acc[5], st_acc[6]);
#endif
#undef PZ_STAMP
#undef PZ_XGROUP
#undef PZ_SGROUP
#undef PZ_KGROUP
#undef PZ_MIDR_FWD
#undef PZ_MID_LANE
}

// (Round 4 experiment, removed - git history has it: the inverse inter-pass twiddle (x conj tw12) left to k_inv_tail.  Bit-exact; this kernel
//  did not move (4.50 -> 4.53 ms per 1024 although it dropped from 256 to 219 VGPRs, 16 LDS reads and 64 flops per thread-tile: its inverse
//  row pass is not what bounds the tile) and the tail paid 0.19 ms for the extra L2 stream: profiles/r04_ab_twtail.txt.)

// HALFIN: a 16-slot tile whose ciphertexts carry at most 8 input polynomials (key switch, automorphism, ggsw_expand_row): the waves of
// the upper 8 slots run the IN = false code.
// (32-slot tiles - 16 limbs, rank 2-3 - carry 8 key values per thread and row: a ring of 3 slots there)
// DS (round 3, late): the digit-group product of dsize > 1 (MidArgs::ds_*: NR product terms, each with its input slot, key row, column
// offset and column bound) - the addressing of k_mid128<.., DS> on this kernel's schedule.
template <int CT, int NP, bool PERM, int NR, bool HALFIN = false, int KR = ((NP == 32 || CT * NP * 8 == 256) ? 3 : PZ_MIDR_KR), bool DS = false, bool C2 = false>
__global__ void __launch_bounds__(CT * NP * 8)
k_mid128r(MidArgs g) {
    static_assert(!HALFIN || NP >= 16, "HALFIN: 16- and 32-slot tiles");
    static_assert(!DS || !PERM, "digit groups: plain product only");
    if constexpr (HALFIN) {
        if (((threadIdx.x >> 3) % NP) >= NP / 2) {   // wave-uniform: a wave owns 8 consecutive slots of one ciphertext
            mid128r_body<CT, NP, PERM, NR, KR, true, false, DS, C2>(g);
            return;
        }
    }
    mid128r_body<CT, NP, PERM, NR, KR, HALFIN, true, DS, C2>(g);
}

// (Round 2 experiment, removed: "k_midr<LPR = 4>" — the same kernel for rows of 64 points owned by 4 lanes (m = 512 x 64): a tile of four
//  ciphertexts is then 65 KiB and TWO 256-thread workgroups share a CU at the same four ciphertexts per key fetch.  Timed at that geometry:
//  5.7 ms against 5.0 ms (profiles/r02_ab_mid64_rows.txt).  With the T' / T2' traffic removed both shapes take 4.1-4.2 ms: the on-CU work
//  (LDS exchanges, product FMAs + key through L1, butterflies) is serial in both and a second workgroup does not overlap it any better,
//  while the 64-byte global runs of 4-lane rows stream worse.)

// (Round 2 experiment, removed: "k_mid128L" — four extra loader waves that request tile t+1 at the top of iteration t, hold it in
//  registers and hand it over through LDS once the compute waves release the tile, so that HBM loads never sit in front of the
//  compute waves' L2-served key loads.  Bit-exact, but 9 % SLOWER (5.19 vs 4.76 ms per 1024 ciphertexts,
//  profiles/r02_ab_mid_loader_waves.txt): the timing ablation of this kernel (POULPY_DBG_MID_SKIP, profiles/r02_mid_ablation.txt)
//  shows why — with every global access, every FMA and every butterfly removed it still takes 1.97 of its 4.96 ms (LDS traffic of
//  the seven exchange passes + the product's operand reads, barriers), and no single component is worth more than 1 ms: HBM waits
//  are not the exposed part, so hiding them buys nothing while the extra LDS pass and the 168-VGPR cap cost.)

// standard device VmpPMat  P[p][q1 + m1*q2]  ->  P'[q1][p][q2]   (p = r*ncols + c), 16x16 tiles through LDS
__global__ void __launch_bounds__(256)
k_permute_pmat(const cplx* __restrict__ P, cplx* __restrict__ Pp, int npolys, int m1, int m2) {
    __shared__ cplx tile[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_q1 = m1 / 16, tiles_q2 = m2 / 16;
    int t = blockIdx.x;
    const int tq1 = t % tiles_q1; t /= tiles_q1;
    const int tq2 = t % tiles_q2; t /= tiles_q2;
    const int p = t;
    const long long m = (long long)m1 * m2;
    tile[ty][tx] = P[(long long)p * m + (long long)(tq2 * 16 + ty) * m1 + tq1 * 16 + tx];
    __syncthreads();
    Pp[((long long)(tq1 * 16 + ty) * npolys + p) * m2 + tq2 * 16 + tx] = tile[tx][ty];
}

}  // namespace pz
