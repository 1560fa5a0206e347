// internal.hpp — interfaces between the translation units of libpoulpy_hip.so.
//
//   launch_fft.hip   the four passes of the two-pass negacyclic FFT (device_fft.hpp)
//   launch_tail.hip  fused inverse column pass + carry chain (k_inv_tail)
//   launch_mid.hip   fused row pass + VMP + inverse row pass (device_mid.hpp), key re-slicing
//   launch_small.hip two-kernel pipeline for N = 4096: whole polynomials in LDS (device_small.hpp)
//   launch_ops.hip   elementwise / permutation / normalize / VMP kernels (device_ops.hpp)
//   launch_br.hip    blind-rotation kernels (device_br.hpp + the block step of device_ops.hpp)
//   launch_cnv.hip   bivariate convolution kernels (device_cnv.hpp)
//   api.hip          C ABI: module, memory, the batched GLWE product (glwe_op) and its direct callers, key pinning / mirrors
//   api_hal.hip      C ABI: the per-op HalImpl methods (VecZnxDft, SVP, VMP, VecZnxBig, i64 VecZnx family)
//   api_br.hip       C ABI: blind rotation, circuit bootstrapping, GLWE packing (composites on glwe_op; HIP-graph replay)
//   api_cnv.hip      C ABI: convolution family, GLWE tensoring, batched i64 family
//   api_lwe.hip      C ABI: LWE glue of the gate bootstrap (+ its index kernels)
//   api_dist.hip     C ABI: RCCL key broadcast (dlopen)
//   api_common.hpp / api_glwe.hpp   what those share (staging, graph cache, glwe_op / glwe_trace / ggsw_expand_row)
//
// Every kernel is instantiated in exactly one translation unit (the one that launches it), so the units compile
// independently and in parallel; only plain functions cross unit boundaries.
#pragma once
#include "module.hpp"

namespace pz {

// ---- shared small helpers -----------------------------------------------------------------------------------------
template <typename K>
inline int set_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        PZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PZ_OK;
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DV {          // a batched VecZnx-like container on the device
    void* p;
    long long bs;    // scalars between consecutive batch objects
    int cols, size;
};
inline long long limb_stride(const pz_module* M, const DV& v) { return (long long)v.cols * (long long)M->n; }
inline void* poly_ptr(const pz_module* M, const DV& v, int col, int limb) {
    return (char*)v.p + 8 * ((long long)M->n * ((long long)limb * v.cols + col));
}

// ---- launch_fft.hip -----------------------------------------------------------------------------------------------
int launch_fwd_pass1(pz_module* M, int npolys, const long long* src, PolyMap smap, cplx* T, bool rowmajor = false, long long mask = -1,
                     bool src32 = false);
// the row-major form on 16-bit digits in the fused tail's tile order (TailD16; smap addresses limbs of n int16)
int launch_fwd_pass1_t16(pz_module* M, int npolys, const short* src, PolyMap smap, cplx* T);
// the row-major form on i64 coefficients that also leaves them as 16-bit values in the tile order, polynomial p at w16 + p n, and raises the module's
// wide flag (module.hpp: wide16) for a value beyond 16 bits
int launch_fwd_pass1_w16(pz_module* M, int npolys, const long long* src, PolyMap smap, cplx* T, short* w16);
bool tail_d16_only_supported(const pz_module* M);   // the tensoring tails that leave 16-bit digits only + the operand form that reads them   // src32 (row-major, 128-point-row plans): `src` holds 32-bit digits at the same element offsets
int launch_fwd_pass2(pz_module* M, int npolys, const cplx* T, double* dst, PolyMap dmap, const cplx* mul);
int launch_inv_pass2(pz_module* M, int npolys, const double* src, PolyMap smap, cplx* T);
int launch_inv_pass1(pz_module* M, int npolys, const cplx* T, long long* dst, PolyMap dmap);

// ---- launch_tail.hip ----------------------------------------------------------------------------------------------
bool tail_supported(const pz_module* M);
// One launch of the fused tail (k_inv_tail, device_fft.hpp): the big value held in T - nlimbs x ncols polynomials per ciphertext, still
// one inverse column pass away from coefficients - leaves as normalized base2k digits in `res`.  Every field has a name at the call
// site (round 3 passed these as 28 positional arguments, 8 of them bool); the defaults are the plain external-product tail.
struct TailCall {
    // ---- the value and where its digits go ----
    int batch = 0;
    const cplx* T = nullptr;          // T[j2][q1] (output of inverse pass 2) or, rowmajor, T2'[q1][j2] (output of the middle kernel)
    bool rowmajor = false;
    int nlimbs = 0, ncols = 0;        // limbs / columns of the VecZnxBig being consumed
    long long* res = nullptr;
    long long res_bs = 0;             // i64 elements between consecutive ciphertexts of res
    int res_cols = 0, res_size = 0;
    int base2k = 0;
    // ---- operand added in front of the carry chain (vec_znx_big_add_small_assign, keyswitching/glwe.rs:237) ----
    const long long* small = nullptr; // null: none (external product)
    long long small_bs = 0;
    int small_cols = 1, small_size = 0;
    bool small_all = false;           // false: column 0 of `small` lands on body_col only; true: column c of `small` on column c
    bool small_neg = false;           // the operand is subtracted (sub forms of the automorphism family)
    int body_col = 0;                 // 0 for a key switch, `col` for ggsw_expand_row
    // ---- body-column operand prepared elsewhere (spectral automorphism forms: phi(body) +- a0 in the workspace) ----
    const long long* body_src = nullptr;
    long long body_bs = 0, body_ls = 0;   // batch / limb strides of body_src
    bool body_only = false;               // only the body column has an operand (plain glwe_automorphism)
    bool body_add = false;                // body column: body_src[n] + small[body column][n] (spectral add / sub forms: the pre-pass only permutes)
    bool body_gather = false;             // instead of body_src: the tail gathers +-phi(body) from column 0 of `small` itself (gather_mul,
                                          // gather_neg) - no pre-pass
    // the pre-pass left the body-column operand as 16-bit values in the tail's tile order, body16[ciphertext][limb][n] with body16_limbs limbs per
    // ciphertext, and raised *body16_wide if a value did not fit; launch_inv_tail then runs that column twice - the 16-bit-operand form (returns at
    // once if the flag is up) and the operand variant on body_src, which a conditional second pre-pass filled (returns at once if the flag is down)
    const short* body16 = nullptr;
    int body16_limbs = 0;
    const unsigned* body16_wide = nullptr;
    // add / sub forms at rank 1: the other column's operand +-a[1] as 16-bit values too (pass 1 of the key switch left them: launch_fwd_pass1_w16),
    // other16[ciphertext][limb][n] with small_size limbs per ciphertext; same flag, same pair of launches on that column
    const short* other16 = nullptr;
    // ---- signs of X -> X^p (automorphism/glwe_ct.rs:96-275; TailArgs in device_fft.hpp) ----
    unsigned auto_mul = 0;            // != 0: the value enters the chain as s(n) (big + small), s(n) = -1 iff (n auto_mul) mod 2N >= N
    bool auto_neg = false;            // flips every s(n)
    bool post_neg = false;            // put s(n) back on the digits (plain form: phi acts on the normalized value)
    unsigned gather_mul = 0;          // != 0: the operand is -+phi^-1(small), gathered inside the tail (older, non-spectral scheme)
    bool gather_neg = false;
    // ---- blind rotation's accumulator between two blocks: 32-bit digits (bit 0: `small`, bit 1: `res`; same element strides) ----
    int acc32 = 0;
    // bit 2 (with small_all): the operand is a GLWETensor held as 16-bit digits in the tail's tile order, small16[column][ciphertext][limb][n],
    // small16_cs int16 elements between columns (`small` only has to be non-null then)
    const short* small16 = nullptr;
    long long small16_cs = 0;
    // ---- glwe_trace: the digits leave through a one-bit vec_znx_rsh_assign ----
    bool post_rsh = false;
};
int launch_inv_tail(pz_module* M, const TailCall& c);
bool tail_rsh_supported(const pz_module* M);
bool tail_acc32_supported(const pz_module* M);   // 32-bit accumulator digits (TailCall::acc32; launch_fwd_pass1's src32 covers the same plans)
bool mid_cnv_supported(const pz_module* M, int a_size, int b_size, int min_size);
int launch_mid_cnv(pz_module* M, int batch, const cplx* a_main, const cplx* a_last, const cplx* b_main, const cplx* b_last, cplx* T2, int cols,
                   int a_size, int b_size, int a_i, int a_j, int b_i, int b_j, int min_size, int offset);
// k_mid_cnv3: the three terms of a rank-1 tensoring in one launch; T2 = [term][pair][limb < min_size][m] (launch_cnv.hip)
bool mid_cnv3_supported(const pz_module* M, int cols, int a_size, int b_size, int min_size);
int launch_mid_cnv3(pz_module* M, int batch, const cplx* a_main, const cplx* a_last, const cplx* b_main, const cplx* b_last, cplx* T2, int a_size,
                    int min_size, int offset);
struct NzCombine;
// 16-bit side copies of the diagonal terms' digits (round 6; base2k <= 16): [pair][res limb][n] int16 in the tail's own tile order.  A diagonal
// launch (NZ1) mirrors every digit it stores into `w`; the pairwise launch (mode 5) reads `ra` / `rb` instead of the low dwords of the two
// i64 tensor columns (8 B fetched per coefficient and column for a 12-bit digit: 4.3 of the pairwise launch's 8.8 GB, profiles/r04_tensor_traffic.json)
// only: the digits leave ONLY as those copies, the i64 column is not written (the fused multiply + relinearize, api_cnv.hip: the pairwise launch
// then writes its own values pair - d_i - d_j into `w` too, base2k <= 14)
struct TailD16 { short* w = nullptr; const short* ra = nullptr; const short* rb = nullptr; bool only = false; };
int launch_inv_tail_nz(pz_module* M, int batch, const cplx* T, int nlimbs, long long* res, long long res_bs, int res_cols, int res_size, int res_col,
                       int base2k, long long res_offset, int a_size, const NzCombine* cb, const TailD16* d16 = nullptr);

// ---- launch_mid.hip -----------------------------------------------------------------------------------------------
// scratch rows behind T2: one 64-row x 128-point tile per persistent workgroup of k_mid128 (<= 256 of them: 32 MiB), which also covers
// the 512 x 256 points k_mid<CT> shares
constexpr size_t kMidDummyBytes = (size_t)256 * 64 * 128 * sizeof(cplx) + (1 << 20);
bool mid_supported(const pz_module* M, int npi, int npo);
int launch_permute_pmat(pz_module* M, const double* P, cplx* Pp, int npolys);
// perm_mul != 0: spectrum permutation of X -> X^p folded into the middle kernel (m2 = 128 plans only; see MidArgs)
// digit-selected product (dsize > 1, m2 = 128 plans): term t = input polynomial in[t] x key row row[t], columns shifted by coff[t],
// reaching the first cb[t] output polynomials
struct MidDigits {
    int n = 0;
    unsigned char in[32], row[32], coff[32], cb[32];
};
// CGGI block step (m2 = 128 plans): Pp = the row-sliced copy of the blk GGSWs of one LWE block (blk * nrows key rows per frequency
// row), nrows = npi; the products are weighted by the monomial factors DFT(X^a_i - 1) of each ciphertext (MidArgs, BR)
struct MidBr {
    const long long* lwe;   // [batch][lwe_bs] mod-switched LWEs
    long long lwe_bs;
    int i0, blk;            // first coefficient of the block, block size (<= 16)
};
int launch_mid(pz_module* M, int batch, const cplx* T, cplx* T2, const cplx* Pp, int npi, int npo, int nrows, int ncols, cplx* dummy,
               unsigned perm_mul = 0, unsigned perm_add = 0, const MidDigits* dg = nullptr, const MidBr* br = nullptr, bool perm_conj = false);

// ---- launch_small.hip ---------------------------------------------------------------------------------------------
// two-kernel pipeline for N = 1024 / 2048 / 4096 (device_small.hpp): full forward transform -> S[poly][q1][q2]; product with the row-sliced key +
// full inverse transform + carry chain per (ciphertext, output column).  dsize 1, one base2k, <= 4 key limbs.
bool small_supported(const pz_module* M, int npi, int key_limbs);
// standard device VmpPMat -> P'[q1][p][q2] with m = M1 x 128 (ring degrees whose plan is not M1 x 128: no launch_permute_pmat there)
int launch_small_permute(pz_module* M, const double* P, cplx* Pp, int npolys);
int launch_small_fwd(pz_module* M, int npolys, const long long* src, PolyMap smap, cplx* S, bool natural_order = false,
                     const PolyMap* dmap = nullptr, const cplx* mul = nullptr);
bool small_transform_supported(const pz_module* M);   // N = 1024 / 2048 / 4096: per-op transforms in one kernel (launch_small.hip)
int launch_small_idft(pz_module* M, int npolys, const double* a, PolyMap smap, long long* res, PolyMap dmap);
int launch_small_inv(pz_module* M, int batch, const cplx* S, const cplx* Pp, int npi, int nrows, int ncols, int cols_out, int ksz,
                     long long* res, long long res_bs, int res_cols, int res_size, const long long* small, long long small_bs,
                     int small_cols, int small_size, int base2k, int body_col, bool noprod = false, cplx* fwd_S = nullptr, int fwd_limbs = 0,
                     bool au = false, unsigned au_p = 0, int au_mode = 0, bool post_rsh = false, int acc32 = 0);
// ONE kernel per call for N = 1024 / 2048 (device_small_one.hpp): forward transforms, product, inverse transforms and carry chains of a ciphertext in
// one workgroup; rank 1 (2 output columns), <= 8 input polynomials, <= 4 key limbs, key columns = ksz * 2
bool small_one_supported(const pz_module* M, int npi, int nrows, int ncols, int cols_out, int ksz, int batch);
int launch_small_one(pz_module* M, int batch, const long long* src, PolyMap smap, const cplx* Pp, int npi, int nrows, int ncols, int ksz, long long* res,
                     long long res_bs, int res_cols, int res_size, const long long* small, long long small_bs, int small_cols, int small_size, int base2k,
                     int body_col);

// ---- launch_ops.hip -----------------------------------------------------------------------------------------------
int launch_ew(pz_module* M, int op, void* res, long long res_bs, long long res_ls, const void* a, long long a_bs,
              long long a_ls, const void* b, long long b_bs, long long b_ls, int nlimbs, int batch);
// zero-fill as a kernel node (not hipMemsetAsync: see launch_ops.hip) - for everything that can run under graph capture
int launch_zero_bytes(pz_module* M, void* ptr, size_t bytes);
// dst = +-src(X^p-gather with multiplier mul) [+ add]; see k_automorphism
int launch_automorphism(pz_module* M, int npolys, const long long* src, PolyMap sm, long long* dst, PolyMap dm, unsigned mul,
                        int flags, const long long* add = nullptr, PolyMap am = PolyMap{1, 1, 0, 0, 0, 0}, short* dst16 = nullptr);
int launch_rotate(pz_module* M, int npolys, const long long* src, PolyMap sm, long long* dst, PolyMap dm, int mode,
                  int polys_per_batch, const long long* shift, long long shift_bs, long long shift_idx, long long shift_const);
int launch_rsh(pz_module* M, int batch, long long* data, long long bs, int cols, int size, int col0, int ncols, int base2k, int k);
// vmp_apply_dft_to_dft  [vmp.rs:144-264, zero-tail semantics for limb_offset > 0]
int dev_vmp(pz_module* M, int batch, DV res, DV a, const double* pmat, int rows, int cols_in, int cols_out, int size, int limb_offset);
// vec_znx_(big_)normalize on one column  [normalize.rs:18-401]
// cb (same base2k only): how the digits reach res_col (mode) and up to two more columns of `res` that take them too - 1 = v, 2 = -v,
// 3 += v, 4 -= v (wrapping), 0 none; nullptr: plain stores
struct NzCombine { int mode; int col2[2]; int mode2[2]; };
int dev_normalize(pz_module* M, int batch, DV res, int res_base2k, long long res_offset, int res_col, DV a, int a_base2k, int a_col,
                  const NzCombine* cb = nullptr);

// ---- launch_br.hip ------------------------------------------------------------------------------------------------
// whole rotation in one kernel (device_br.hpp) when the shape fits: *launched says whether it did
int br_try_fused(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                 const pz_blind_rotation_params* p, size_t batch, bool* launched);
// zero + block_size x (vmp, svp, add, sub) of one LWE block in one kernel; *launched = false: shape not covered
int br_block_step(pz_module* M, const double* acc_dft, long long a_bs, double* acc_add, long long o_bs, const double* brk,
                  size_t pmat_doubles, int row_max, int ncols, int B, int i0, int blk, const int64_t* lwe_2n, long long lwe_bs,
                  bool* launched);
int launch_xai_acc(pz_module* M, double* acc_add, long long acc_bs, const double* v, long long v_bs, int polys, int B,
                   const int64_t* lwe_2n, long long lwe_bs, int idx);
int launch_xai_ext(pz_module* M, double* acc_add, const double* v, int polys, int log_ext, int B, const int64_t* lwe_2n, long long lwe_bs,
                   int idx);
int launch_br_ext_init(pz_module* M, int64_t* acc, const int64_t* lut, const int64_t* lwe_2n, long long lwe_bs, int log_ext, int cols,
                       int rsz, int lut_size, int B);

// ---- launch_cnv.hip -----------------------------------------------------------------------------------------------
// bivariate convolution in the DFT domain (reference/fft64/convolution.rs:210-345): res limb kk of column res_col =
//   sum_j A[kk + offset - j] * B[j],  A = a[col_i] (+ a[col_j] when pairwise), B = b[b_i] (+ b[b_j]); limbs >= min_size untouched here.
// a / b: device CnvPVec layout [col][limb][m points], batch strides in doubles; res: VecZnxDft layout.
int launch_cnv_apply(pz_module* M, int batch, double* res, long long res_bs, int res_cols, int res_col, int min_size, int offset,
                     const double* a, long long a_bs, int a_size, int a_i, int a_j, const double* b, long long b_bs, int b_size, int b_i,
                     int b_j);
// convolution.rs:147-203: i64, wrapping; bconst: b_size device constants
int launch_cnv_by_const(pz_module* M, long long* res, int res_cols, int res_col, int min_size, int offset, const long long* a, int a_cols,
                        int a_size, int a_col, const long long* bconst, int b_size);

}  // namespace pz
