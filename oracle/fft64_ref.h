/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU restatement (plain C) of poulpy-cpu-ref's FFT64 family, the path
 * BASELINE.json's north_star names.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call this; poulpy_amd/ never does.
 *
 * PARITY STATUS: "parity unpinned" by reference fixtures.  The reference
 * (Rust, nightly toolchain) cannot be compiled in this image and its test
 * suite holds no golden vectors (all tests are differential).
 * The restatement is instead pinned by exact-arithmetic properties that the
 * reference's own cross-backend tests rely on (FFT64Ref == NTT120Ref on the
 * normalized i64 limbs, poulpy-cpu-ref/src/tests.rs:133-141): see
 * oracle/exact.py and tests/test_oracle_*.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Compile with -O2 -ffp-contract=off: Rust never contracts
 * a*b+c, and the reference butterflies are written unfused.
 *
 * Layout conventions (poulpy-hal/src/layouts/znx_base.rs:52-82):
 *   VecZnx / VecZnxDft / VecZnxBig (n, cols, size): limb j of column i starts
 *   at scalar offset n*(j*cols + i).  A DFT polynomial is [re(0..m) | im(0..m)],
 *   m = n/2 (reim/fft_ref.rs:25-27).
 */
#ifndef PZR_FFT64_REF_H
#define PZR_FFT64_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pzr_tables pzr_tables;

/* FFT64RefHandle { table_fft, table_ifft }  (poulpy-cpu-ref/src/fft64/module.rs:34-69) */
pzr_tables* pzr_tables_new(uint64_t n);
void pzr_tables_free(pzr_tables* t);
uint64_t pzr_tables_m(const pzr_tables* t);
const double* pzr_tables_omg_fft(const pzr_tables* t);  /* 2m doubles */
const double* pzr_tables_omg_ifft(const pzr_tables* t); /* 2m doubles */

/* reim/fft_ref.rs:25-43, reim/ifft_ref.rs:24-42; data = [re(m) | im(m)] in place */
void pzr_fft(const pzr_tables* t, double* data);
void pzr_ifft(const pzr_tables* t, double* data);

/* reim/conversion.rs:19-60 */
void pzr_reim_from_znx_i64(double* res, const int64_t* a, size_t len);
/* rounding-margin probe of the oracle itself (not in the reference; diagnostic, single-threaded) */
void pzr_margin_probe_set(int on);
double pzr_margin_probe_get(void);
void pzr_reim_to_znx_i64(int64_t* res, double divisor, const double* a, size_t len);
void pzr_reim_to_znx_i64_assign(double* res, double divisor, size_t len);

/* reference/fft64/vec_znx_dft.rs */
void pzr_vec_znx_dft_apply(const pzr_tables* t, size_t step, size_t offset,
                           double* res, size_t res_cols, size_t res_size, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_idft_apply(const pzr_tables* t,
                            int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                            const double* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_idft_apply_tmpa(const pzr_tables* t,
                                 int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                 double* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_idft_apply_consume(const pzr_tables* t, double* data, size_t cols, size_t size);
void pzr_vec_znx_dft_add_into(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* a, size_t a_cols, size_t a_size, size_t a_col,
                              const double* b, size_t b_cols, size_t b_size, size_t b_col);
void pzr_vec_znx_dft_sub(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                         const double* a, size_t a_cols, size_t a_size, size_t a_col,
                         const double* b, size_t b_cols, size_t b_size, size_t b_col);
void pzr_vec_znx_dft_add_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                const double* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_dft_add_scaled_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                       const double* a, size_t a_cols, size_t a_size, size_t a_col, int64_t a_scale);
void pzr_vec_znx_dft_sub_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                const double* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_dft_sub_negate_assign(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                       const double* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_dft_copy(size_t n, size_t step, size_t offset,
                          double* res, size_t res_cols, size_t res_size, size_t res_col,
                          const double* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_dft_zero(size_t n, double* res, size_t res_cols, size_t res_size, size_t res_col);

/* reference/fft64/svp.rs */
void pzr_svp_prepare(const pzr_tables* t, double* res, size_t res_cols, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_col);
void pzr_svp_apply_dft(const pzr_tables* t,
                       double* res, size_t res_cols, size_t res_size, size_t res_col,
                       const double* ppol, size_t a_cols, size_t a_col,
                       const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
void pzr_svp_apply_dft_to_dft(size_t n,
                              double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* ppol, size_t a_cols, size_t a_col,
                              const double* b, size_t b_cols, size_t b_size, size_t b_col);
void pzr_svp_apply_dft_to_dft_assign(size_t n,
                                     double* res, size_t res_cols, size_t res_size, size_t res_col,
                                     const double* ppol, size_t a_cols, size_t a_col);

/* reference/fft64/vmp.rs */
size_t pzr_vmp_prepare_tmp_bytes(size_t n);
size_t pzr_vmp_apply_dft_to_dft_tmp_bytes(size_t a_size, size_t prows, size_t pcols_in);
size_t pzr_vmp_apply_dft_tmp_bytes(size_t n, size_t a_size, size_t prows, size_t pcols_in);
void pzr_vmp_prepare(const pzr_tables* t, double* pmat, const int64_t* mat,
                     size_t rows, size_t cols_in, size_t cols_out, size_t size);
void pzr_vmp_apply_dft_to_dft(size_t n,
                              double* res, size_t res_cols, size_t res_size,
                              const double* a, size_t a_cols, size_t a_size,
                              const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size,
                              size_t limb_offset);
void pzr_vmp_apply_dft(const pzr_tables* t,
                       double* res, size_t res_cols, size_t res_size,
                       const int64_t* a, size_t a_cols, size_t a_size,
                       const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size);

/* reference/fft64/vec_znx_big.rs:122-138, 236-278 ; reference/vec_znx/normalize.rs */
void pzr_vec_znx_big_add_small_assign(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
size_t pzr_vec_znx_normalize_tmp_bytes(size_t n);
void pzr_vec_znx_normalize(size_t n,
                           int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col);

/* poulpy-core callers restated on top of the primitives above (rank = cols-1):
 *   external_product/glwe.rs:99-141,197-271 ; keyswitching/glwe.rs:53-109,207-239,298-380 */
void pzr_glwe_external_product(const pzr_tables* t, size_t rank,
                               int64_t* res, size_t res_size, size_t res_base2k,
                               const int64_t* a, size_t a_size, size_t a_base2k,
                               const double* ggsw_pmat, size_t dnum, size_t ggsw_size, size_t dsize, size_t ggsw_base2k);
void pzr_glwe_keyswitch(const pzr_tables* t, size_t rank_in, size_t rank_out,
                        int64_t* res, size_t res_size, size_t res_base2k,
                        const int64_t* a, size_t a_size, size_t a_base2k,
                        const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k);


/* reference/vec_znx/automorphism.rs:10-51 (= vec_znx_big_automorphism[_assign], fft64/vec_znx_big.rs:144-188) */
void pzr_vec_znx_automorphism(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                              const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_automorphism_assign(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);
void pzr_vec_znx_big_sub_small_assign(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_big_sub_small_negate_assign(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                             const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);

/* automorphism/glwe_ct.rs:51-275 */
enum { PZR_KS_PLAIN = 0, PZR_KS_AUTO = 1, PZR_KS_AUTO_ADD = 2, PZR_KS_AUTO_SUB = 3, PZR_KS_AUTO_SUB_NEGATE = 4 };
void pzr_glwe_automorphism(const pzr_tables* t, size_t rank, int mode, int64_t p,
                           int64_t* res, size_t res_size, size_t res_base2k,
                           const int64_t* a, size_t a_size, size_t a_base2k,
                           const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k);

/* i64 VecZnx limb-wise family: reference/vec_znx/add.rs, sub.rs, negate.rs, copy.rs (wrapping arithmetic).
 * assign_op mode: 0 res += a, 1 res -= a, 2 res = a - res (and -res beyond a's limbs); negate with a == NULL is in place */
void pzr_vec_znx_add_into(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                          const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                          const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
void pzr_vec_znx_sub(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                     const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
void pzr_vec_znx_assign_op(int mode, size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_negate(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                        const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_copy(size_t n, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);

/* reference/vec_znx/shift.rs:68-135 (lsh), :16-66 (lsh_assign), :245-342 (rsh), OVERWRITE = true */
void pzr_vec_znx_lsh(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_lsh_assign(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);
void pzr_vec_znx_rsh(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                     const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);

/* conversion/gglwe_to_ggsw.rs:116-268 (ggsw_expand_row); ggsw = MatZnx(dnum, cols, cols, size), keys[c] = tsk.at(c) */
void pzr_ggsw_expand_row(const pzr_tables* t, size_t rank, int64_t* ggsw, size_t dnum, size_t size, size_t base2k,
                         const double* const* keys, size_t key_dnum, size_t key_size, size_t dsize, size_t key_base2k);

/* conversion/gglwe_to_ggsw.rs:32-61 (ggsw_from_gglwe): copy of the a.at(row, 0) entries, then ggsw_expand_row */
void pzr_ggsw_from_gglwe(const pzr_tables* t, size_t rank, int64_t* ggsw, size_t dnum, size_t size, size_t base2k,
                         const int64_t* a, size_t a_cols_in, size_t a_size,
                         const double* const* keys, size_t key_dnum, size_t key_size, size_t dsize, size_t key_base2k);

/* reference/vec_znx/rotate.rs:10-36, mul_xp_minus_one.rs:23-37, normalize.rs:403-425 */
void pzr_vec_znx_rotate(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                        const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
void pzr_vec_znx_mul_xp_minus_one_assign(size_t n, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);
void pzr_vec_znx_normalize_assign(size_t n, size_t base2k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);

/* poulpy-bin-fhe/src/blind_rotation: key_prepared.rs:66-74 (x_pow_a table) and algorithms/cggi/algorithm.rs:265-440 */
void pzr_blind_rotation_x_pow_a(const pzr_tables* t, double* out /* 2n * n doubles */);
void pzr_blind_rotation_execute(const pzr_tables* t, size_t rank, size_t n_lwe, size_t block_size,
                                int64_t* res, size_t res_size, size_t base2k,
                                const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                const double* brk, size_t dnum, size_t brk_size, const double* x_pow_a);

/* algorithm.rs:121-273 (execute_block_binary_extended): ext accumulators, lut = ext polynomials VecZnx(1, lut_size), x_pow_a with
 * 2n entries (X^(2n) = X^0 is never used: :233) */
void pzr_blind_rotation_execute_extended(const pzr_tables* t, size_t rank, size_t n_lwe, size_t block_size, size_t ext,
                                         int64_t* res, size_t res_size, size_t base2k,
                                         const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                         const double* brk, size_t dnum, size_t brk_size, const double* x_pow_a);

/* reference/vec_znx/shift.rs:186-243 ; poulpy-core/src/glwe_trace.rs:129-176 (equal bases) */
void pzr_vec_znx_rsh_assign(size_t n, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);
void pzr_glwe_trace_assign(const pzr_tables* t, size_t rank, int64_t* res, size_t res_size, size_t base2k,
                           size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                           size_t dnum, size_t key_size, size_t dsize);

/* poulpy-bin-fhe/src/circuit_bootstrapping/circuit.rs:219-370, to_exponent = false, one base2k everywhere; the lookup
 * table and gap (:274-301, :333) are inputs */
void pzr_circuit_bootstrap_to_constant(const pzr_tables* t, size_t rank, size_t base2k,
                                       size_t n_lwe, size_t block_size, const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                       const double* brk, size_t brk_dnum, size_t brk_size, size_t glwe_size, const double* x_pow_a,
                                       size_t nsteps, const int64_t* gals, const double* const* atk, size_t atk_dnum, size_t atk_size,
                                       int64_t* ggsw, size_t res_dnum, size_t res_size, size_t gap,
                                       const double* const* tsk, size_t tsk_dnum, size_t tsk_size);

/* poulpy-core/src/glwe_packing.rs:15-87, :122-176 (one base2k, one size); slots: n pointers (NULL = absent), clobbered */
void pzr_glwe_pack(const pzr_tables* t, size_t rank, int64_t* res, int64_t** slots, size_t size, size_t base2k, size_t log_gap_out,
                   const int64_t* gals, const double* const* keys, size_t dnum, size_t key_size);

/* glwe_pack with the automorphism keys in their own base (test_suite/glwe_packing.rs:40-42); trace_size = limbs of the closing trace's
 * temporary, ceil(max(a.k, res.k) / key_base2k) (glwe_trace.rs:107-112) */
void pzr_glwe_pack_bases(const pzr_tables* t, size_t rank, int64_t* res, int64_t** slots, size_t size, size_t base2k, size_t key_base2k,
                         size_t trace_size, size_t log_gap_out, const int64_t* gals, const double* const* keys, size_t dnum, size_t key_size);
/* glwe_trace_assign with res in another base than the keys (glwe_trace.rs:153-163); conv_size = ceil(res.max_k / key_base2k) */
void pzr_glwe_trace_assign_bases(const pzr_tables* t, size_t rank, int64_t* res, size_t res_size, size_t res_base2k, size_t conv_size,
                                 size_t key_base2k, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                                 size_t dnum, size_t key_size, size_t dsize);

/* circuit.rs:219-370 with to_exponent = true (+ post_process :373-421), one base2k, res_size <= glwe_size */
void pzr_circuit_bootstrap_to_exponent(const pzr_tables* t, size_t rank, size_t base2k,
                                       size_t n_lwe, size_t block_size, const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                       const double* brk, size_t brk_dnum, size_t brk_size, size_t glwe_size, const double* x_pow_a,
                                       const int64_t* gals, const double* const* atk, size_t atk_dnum, size_t atk_size,
                                       int64_t* ggsw, size_t res_dnum, size_t res_size, size_t gap,
                                       size_t log_gap_in, size_t log_gap_out, size_t log_domain,
                                       const double* const* tsk, size_t tsk_dnum, size_t tsk_size);

/* circuit.rs:219-421 with one base2k per object (bases = {brk, atk, tsk, res}), the way the reference's tests run it
 * (circuit_bootstrapping/tests/circuit_bootstrapping.rs:49-53); glwe_size = limbs of the rotation in the brk base, atk_glwe_size =
 * ceil(brk.max_k / atk base), trace_size = ceil(max(brk.max_k, res.max_k) / atk base) (glwe_trace.rs:107-112).  to_exponent = 0: gals / atk
 * hold the nsteps steps of the full trace; 1: all log2(n) steps, gaps / domain as in pzr_circuit_bootstrap_to_exponent */
void pzr_circuit_bootstrap_bases(const pzr_tables* t, size_t rank, const size_t* bases, int to_exponent,
                                 size_t n_lwe, size_t block_size, const int64_t* lwe_2n, const int64_t* lut, size_t lut_size,
                                 const double* brk, size_t brk_dnum, size_t brk_size, size_t glwe_size, size_t atk_glwe_size, size_t trace_size,
                                 const double* x_pow_a, size_t nsteps, const int64_t* gals, const double* const* atk, size_t atk_dnum,
                                 size_t atk_size, int64_t* ggsw, size_t res_dnum, size_t res_size, size_t gap,
                                 size_t log_gap_in, size_t log_gap_out, size_t log_domain,
                                 const double* const* tsk, size_t tsk_dnum, size_t tsk_size);

/* reference/fft64/convolution.rs (HalImpl cnv_*, poulpy-hal/src/oep/hal_impl.rs:670-754).  CnvPVecL / CnvPVecR bytes (FFT64):
 * [col][blk < m/4][limb < size][re x4 | im x4]; a / b below are such buffers of (cols, a_size) / (cols, b_size). */
size_t pzr_cnv_prepare_tmp_bytes(size_t n, size_t res_size, size_t a_size);
void pzr_cnv_prepare(const pzr_tables* t, double* res, size_t res_cols, size_t res_size,
                     const int64_t* a, size_t a_cols, size_t a_size, int64_t mask);
void pzr_cnv_prepare_self(const pzr_tables* t, double* left, double* right, size_t cols, size_t size,
                          const int64_t* a, size_t a_cols, size_t a_size, int64_t mask);
size_t pzr_cnv_apply_dft_tmp_bytes(size_t res_size, size_t a_size, size_t b_size);
size_t pzr_cnv_pairwise_apply_dft_tmp_bytes(size_t res_size, size_t a_size, size_t b_size);
size_t pzr_cnv_by_const_apply_tmp_bytes(size_t res_size, size_t a_size, size_t b_size);
void pzr_cnv_apply_dft(size_t n, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                       const double* a, size_t a_size, size_t a_col, const double* b, size_t b_size, size_t b_col);
void pzr_cnv_pairwise_apply_dft(size_t n, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                const double* a, size_t a_size, const double* b, size_t b_size, size_t col_i, size_t col_j);
void pzr_cnv_by_const_apply(size_t n, size_t cnv_offset, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                            const int64_t* a, size_t a_cols, size_t a_size, size_t a_col, const int64_t* b, size_t b_size);

/* poulpy-core/src/operations/glwe.rs: msb_mask_bottom_limb :921-926; glwe_tensor_apply :700-807 / _add_assign :809-913;
 * glwe_tensor_square_apply :609-698; glwe_tensor_relinearize :541-607.  GLWETensor data = VecZnx((rank+1)(rank+2)/2, size). */
int64_t pzr_msb_mask_bottom_limb(size_t base2k, size_t k);
void pzr_glwe_tensor_apply(const pzr_tables* t, size_t rank, size_t cnv_offset, int add_assign,
                           int64_t* res, size_t res_size, size_t res_base2k,
                           const int64_t* a, size_t a_size, size_t a_effective_k,
                           const int64_t* b, size_t b_size, size_t b_effective_k, size_t ab_base2k);
void pzr_glwe_tensor_square_apply(const pzr_tables* t, size_t rank, size_t cnv_offset,
                                  int64_t* res, size_t res_size, size_t res_base2k,
                                  const int64_t* a, size_t a_size, size_t a_effective_k, size_t a_base2k);
void pzr_glwe_tensor_relinearize(const pzr_tables* t, size_t rank,
                                 int64_t* res, size_t res_size, size_t res_base2k,
                                 const int64_t* a, size_t a_size, size_t a_base2k,
                                 const double* tsk_pmat, size_t dnum, size_t tsk_size, size_t dsize, size_t key_base2k);

/* LWE glue of the gate bootstrap: poulpy-bin-fhe blind_rotation/algorithms/mod.rs:136-176 (mod_switch_2n), poulpy-core
 * api/conversion.rs:15-40 (lwe_sample_extract), keyswitching/lwe.rs:49-94 (lwe_keyswitch), conversion/lwe_to_glwe.rs:46-121
 * (glwe_from_lwe), conversion/glwe_to_lwe.rs:42-90 (lwe_from_glwe).  LWE = VecZnx(n_lwe + 1, 1 column, size): [b, a_0..] per limb. */
void pzr_mod_switch_2n(size_t n2, int64_t* res, const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t base2k, int negate);
void pzr_lwe_sample_extract(size_t n, int64_t* res, size_t res_n_lwe, size_t res_size, const int64_t* a, size_t a_cols, size_t a_size);
void pzr_lwe_keyswitch(const pzr_tables* t, int64_t* res, size_t res_n_lwe, size_t res_size, size_t res_base2k,
                       const int64_t* a, size_t a_n_lwe, size_t a_size, size_t a_base2k,
                       const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k);
void pzr_glwe_from_lwe(const pzr_tables* t, size_t rank_out, int64_t* res, size_t res_size, size_t res_base2k,
                       const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t lwe_base2k, size_t glwe_size,
                       const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k);
void pzr_lwe_from_glwe(const pzr_tables* t, size_t rank_in, int64_t* res, size_t res_n_lwe, size_t res_size, size_t res_base2k,
                       const int64_t* a, size_t a_size, size_t a_base2k, size_t a_idx,
                       const double* key_pmat, size_t dnum, size_t key_size, size_t dsize, size_t key_base2k);

#ifdef __cplusplus
}
#endif
#endif
