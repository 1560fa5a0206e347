/*
 * poulpy_hip.h — C ABI of libpoulpy_hip.so, the MI355X (gfx950) FFT64 backend
 * for poulpy-hal.
 *
 * Every entry point below is what a `poulpy-hip-mi355x` Rust crate binds with
 * `extern "C"` to implement `unsafe trait HalImpl<FFT64Hip>`
 * (poulpy-hal/src/oep/hal_impl.rs:25) for the FFT64 hot path; the reference
 * method each function replaces is cited as hal_impl.rs:<line> (all paths
 * relative to /root/reference).  INTEGRATION.md shows the Rust side.
 *
 * Conventions
 *  - Plain pointers + sizes only.  A VecZnx / VecZnxDft / VecZnxBig is passed as
 *    (ptr, cols, size): limb j of column i starts at scalar offset
 *    n*(j*cols + i) (poulpy-hal/src/layouts/znx_base.rs:52-82); n comes from the
 *    module.  `size` is the *current* size (VecZnx::size(), after set_size).
 *  - Pointers may be host memory (pageable, or pinned from pz_alloc_bytes) or
 *    device memory (pz_device_alloc): the library detects which
 *    (hipPointerGetAttributes).  Host buffers are staged through the module's
 *    device workspace (H2D, kernels, D2H) and the call returns only when the
 *    results are visible in the caller's buffer — "logically synchronous"
 *    (poulpy-hal/docs/backend_safety_contract.md:16-19).  Device buffers are
 *    processed in place on the module stream; the call returns after enqueue
 *    and pz_module_sync() (or any host-pointer call) drains the stream.  The
 *    module stream is NOT ordered with any other stream (it is non-blocking with
 *    respect to the null stream too): a device buffer written by the caller's own
 *    kernels or copies on another stream must be complete (event or stream sync)
 *    before it is passed in, and must stay alive until pz_module_sync().
 *  - The bytes of VecZnxDft / SvpPPol / VmpPMat (ScalarPrep = f64) are
 *    backend-private ("device order"): same byte sizes as the
 *    reference (n*cols*size*8, module.rs:51-65) but NOT the reference's
 *    [re | im] bit-reversed layout; callers must treat them as opaque, which is
 *    what poulpy-core does.
 *  - Return value: 0 on success, negative pz_status on failure.  Nothing throws
 *    across the ABI; pz_last_error() gives a thread-local message.  The Rust
 *    shim panics on non-zero, matching the reference's assert!/panic! behaviour.
 *  - All *_batched entry points take DEVICE pointers only and treat `batch`
 *    independent objects laid out back to back (object b at ptr + b*object_len).
 *    Exception: the four GLWE-level calls pz_glwe_external_product_batched, pz_glwe_keyswitch_batched,
 *    pz_glwe_automorphism_batched and pz_glwe_tensor_relinearize_batched also accept HOST containers for res / a (staged, the
 *    call is then logically synchronous; when both are PINNED and batch >= 2 the first three run upload, kernels and download
 *    of successive slices on three streams at once) and a HOST-resident prepared key: poulpy-hal's buffers are host-addressable by contract
 *    (Backend::OwnedBuf: DataMut), so this is what the Rust shim's CoreImpl overrides pass.  A host key is mirrored on the
 *    device on first use and re-used afterwards; the mirror is validated on every call by a sampled fingerprint of the host
 *    bytes and dropped by pz_vmp_prepare / pz_vmp_zero on that buffer and by pz_free_bytes of its block.  A caller that modifies a prepared matrix in place by
 *    other means (e.g. deserializes into it) must call pz_module_forget_host_key before the next use.
 */
#ifndef POULPY_HIP_H
#define POULPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pz_module pz_module;

/* the ABI revision this header describes (pz_abi_version() of a matching library returns it) */
#define PZ_ABI_VERSION 4u

typedef enum {
    PZ_OK = 0,
    PZ_ERR_INVALID = -1,     /* shape / argument violation (reference: assert!/debug_assert!) */
    PZ_ERR_UNSUPPORTED = -2, /* e.g. n < 32 */
    PZ_ERR_HIP = -3,         /* HIP runtime failure (no device, OOM, launch error) */
    PZ_ERR_ALIAS = -4,       /* overlapping arguments that the op cannot honour */
    PZ_ERR_RCCL = -5
} pz_status;

const char* pz_last_error(void);
/* library/ABI version, bumped on any signature change.  PZ_ABI_VERSION is the version this header describes: clients compare it
 * with pz_abi_version() of the library they loaded before the first call (poulpy_hip.hpp's Module, hal.py and the Rust handle do) */
uint32_t pz_abi_version(void);

/* ---- module (Backend + HalImpl::new) ------------------------------------ */
/* HalImpl::new, hal_impl.rs:320 ; FFT64RefHandle, poulpy-cpu-ref/src/fft64/module.rs:34-69 */
int pz_module_new(uint64_t n, pz_module** out);
int pz_module_new_on_device(uint64_t n, int device, pz_module** out);
/* Backend::destroy, poulpy-hal/src/layouts/module.rs:74-82,260-266 */
void pz_module_free(pz_module* m);
uint64_t pz_module_n(const pz_module* m);
int pz_module_device(const pz_module* m);
/* drains the module stream (needed only after device-pointer calls) */
/* A sibling of `m` for another host thread.  Calls on ONE module are serialised by its lock (one HIP stream, one workspace); poulpy
 * callers share `&Module` across scoped threads (poulpy-bin-fhe bdd_arithmetic/eval.rs:210-221), so the shim gives every other thread a
 * sibling: it shares m's immutable device tables (reference-counted: free in any order) and owns its stream, workspaces, staging arena,
 * pinned-key list, key mirrors, graph cache and lock — calls on different siblings overlap on the device (copies of one with kernels of
 * another).  Knobs (fusion, chunk, graphs) are copied at clone time. */
int pz_module_clone(pz_module* m, pz_module** out);
int pz_module_sync(pz_module* m);
/* raw hipStream_t of the module, for callers that want to order their own work */
void* pz_module_stream(pz_module* m);

/* Backend::alloc_bytes, module.rs:38-43 — pinned host memory, 64 B aligned (lib.rs:146) */
void* pz_alloc_bytes(size_t len);
/* also drops, in every live module, the device mirror of a host-resident prepared key that lies inside the block (see "host
 * containers" below) — a prepared key's `Drop` needs no other hook */
void pz_free_bytes(void* p);
/* device-resident buffers for the batched path */
int pz_device_alloc(pz_module* m, size_t len, void** out);
int pz_device_free(pz_module* m, void* p);
int pz_memcpy_h2d(pz_module* m, void* dst_dev, const void* src_host, size_t len);
int pz_memcpy_d2h(pz_module* m, void* dst_host, const void* src_dev, size_t len);
int pz_memset_d(pz_module* m, void* dst_dev, int value, size_t len);

/* Backend::bytes_of_*, module.rs:51-65 */
size_t pz_bytes_of_vec_znx(uint64_t n, size_t cols, size_t size);
size_t pz_bytes_of_vec_znx_dft(uint64_t n, size_t cols, size_t size);
size_t pz_bytes_of_vec_znx_big(uint64_t n, size_t cols, size_t size);
size_t pz_bytes_of_svp_ppol(uint64_t n, size_t cols);
size_t pz_bytes_of_vmp_pmat(uint64_t n, size_t rows, size_t cols_in, size_t cols_out, size_t size);

/* ---- VecZnxDft ----------------------------------------------------------- */
/* hal_impl.rs:529 vec_znx_dft_apply */
int pz_vec_znx_dft_apply(pz_module* m, size_t step, size_t offset,
                         double* res, size_t res_cols, size_t res_size, size_t res_col,
                         const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
/* hal_impl.rs:534 */
size_t pz_vec_znx_idft_apply_tmp_bytes(const pz_module* m);
/* hal_impl.rs:536 vec_znx_idft_apply */
int pz_vec_znx_idft_apply(pz_module* m,
                          int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                          const double* a, size_t a_cols, size_t a_size, size_t a_col);
/* hal_impl.rs:541 vec_znx_idft_apply_tmpa (a may be clobbered) */
int pz_vec_znx_idft_apply_tmpa(pz_module* m,
                               int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                               double* a, size_t a_cols, size_t a_size, size_t a_col);
/* hal_impl.rs:546 vec_znx_idft_apply_consume: all cols x limbs in place, buffer re-typed to VecZnxBig */
int pz_vec_znx_idft_apply_consume(pz_module* m, void* data, size_t cols, size_t size);
/* hal_impl.rs:553 */
int pz_vec_znx_dft_add_into(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                            const double* a, size_t a_cols, size_t a_size, size_t a_col,
                            const double* b, size_t b_cols, size_t b_size, size_t b_col);
/* hal_impl.rs:559 */
int pz_vec_znx_dft_add_scaled_assign(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                     const double* a, size_t a_cols, size_t a_size, size_t a_col, int64_t a_scale);
/* hal_impl.rs:564 */
int pz_vec_znx_dft_add_assign(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* a, size_t a_cols, size_t a_size, size_t a_col);
/* hal_impl.rs:569 */
int pz_vec_znx_dft_sub(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                       const double* a, size_t a_cols, size_t a_size, size_t a_col,
                       const double* b, size_t b_cols, size_t b_size, size_t b_col);
/* hal_impl.rs:575 */
int pz_vec_znx_dft_sub_assign(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* a, size_t a_cols, size_t a_size, size_t a_col);
/* hal_impl.rs:580 */
int pz_vec_znx_dft_sub_negate_assign(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                     const double* a, size_t a_cols, size_t a_size, size_t a_col);
/* hal_impl.rs:585 */
int pz_vec_znx_dft_copy(pz_module* m, size_t step, size_t offset,
                        double* res, size_t res_cols, size_t res_size, size_t res_col,
                        const double* a, size_t a_cols, size_t a_size, size_t a_col);
/* hal_impl.rs:590 */
int pz_vec_znx_dft_zero(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col);

/* ---- SVP ------------------------------------------------------------------ */
/* hal_impl.rs:595 svp_prepare (ScalarZnx(n, a_cols) -> SvpPPol(n, res_cols)) */
int pz_svp_prepare(pz_module* m, double* res, size_t res_cols, size_t res_col,
                   const int64_t* a, size_t a_cols, size_t a_col);
/* hal_impl.rs:600 svp_apply_dft */
int pz_svp_apply_dft(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                     const double* ppol, size_t a_cols, size_t a_col,
                     const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
/* hal_impl.rs:606 svp_apply_dft_to_dft */
int pz_svp_apply_dft_to_dft(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                            const double* ppol, size_t a_cols, size_t a_col,
                            const double* b, size_t b_cols, size_t b_size, size_t b_col);
/* hal_impl.rs:612 svp_apply_dft_to_dft_assign */
int pz_svp_apply_dft_to_dft_assign(pz_module* m, double* res, size_t res_cols, size_t res_size, size_t res_col,
                                   const double* ppol, size_t a_cols, size_t a_col);

/* ---- VMP ------------------------------------------------------------------ */
/* hal_impl.rs:618 ; the reference's scratch sizes are returned unchanged so that
 * poulpy-core sizes its host arena identically — the device path
 * does not use caller scratch. */
size_t pz_vmp_prepare_tmp_bytes(const pz_module* m, size_t rows, size_t cols_in, size_t cols_out, size_t size);
/* hal_impl.rs:620 vmp_prepare (MatZnx -> VmpPMat, device order) */
int pz_vmp_prepare(pz_module* m, double* pmat, const int64_t* mat,
                   size_t rows, size_t cols_in, size_t cols_out, size_t size);
/* hal_impl.rs:626 */
size_t pz_vmp_apply_dft_tmp_bytes(const pz_module* m, size_t res_size, size_t a_size,
                                  size_t b_rows, size_t b_cols_in, size_t b_cols_out, size_t b_size);
/* hal_impl.rs:636 vmp_apply_dft */
int pz_vmp_apply_dft(pz_module* m, double* res, size_t res_cols, size_t res_size,
                     const int64_t* a, size_t a_cols, size_t a_size,
                     const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size);
/* hal_impl.rs:643 */
size_t pz_vmp_apply_dft_to_dft_tmp_bytes(const pz_module* m, size_t res_size, size_t a_size,
                                         size_t b_rows, size_t b_cols_in, size_t b_cols_out, size_t b_size);
/* hal_impl.rs:653 vmp_apply_dft_to_dft.  DEVIATION for limb_offset > 0: cpu-ref's FFT64 core clamps the key columns it reads to
 * res_size and leaves the last limb_offset limbs of res unwritten (stale scratch, vmp.rs:217-263); this backend follows the NTT120
 * sibling's semantics instead (reference/ntt120/vmp.rs:190,281-287): res[c] = sum_r a[r] * P[r][c + off] for every c with a key
 * column, zero beyond — every limb of res is written.  Byte parity with FFT64Ref is therefore claimed for
 * limb_offset = 0 (dsize = 1: every BASELINE config) only; limb_offset > 0 (the dsize > 1 callers) is validated against the exact
 * bivariate product. */
int pz_vmp_apply_dft_to_dft(pz_module* m, double* res, size_t res_cols, size_t res_size,
                            const double* a, size_t a_cols, size_t a_size,
                            const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size,
                            size_t limb_offset);
/* hal_impl.rs:665 vmp_zero */
int pz_vmp_zero(pz_module* m, double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size);

/* ---- VecZnxBig ------------------------------------------------------------- */
/* hal_impl.rs:428 */
size_t pz_vec_znx_big_normalize_tmp_bytes(const pz_module* m);
/* hal_impl.rs:431 vec_znx_big_normalize (same- and cross-base2k, any res_offset) */
int pz_vec_znx_big_normalize(pz_module* m,
                             int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset, size_t res_col,
                             const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col);
/* hal_impl.rs:362 vec_znx_big_add_small_assign */
int pz_vec_znx_big_add_small_assign(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                    const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);

/* hal_impl.rs:225 vec_znx_rotate, :230 _assign_tmp_bytes, :232 _assign (reference/znx/rotate.rs:3-27: res = X^k * a in Z[X]/(X^n+1),
 * any k; limbs of res beyond a.size zeroed).  Used by glwe_rotate_assign between the rows of a circuit bootstrapping. */
size_t pz_vec_znx_rotate_assign_tmp_bytes(const pz_module* m);
int pz_vec_znx_rotate(pz_module* m, int64_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_rotate_assign(pz_module* m, int64_t k, int64_t* res, size_t cols, size_t size, size_t col);
/* hal_impl.rs:217 vec_znx_rsh_assign (reference/vec_znx/shift.rs:186-243): res >>= k bits in place, normalizing; used by
 * glwe_rsh / glwe_trace.  k <= base2k * size. */
size_t pz_vec_znx_rsh_tmp_bytes(const pz_module* m);
int pz_vec_znx_rsh_assign(pz_module* m, size_t base2k, size_t k, int64_t* res, size_t cols, size_t size, size_t col);

/* ---- i64 VecZnx limb-wise family (hal_impl.rs:34 zero, :41 normalize, :55 normalize_assign, :59 add_into,
 * :65 add_assign, :90 sub, :96 sub_assign, :101 sub_negate_assign, :126 negate, :131 negate_assign, :289 copy): the i64
 * operations poulpy-core runs between the hot-path calls (glwe_add / sub / copy / normalize ...), so that ciphertexts can
 * stay on the device.  Limb-range rules of reference/vec_znx/{add,sub,negate,copy}.rs (common limbs combined, the longer
 * operand copied / negated, the rest of res zeroed), wrapping i64 arithmetic.  Host or device pointers. ------------------- */
int pz_vec_znx_add_into(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                        const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                        const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
int pz_vec_znx_sub(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                   const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                   const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
int pz_vec_znx_add_assign(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                          const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_sub_assign(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                          const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_sub_negate_assign(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                 const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_negate(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                      const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_negate_assign(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);
int pz_vec_znx_copy(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                    const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_zero(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);
/* vec_znx_normalize: the function vec_znx_big_normalize forwards to when ScalarBig = i64; _assign: in place, same base
 * (reference/vec_znx/normalize.rs:18-48, :403-425) */
size_t pz_vec_znx_normalize_tmp_bytes(const pz_module* m);
int pz_vec_znx_normalize(pz_module* m, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                         size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col);
int pz_vec_znx_normalize_assign(pz_module* m, size_t base2k, int64_t* res, size_t cols, size_t size, size_t col);
/* vec_znx_lsh (hal_impl.rs:165), vec_znx_rsh (:137), vec_znx_lsh_assign (:221), tmp bytes (:163): bit shifts of the torus value
 * by k bits with renormalization (reference/vec_znx/shift.rs:68-135, :245-342, :16-66); at equal bases these are the limb
 * walks of vec_znx_normalize with res_offset = +k / -k.  (vec_znx_rsh_assign: below.) */
size_t pz_vec_znx_lsh_tmp_bytes(const pz_module* m);
int pz_vec_znx_lsh(pz_module* m, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                   const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_rsh(pz_module* m, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                   const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_lsh_assign(pz_module* m, size_t base2k, size_t k, int64_t* res, size_t cols, size_t size, size_t col);

/* ---- X -> X^p on i64 containers (the glwe_automorphism callers) ---------------------- *
 * hal_impl.rs:236 vec_znx_automorphism, :241 _assign_tmp_bytes, :243 _assign; :517 vec_znx_big_automorphism, :522, :524.
 * reference/znx/automorphism.rs:1-17: res[(i*p) mod 2n] = a[i], negated when the index wraps past n; limbs of res beyond
 * a.size are zeroed (vec_znx/automorphism.rs:32-34).  p may be negative; it must be odd (an even p is not a ring
 * automorphism: PZ_ERR_INVALID).  The non-assign forms reject res == a.                                           */
size_t pz_vec_znx_automorphism_assign_tmp_bytes(const pz_module* m);
int pz_vec_znx_automorphism(pz_module* m, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                            const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_automorphism_assign(pz_module* m, int64_t p, int64_t* res, size_t cols, size_t size, size_t col);
size_t pz_vec_znx_big_automorphism_assign_tmp_bytes(const pz_module* m);
int pz_vec_znx_big_automorphism(pz_module* m, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_big_automorphism_assign(pz_module* m, int64_t p, int64_t* res, size_t cols, size_t size, size_t col);

/* ---- batched, device-resident path (the measured one) ---------------------- *
 * Coarser boundary: unsafe trait CoreImpl<BE>, poulpy-core/src/oep/core_impl.rs:34
 *   glwe_external_product :120  (poulpy-core/src/external_product/glwe.rs:99-141,197-271)
 *   glwe_keyswitch        :42   (poulpy-core/src/keyswitching/glwe.rs:53-109,207-239,298-380)
 * applied to `batch` independent GLWE ciphertexts that share one prepared key.
 * All pointers are device pointers; ciphertext b of `a` starts at
 * a + b*n*(rank+1)*a_size, of `res` at res + b*n*(rank+1)*res_size.            */
typedef struct {
    uint64_t rank;        /* GLWE rank; cols = rank + 1 */
    uint64_t dnum;        /* rows of the prepared GGSW / GGLWE */
    uint64_t dsize;       /* digit size (1 on every BASELINE config) */
    uint64_t key_size;    /* limbs of the prepared key */
    uint64_t key_base2k;
    uint64_t a_size;
    uint64_t a_base2k;
    uint64_t res_size;
    uint64_t res_base2k;
    uint64_t rank_out;    /* keyswitch only: output rank (key cols_out = rank_out+1, cols_in = rank) */
} pz_glwe_op_params;

int pz_glwe_external_product_batched(pz_module* m, int64_t* res, const int64_t* a, const double* ggsw_pmat,
                                     const pz_glwe_op_params* p, size_t batch);
int pz_glwe_keyswitch_batched(pz_module* m, int64_t* res, const int64_t* a, const double* key_pmat,
                              const pz_glwe_op_params* p, size_t batch);
/* CoreImpl glwe_automorphism family (poulpy-core/src/automorphism/glwe_ct.rs) on `batch` ciphertexts sharing one
 * prepared automorphism key (a GGLWE rank -> rank; `gal` = key.p(), odd):
 *   PZ_AUTO             res = phi(keyswitch(a))                      glwe_ct.rs:51-72
 *   PZ_AUTO_ADD         res = normalize(phi(big) + a)                :96-140
 *   PZ_AUTO_SUB         res = normalize(phi(big) - a)                :185-229
 *   PZ_AUTO_SUB_NEGATE  res = normalize(a - phi(big))                :231-275
 * with big = the key-switch value before normalization (keyswitching/glwe.rs:207-239) and phi = X -> X^gal.
 * The *_assign forms of the reference are the same calls with res == a (same layout), which is allowed: every
 * ciphertext is fully consumed before its result is written.                                                    */
enum { PZ_AUTO = 0, PZ_AUTO_ADD = 1, PZ_AUTO_SUB = 2, PZ_AUTO_SUB_NEGATE = 3 };
int pz_glwe_automorphism_batched(pz_module* m, int64_t* res, const int64_t* a, const double* key_pmat,
                                 const pz_glwe_op_params* p, int64_t gal, int mode, size_t batch);
/* CoreImpl glwe_trace_assign (poulpy-core/src/glwe_trace.rs:129-176) on `batch` ciphertexts:
 * for s < nsteps:  res = rsh(res, 1 bit);  res = glwe_automorphism_add_assign(res, key_s)   (:164-174).
 * The caller resolves the steps skip..log_n into Galois elements (i = 0: -1, else galois_element(2^(i-1)),
 * poulpy-hal/src/layouts/module.rs:214-226) and the matching prepared automorphism keys: gals[s], key_pmats[s] are HOST
 * arrays; each key_pmats[s] and res are device pointers.  res and keys of one base2k: p describes one step (a_size = res_size,
 * equal base2k).  res in another base than the keys (:153-163, the case test_suite/trace.rs runs): (res_size, res_base2k) is res,
 * (a_size, a_base2k = key_base2k) its layout re-expressed in the keys' base, a_size = ceil(res.max_k / key_base2k); res is
 * normalized into a temporary of that layout, traced there and normalized back. */
int pz_glwe_trace_batched(pz_module* m, int64_t* res, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                          const pz_glwe_op_params* p, size_t batch);
/* GLWEPacking::glwe_pack (poulpy-core/src/glwe_packing.rs:122-176, pack_internal :15-87) on `batch` independent packing
 * problems that share the occupancy pattern; ciphertexts, keys and result share base2k and size (p: a_size = res_size).
 *   indices / cts   HOST arrays of nslots entries: cts[s] -> the `batch` contiguous device GLWEs of index indices[s] (the
 *                   reference's HashMap<usize, &mut GLWE>); they are CLOBBERED, as the reference's entries are
 *   gals / key_pmats HOST arrays of log2(n) entries: Galois element and prepared automorphism key of trace step i
 *                   (i = 0: -1, else galois_element(2^(i-1)); glwe_pack_galois_elements :100-102)
 *   res             batch contiguous GLWEs: the packed result after the final partial trace (:175)
 *   tmp             device scratch of pz_glwe_pack_tmp_bytes */
size_t pz_glwe_pack_tmp_bytes(const pz_module* m, const pz_glwe_op_params* p, size_t batch);
int pz_glwe_pack_batched(pz_module* m, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts,
                         size_t log_gap_out, const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p,
                         void* tmp, size_t tmp_bytes, size_t batch);
/* the same with the automorphism keys in their own base (the case test_suite/glwe_packing.rs:40-42 runs: ciphertexts and result in
 * base2k - 1, keys in base2k): p->a = p->res = the ciphertexts' layout, p->key_base2k the keys'; trace_size = limbs of the closing
 * trace's temporary in the keys' base, ceil(max(a.max_k, res.max_k) / key_base2k) (glwe_trace.rs:107-112) */
size_t pz_glwe_pack_bases_tmp_bytes(const pz_module* m, const pz_glwe_op_params* p, size_t trace_size, size_t batch);
int pz_glwe_pack_bases_batched(pz_module* m, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts,
                               size_t log_gap_out, const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p,
                               size_t trace_size, void* tmp, size_t tmp_bytes, size_t batch);
/* CoreImpl ggsw_external_product (poulpy-core/src/external_product/ggsw.rs:54-58): res[row][col] = a[row][col] (x) ggsw
 * for the a_dnum * (rank+1) GLWE entries of the GGSW `a` (MatZnx layout: entries are contiguous), device pointers. */
int pz_ggsw_external_product(pz_module* m, int64_t* res, const int64_t* a, size_t a_dnum, const double* ggsw_pmat,
                             const pz_glwe_op_params* p);
/* CoreImpl ggsw_expand_row (poulpy-core/src/conversion/gglwe_to_ggsw.rs:116-268; the second half of ggsw_from_gglwe
 * :32-61 and of circuit bootstrapping) on `count` contiguous GGSWs (MatZnx layout, rows = dnum, cols_in = cols_out =
 * rank+1, size = p->res_size), in place: entry (row, col), col >= 1, becomes the key switch of the mask of entry (row, 0)
 * by tsk_pmat[col-1] (= tsk.at(col-1), a prepared GGLWE rank -> rank) with the body of entry (row, 0) added to column
 * `col`; entries (row, 0) are not written.  tsk_pmat is a HOST array of rank device pointers; p describes the key switch
 * (a_size = res_size, a_base2k = res_base2k = the GGSW's; rank_out = rank). */
int pz_ggsw_expand_row_batched(pz_module* m, int64_t* ggsw, size_t dnum, const double* const* tsk_pmat,
                               const pz_glwe_op_params* p, size_t count);
/* CoreImpl ggsw_from_gglwe (poulpy-core/src/conversion/gglwe_to_ggsw.rs:32-61) on `count` contiguous device GGLWEs `a`
 * (MatZnx layout, rows = dnum, cols_in = a_cols_in, cols_out = rank+1, size = p->res_size: same size and base as the
 * GGSW) -> `count` contiguous GGSWs: entries (row, 0) are copied from a.at(row, 0), then pz_ggsw_expand_row_batched. */
int pz_ggsw_from_gglwe_batched(pz_module* m, int64_t* ggsw, const int64_t* a, size_t a_cols_in, size_t dnum,
                               const double* const* tsk_pmat, const pz_glwe_op_params* p, size_t count);
/* ---- convolution family (BASELINE configs[4], CKKS tensoring) ----------------------------- *
 * Bivariate convolution over Z[X, Y]/(X^N + 1), Y = 2^-base2k (poulpy-hal/src/api/convolution.rs; reference
 * poulpy-cpu-ref/src/reference/fft64/convolution.rs).  CnvPVecL / CnvPVecR (ScalarPrep = f64) are opaque prepared operands of
 * n * cols * size scalars (module.rs:66-73), passed as (ptr, cols, size); in this backend polynomial (col, limb) is its spectrum
 * in device order at (col*size + limb)*n.  Host or device pointers, like every per-op entry point.
 * cnv_apply_dft: res limb k (k < min(res_size, a_size + b_size - 1)) = sum_j a[k + offset - j] * b[j], offset = min(cnv_offset,
 * a_size + b_size - 1); the other limbs of column res_col are zeroed.  NOTE: the reference's FFT64 implementation stores the
 * product at the raw start of `res` (convolution.rs:232, :251), i.e. it is only meaningful for a one-column res with
 * res_col = 0, which is what every caller passes; this backend writes column res_col of a res with any number of columns. */
/* hal_impl.rs:670 */
size_t pz_cnv_prepare_left_tmp_bytes(const pz_module* m, size_t res_size, size_t a_size);
/* hal_impl.rs:672 cnv_prepare_left: DFT of every limb of every column; `mask` is ANDed into the coefficients of the last active
 * limb min(res_size, a_size) - 1 (convolution.rs:56-61); limbs beyond it are zero */
int pz_cnv_prepare_left(pz_module* m, double* res, size_t res_cols, size_t res_size,
                        const int64_t* a, size_t a_cols, size_t a_size, int64_t mask);
/* hal_impl.rs:677, :679 */
size_t pz_cnv_prepare_right_tmp_bytes(const pz_module* m, size_t res_size, size_t a_size);
int pz_cnv_prepare_right(pz_module* m, double* res, size_t res_cols, size_t res_size,
                         const int64_t* a, size_t a_cols, size_t a_size, int64_t mask);
/* hal_impl.rs:684, :686 */
size_t pz_cnv_apply_dft_tmp_bytes(const pz_module* m, size_t cnv_offset, size_t res_size, size_t a_size, size_t b_size);
size_t pz_cnv_by_const_apply_tmp_bytes(const pz_module* m, size_t cnv_offset, size_t res_size, size_t a_size, size_t b_size);
/* hal_impl.rs:695 cnv_by_const_apply: i64 domain, res (VecZnxBig) limb k = sum_j a[k + offset - j] * b[j] (wrapping), b = b_len constants */
int pz_cnv_by_const_apply(pz_module* m, size_t cnv_offset, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                          const int64_t* a, size_t a_cols, size_t a_size, size_t a_col, const int64_t* b, size_t b_len);
/* hal_impl.rs:709 cnv_apply_dft */
int pz_cnv_apply_dft(pz_module* m, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                     const double* a, size_t a_cols, size_t a_size, size_t a_col,
                     const double* b, size_t b_cols, size_t b_size, size_t b_col);
/* hal_impl.rs:724, :733 cnv_pairwise_apply_dft: (a[i] + a[j]) * (b[i] + b[j]); i == j: a[i] * b[i] */
size_t pz_cnv_pairwise_apply_dft_tmp_bytes(const pz_module* m, size_t cnv_offset, size_t res_size, size_t a_size, size_t b_size);
int pz_cnv_pairwise_apply_dft(pz_module* m, size_t cnv_offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                              const double* a, size_t a_cols, size_t a_size, const double* b, size_t b_cols, size_t b_size,
                              size_t col_i, size_t col_j);
/* hal_impl.rs:748, :750 cnv_prepare_self: left and right prepared from the same `a` (both (cols, size)) */
size_t pz_cnv_prepare_self_tmp_bytes(const pz_module* m, size_t res_size, size_t a_size);
int pz_cnv_prepare_self(pz_module* m, double* left, double* right, size_t cols, size_t size,
                        const int64_t* a, size_t a_cols, size_t a_size, int64_t mask);

/* GLWE tensoring on `batch` device-resident ciphertext pairs (poulpy-core/src/operations/glwe.rs): PZ_TENSOR_APPLY glwe_tensor_apply
 * :700-807, PZ_TENSOR_APPLY_ADD_ASSIGN glwe_tensor_apply_add_assign :809-913, PZ_TENSOR_SQUARE glwe_tensor_square_apply :609-698
 * (b is ignored).  a / b: batch GLWEs (rank+1 columns, a_size / b_size limbs, one base2k); res: batch GLWETensors =
 * VecZnx((rank+1)(rank+2)/2, res_size), the pair (i, j >= i) in column i*(rank+1) - i*(i+1)/2 + j.  a_effective_k / b_effective_k:
 * the callers' precision in bits (div_ceil(base2k) must equal the size; the bits below it in the bottom limb are masked). */
typedef struct {
    uint64_t rank;
    uint64_t a_size, b_size, ab_base2k;
    uint64_t a_effective_k, b_effective_k;
    uint64_t res_size, res_base2k;
    uint64_t cnv_offset;
} pz_glwe_tensor_params;
enum { PZ_TENSOR_APPLY = 0, PZ_TENSOR_APPLY_ADD_ASSIGN = 1, PZ_TENSOR_SQUARE = 2 };
size_t pz_glwe_tensor_apply_workspace_bytes(const pz_module* m, const pz_glwe_tensor_params* p, int mode, size_t batch);
int pz_glwe_tensor_apply_batched(pz_module* m, int64_t* res, const int64_t* a, const int64_t* b, const pz_glwe_tensor_params* p, int mode,
                                 size_t batch);
/* glwe_tensor_relinearize (operations/glwe.rs:541-607) on `batch` GLWETensors `a` (VecZnx((rank+1)(rank+2)/2, p->a_size)) sharing one
 * prepared tensor key tsk_pmat (GGLWE rank*(rank+1)/2 -> rank: rows = p->dnum, cols_in = rank*(rank+1)/2, cols_out = rank+1,
 * size = p->key_size): the pair columns are key-switched, the first rank+1 columns are added to the big value, res = batch GLWEs
 * (rank+1, p->res_size).  Runs on the fused three-kernel pipeline for dsize = 1 and equal base2k (the configs[4] case). */
int pz_glwe_tensor_relinearize_batched(pz_module* m, int64_t* res, const int64_t* a, const double* tsk_pmat, const pz_glwe_op_params* p,
                                       size_t batch);
/* glwe_tensor_apply (mode PZ_TENSOR_APPLY) or glwe_tensor_square_apply (PZ_TENSOR_SQUARE, b ignored) followed by glwe_tensor_relinearize
 * with the GLWETensor in scratch - poulpy-ckks's ciphertext multiplication / square (poulpy-ckks/src/leveled/default/mul.rs:49-85,
 * :131-170: `tmp` taken from the scratch space, filled by the tensoring, consumed by the relinearization).  tp describes the tensoring
 * (tp->res_size / res_base2k: the tensor), rp the relinearization (rp->a_size = tp->res_size, rp->a_base2k = tp->res_base2k).  Same
 * digits as the two calls above on an i64 tensor; where every digit of the tensor fits 16 bits (one base2k <= 14 throughout, pipeline
 * plans) the tensor only exists as 16-bit copies in the workspace: 2 B per coefficient written and read instead of 8.  Device pointers. */
int pz_glwe_tensor_mul_relinearize_batched(pz_module* m, int64_t* res, const int64_t* a, const int64_t* b, const double* tsk_pmat,
                                           const pz_glwe_tensor_params* tp, const pz_glwe_op_params* rp, int mode, size_t batch);

/* BlindRotationExecute<CGGI>::blind_rotation_execute (poulpy-bin-fhe/src/blind_rotation/algorithms/cggi/algorithm.rs:76-118)
 * on `batch` LWE ciphertexts that share the lookup table and the prepared blind-rotation key:
 *   block_size > 1 : execute_block_binary  (:265-368)       block_size == 1 : execute_standard (:370-440)
 * (extension_factor > 1, execute_block_binary_extended :121-273: pz_blind_rotation_execute_extended_batched below).
 *   res     batch x GLWE(rank+1, res_size), overwritten
 *   lwe_2n  batch x (n_lwe+1) i64: the output of mod_switch_2n (algorithms/mod.rs:136-171, host i64 code that stays in the
 *           caller): [b, a_1 .. a_n_lwe]
 *   lut     VecZnx(1, lut_size), shared                      brk  n_lwe prepared GGSWs (pz_vmp_prepare; rows = dnum,
 *           cols_in = cols_out = rank+1, size = brk_size), contiguous, shared
 * The prepared monomials x_pow_a of BlindRotationKeyPrepared (key_prepared.rs:66-74) are not an argument: in this
 * backend's spectrum order DFT(X^a) is a row of roots of unity and is generated from a 2n-entry table.
 * All pointers are device pointers. */
typedef struct {
    uint64_t rank;
    uint64_t n_lwe;
    uint64_t block_size;
    uint64_t dnum;      /* rows of each GGSW of the key */
    uint64_t brk_size;  /* limbs of the key */
    uint64_t base2k;    /* of res, lut and key (the reference asserts they agree) */
    uint64_t res_size;
    uint64_t lut_size;
} pz_blind_rotation_params;
int pz_blind_rotation_execute_batched(pz_module* m, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                      const pz_blind_rotation_params* p, size_t batch);
size_t pz_blind_rotation_workspace_bytes(const pz_module* m, const pz_blind_rotation_params* p, size_t batch);

/* ---- LWE glue of the gate bootstrap (BASELINE configs[3]: mod switch -> blind rotation -> sample extract / LWE key switch) on
 * device-resident batches.  An LWE is the reference's container VecZnx(n = n_lwe + 1, one column, `size` limbs): limb i =
 * [b, a_0 .. a_{n_lwe-1}] at i * (n_lwe + 1); a batch is `batch` of them back to back.  Device pointers only.
 *
 * mod_switch_2n (poulpy-bin-fhe/src/blind_rotation/algorithms/mod.rs:136-176): res = batch vectors of n_lwe + 1 values, the input
 * of pz_blind_rotation_execute_batched; n2 = 2 * extension_factor * n_glwe; negate != 0 = LookUpTableRotationDirection::Left.
 * Restated literally: base2k > log2n rounds limb 0, otherwise the following limbs are appended (un-negated, as in the reference). */
int pz_lwe_mod_switch_2n_batched(pz_module* m, int64_t* res, const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t base2k, size_t n2,
                                 int negate, size_t batch);
/* LWESampleExtract::lwe_sample_extract (poulpy-core/src/api/conversion.rs:15-40): coefficient 0 of column 0 and the first
 * res_n_lwe coefficients of column 1 of each GLWE (a_cols columns, a_size limbs); limbs beyond min(res_size, a_size) are zero */
int pz_lwe_sample_extract_batched(pz_module* m, int64_t* res, size_t res_n_lwe, size_t res_size, const int64_t* a, size_t a_cols,
                                  size_t a_size, size_t batch);
/* LWEKeySwitch::lwe_keyswitch (poulpy-core/src/keyswitching/lwe.rs:49-94): embed -> glwe_keyswitch -> sample extract.
 * p: rank = rank_out = 1; a_size / a_base2k = the input LWE's, res_size / res_base2k = the output LWE's; ksk_pmat: the prepared
 * GGLWE (rows = p->dnum, cols_in = 1, cols_out = 2, size = p->key_size), device or host-resident like any prepared key */
int pz_lwe_keyswitch_batched(pz_module* m, int64_t* res, size_t res_n_lwe, const int64_t* a, size_t a_n_lwe, const double* ksk_pmat,
                             const pz_glwe_op_params* p, size_t batch);
/* GLWEFromLWE::glwe_from_lwe (poulpy-core/src/conversion/lwe_to_glwe.rs:46-121): the LWE is embedded into a rank-1 GLWE in the
 * KEY's base with p->a_size = ceil(lwe_size * lwe_base2k / key_base2k) limbs (cross-base: vec_znx_normalize per column, :82-116),
 * then key-switched rank 1 -> p->rank_out.  res = batch GLWEs (rank_out + 1, p->res_size). */
int pz_glwe_from_lwe_batched(pz_module* m, int64_t* res, const int64_t* lwe, size_t n_lwe, size_t lwe_size, size_t lwe_base2k,
                             const double* ksk_pmat, const pz_glwe_op_params* p, size_t batch);
/* LWEFromGLWE::lwe_from_glwe (poulpy-core/src/conversion/glwe_to_lwe.rs:42-90): multiply by X^-a_idx (a_idx != 0), key switch
 * p->rank -> 1 into a GLWE with the LWE's base and size, sample extract.  a = batch GLWEs (p->rank + 1, p->a_size). */
int pz_lwe_from_glwe_batched(pz_module* m, int64_t* res, size_t res_n_lwe, const int64_t* a, size_t a_idx, const double* ksk_pmat,
                             const pz_glwe_op_params* p, size_t batch);
/* execute_block_binary_extended (algorithm.rs:121-273): extension_factor > 1 (a power of two), block_size > 1.  lwe_2n is the
 * output of mod_switch_2n(2 * n * extension_factor); lut = the extension_factor polynomials lut.data[j], each
 * VecZnx(1, lut_size), contiguous; tmp = device scratch of pz_blind_rotation_extended_tmp_bytes; batch * extension_factor
 * <= 65535 per call.  The reference's skipped updates (:217, :233, :244) are reproduced. */
size_t pz_blind_rotation_extended_tmp_bytes(const pz_module* m, const pz_blind_rotation_params* p, size_t extension_factor, size_t batch);
int pz_blind_rotation_execute_extended_batched(pz_module* m, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                               const pz_blind_rotation_params* p, size_t extension_factor, void* tmp, size_t tmp_bytes,
                                               size_t batch);
/* CircuitBootstrappingExecute::circuit_bootstrapping_execute_to_constant (poulpy-bin-fhe/src/circuit_bootstrapping/
 * circuit.rs:177-195, core :219-370 with to_exponent = false) on `batch` LWE ciphertexts -> `batch` contiguous GGSWs
 * (MatZnx layout, rows = res_dnum, cols_in = cols_out = rank+1, size = res_size), for the case the reference's own
 * benchmark runs (poulpy-bench bench_suite/schemes/circuit_bootstrapping.rs: one base2k for the blind-rotation key, the
 * automorphism keys, the tensor keys and the result) and the one its tests run (a base2k per object: the fields at the end of
 * the params).  execute_to_exponent (:197-216) with
 * log_gap_in == log_gap_out is the same call with the step list of the partial trace (post_process :418-420: steps
 * log_n - log_gap_in + 1 .. log_n) and the table / mod-switch direction the shim builds for that mode (:276-301);
 * pz_circuit_bootstrapping_execute_to_exponent_batched below does that mapping and the repacking branch (:392-417).
 *   lwe_2n, lut, brk   as for pz_blind_rotation_execute_batched (the shim builds the table with the reference's host code
 *                      lookup_table.rs and passes gap = 2*lut.drift/extension_factor, circuit.rs:333)
 *   gals / atk_pmats   HOST arrays, one per trace step skip..log_n (as for pz_glwe_trace_batched; skip = 0 in constant mode): prepared automorphism keys
 *                      (rank -> rank, rows = atk_dnum, size = atk_size)
 *   tsk_pmats          HOST array of rank prepared tensor keys tsk.at(c) (rows = tsk_dnum, size = tsk_size)
 *   tmp                device scratch of pz_circuit_bootstrapping_tmp_bytes (the reference's `scratch`, circuit.rs:149-175) */
typedef struct {
    pz_blind_rotation_params br; /* res_size = limbs of the GLWE the rotation produces (brk layout = atk layout) */
    uint64_t atk_dnum, atk_size;
    uint64_t tsk_dnum, tsk_size;
    uint64_t res_dnum, res_size; /* the output GGSW */
    uint64_t gap;
    uint64_t extension_factor;   /* 0 or 1: execute_block_binary / execute_standard; > 1: the extended rotation (lut = that many
                                    polynomials, lwe_2n switched to 2*n*extension_factor, gap as circuit.rs:333 computes it) */
    /* One base2k per object, as the reference's own tests run it (circuit_bootstrapping/tests/circuit_bootstrapping.rs:49-53: result 15,
     * blind-rotation key 13, automorphism keys 11, tensor keys 12); br.base2k is the blind-rotation key's.  0 = br.base2k. */
    uint64_t atk_base2k, tsk_base2k, res_base2k;
    uint64_t atk_glwe_size;      /* limbs of the rotated GLWE re-expressed in the automorphism keys' base (circuit.rs:311-331:
                                    ceil(brk.max_k / atk_base2k)); 0 = br.res_size */
    uint64_t trace_size;         /* limbs of glwe_trace's temporary (glwe_trace.rs:107-112: ceil(max(brk.max_k, res.max_k) / atk_base2k));
                                    0 = max(atk_glwe_size, res_size), which is that value when the bases are equal */
} pz_circuit_bootstrapping_params;
size_t pz_circuit_bootstrapping_tmp_bytes(const pz_module* m, const pz_circuit_bootstrapping_params* p, size_t batch);
int pz_circuit_bootstrapping_execute_to_constant_batched(pz_module* m, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut,
                                                         const double* brk, size_t nsteps, const int64_t* gals,
                                                         const double* const* atk_pmats, const double* const* tsk_pmats,
                                                         const pz_circuit_bootstrapping_params* p, void* tmp, size_t tmp_bytes,
                                                         size_t batch);
/* circuit_bootstrapping_execute_to_exponent (circuit.rs:197-216, post_process :373-421), same conventions; gals / atk_pmats
 * cover ALL log2(n) trace steps; log_gap_in = bits(gap * next_pow2(res_dnum) - 1) (circuit.rs:342) comes from the shim.
 * log_gap_in == log_gap_out: the partial trace (:418-420); otherwise the repacking branch (:392-417: 2^log_domain shifted
 * copies + glwe_pack), which needs res_size <= br.res_size and the larger scratch of *_to_exponent_tmp_bytes. */
size_t pz_circuit_bootstrapping_to_exponent_tmp_bytes(const pz_module* m, const pz_circuit_bootstrapping_params* p, size_t log_domain,
                                                      size_t batch);
int pz_circuit_bootstrapping_execute_to_exponent_batched(pz_module* m, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut,
                                                         const double* brk, const int64_t* gals, const double* const* atk_pmats,
                                                         const double* const* tsk_pmats, const pz_circuit_bootstrapping_params* p,
                                                         size_t log_gap_in, size_t log_gap_out, size_t log_domain, void* tmp,
                                                         size_t tmp_bytes, size_t batch);
/* device workspace the GLWE-level calls reserve for `batch` ciphertexts, for the pipeline the call will actually take (fused
 * three-kernel or five-kernel) incl. the growth slack of the module's grow-only arena; keyswitch: 0 external product, 1 key switch,
 * 2 automorphism family, 3 tensor relinearization */
size_t pz_glwe_op_workspace_bytes(const pz_module* m, const pz_glwe_op_params* p, size_t batch, int keyswitch);
/* Optional: declare a prepared key (device pointer from pz_vmp_prepare) immutable until unpinned.  The batched calls
 * above then reuse a row-sliced copy built once here instead of rebuilding it per call (+~3 % at the metric shape,
 * costs one extra copy of the key in HBM).  Modifying a pinned key without unpinning it first is a caller error. */
int pz_module_pin_key(pz_module* m, const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size);
int pz_module_unpin_key(pz_module* m, const double* pmat);
/* drops the device mirror of a host-resident prepared key (see "Conventions" above); unknown pointers are ignored */
int pz_module_forget_host_key(pz_module* m, const double* host_pmat);
/* number of host-resident prepared keys currently mirrored on the device (diagnostic) */
size_t pz_module_host_key_mirrors(pz_module* m);
/* Tuning knob: number of ciphertexts pushed through the three-kernel pipeline per
 * wave so that intermediates stay in the 256 MiB Infinity Cache (0 = auto). */
int pz_module_set_chunk(pz_module* m, size_t cts_per_chunk);
/* Kernel-fusion knobs of the batched GLWE ops (both on by default; the unfused path is the per-op one,
 * kept selectable so that tests can compare the two bit for bit). */
int pz_module_set_fusion(pz_module* m, int fuse_tail, int fuse_mid);
/* N = 4096 only: plain external products / key switches with <= 4 key limbs run a two-kernel pipeline (whole polynomials in LDS,
 * the spectra cross HBM once) instead of the three-kernel one; on by default, selectable so that tests can compare the two. */
int pz_module_set_small_path(pz_module* m, int enable);
/* The launch-bound composite calls (pz_blind_rotation_execute_batched, pz_glwe_trace_batched,
 * pz_circuit_bootstrapping_execute_to_constant_batched: hundreds of short kernels per call) are captured into a HIP graph
 * the second time they are issued with the same arguments and replayed as one graph launch afterwards (on by default;
 * the buffers' CONTENTS may change between calls, their addresses and shapes are the key).  pz_module_graph_launches
 * counts the calls served by a graph. */
int pz_module_set_graphs(pz_module* m, int enable);
uint64_t pz_module_graph_launches(const pz_module* m);
/* Diagnostic only: run a subset of the fused pipeline's stages (bit 0 pass 1, bit 1 middle, bit 2 tail); outputs are
 * meaningless unless mask == 7.  Used by tools/dbg to measure stage overlap. */
int pz_module_set_debug_stages(pz_module* m, int mask);

/* batched primitives (device pointers; object b at ptr + b*len(object)) */
int pz_vec_znx_dft_apply_batched(pz_module* m, size_t batch, size_t step, size_t offset,
                                 double* res, size_t res_cols, size_t res_size, size_t res_col,
                                 const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_idft_apply_consume_batched(pz_module* m, size_t batch, void* data, size_t cols, size_t size);
int pz_vmp_apply_dft_to_dft_batched(pz_module* m, size_t batch, double* res, size_t res_cols, size_t res_size,
                                    const double* a, size_t a_cols, size_t a_size,
                                    const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size,
                                    size_t limb_offset);
int pz_vec_znx_big_normalize_batched(pz_module* m, size_t batch,
                                     int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset, size_t res_col,
                                     const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col);

/* ---- batched i64 VecZnx family: the limb-wise ops poulpy-core runs between the hot-path calls, on `batch`
 * device-resident containers back to back (object b at ptr + b*n*cols*size) — one launch per limb range instead of one call per
 * ciphertext.  Same semantics as the per-container functions above (hal_impl.rs:59-131, :225, :289, :41, :137, :165). ----------- */
int pz_vec_znx_add_into_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                                const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
int pz_vec_znx_sub_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                           const int64_t* a, size_t a_cols, size_t a_size, size_t a_col,
                           const int64_t* b, size_t b_cols, size_t b_size, size_t b_col);
int pz_vec_znx_add_assign_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                  const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_sub_assign_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                  const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_sub_negate_assign_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                         const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_negate_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                              const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_copy_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                            const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_zero_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_col);
int pz_vec_znx_rotate_batched(pz_module* m, size_t batch, int64_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                              const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_normalize_batched(pz_module* m, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k,
                                 int64_t res_offset, size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k,
                                 size_t a_col);
int pz_vec_znx_lsh_batched(pz_module* m, size_t batch, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size,
                           size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);
int pz_vec_znx_rsh_batched(pz_module* m, size_t batch, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size,
                           size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_col);

/* ---- multi-GPU ---------------------------------------------------------------------------------- *
 * One process per GPU; independent ciphertexts are block-sharded over the ranks by the caller and never exchanged.  The only
 * collective of the path is the broadcast of a prepared evaluation key from the rank that prepared it: RCCL (ncclBroadcast)
 * over xGMI, on the module stream.  RCCL is loaded on first use.  Bootstrap like NCCL: rank 0 obtains an id, every rank
 * receives it out of band (MPI, a file, torch.distributed ...) and calls pz_comm_init_rank. */
size_t pz_comm_unique_id_bytes(void);                 /* 128 */
int pz_comm_available(void);                          /* PZ_OK when RCCL can be loaded in this process (dlopen + symbols); no id drawn, no socket opened */
int pz_comm_unique_id(void* out_id);                  /* ncclGetUniqueId */
int pz_comm_init_rank(pz_module* m, int world_size, int rank, const void* unique_id);
int pz_comm_destroy(pz_module* m);
int pz_comm_rank(const pz_module* m);                 /* -1 without a communicator */
int pz_comm_world_size(const pz_module* m);
/* in-place broadcast of `bytes` bytes of device memory from `root` (64 MiB buckets); asynchronous on the module stream */
int pz_bcast_key(pz_module* m, void* dev_buf, size_t bytes, int root);

/* ---- instrumentation ------------------------------------------------------- */
/* Times `reps` back-to-back launches of the last pipeline's dominant kernel with HIP
 * events on the module stream; bench.py uses pz_event_* to bracket launches. */
int pz_event_create(void** ev);
int pz_event_destroy(void* ev);
int pz_event_record(pz_module* m, void* ev);
int pz_event_elapsed_ms(void* ev0, void* ev1, float* ms); /* synchronises on ev1 */
/* Per-kernel-class timing with HIP events on the module stream (what bench.py's "roofline"
 * object is computed from).  Classes: */
enum {
    PZ_K_FWD_PASS1 = 0, /* i64 -> T      (column pass of the forward transform)        */
    PZ_K_FWD_PASS2 = 1, /* T -> spectrum  (row pass)                                    */
    PZ_K_VMP = 2,       /* vector-matrix product in the DFT domain                      */
    PZ_K_INV_PASS2 = 3, /* spectrum -> T                                                */
    PZ_K_INV_PASS1 = 4, /* T -> i64, round(x/m)                                         */
    PZ_K_NORMALIZE = 5, /* base-2^k carry chain                                         */
    PZ_K_ELEMENTWISE = 6,
    PZ_K_FUSED_MID = 7, /* fused row pass + VMP + inverse row pass                      */
    PZ_K_FUSED_TAIL = 8, /* fused inverse column pass + normalize                       */
    PZ_KCLASS_COUNT = 9
};
int pz_module_set_kernel_timing(pz_module* m, int enable); /* enabling resets the counters */
/* synchronises the stream and returns accumulated launches / milliseconds of one class */
int pz_module_get_kernel_stats(pz_module* m, int kclass, uint64_t* launches, double* total_ms);
const char* pz_kernel_class_name(int kclass);

/* Rounding-margin probe.  While enabled, EVERY kernel that rounds an inverse-FFT value to an integer (the fused tail in all its
 * forms, the per-op inverse pass, the small-ring inverse, the one-kernel blind rotation) records the largest |x - round(x)| it
 * saw; pz_module_get_margin returns that maximum since the probe was last enabled.  0.5 is the point where a rounding flips,
 * the tests assert < 0.05 on every path.  The probed instantiations are separate (slower) kernels: measure with the probe off. */
int pz_module_set_margin_probe(pz_module* m, int enable);
int pz_module_get_margin(pz_module* m, double* max_frac);

/* Which kernel instantiations the hot dispatch sites (middle kernel, blind-rotation kernels) have chosen since the last reset, as one
 * "; "-separated string: measurement tools print it next to their numbers. */
int pz_module_dispatch_notes(pz_module* m, char* buf, size_t len, int reset);

/* Debug: workspace guards.  With POULPY_DBG_CANARY=1 in the environment every segment the library carves out of its workspaces is
 * followed by a 256-byte guard that is verified when the API call returns (the process aborts with a message on an overrun; HIP
 * graphs are not used in this mode).  pz_debug_workspace_overrun carves two segments of `bytes` and writes `overrun` bytes past the
 * end of the first one: the self-test of that mechanism (returns PZ_OK when the mode is off: nothing is checked then). */
int pz_debug_workspace_overrun(pz_module* m, size_t bytes, size_t overrun);

#ifdef __cplusplus
}
#endif
#endif
