// api_common.hpp — host-side helpers shared by the api*.hip translation units: pointer resolution / staging of host
// containers, the per-call prologue, and the device-level transforms every composite op is built from.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "internal.hpp"

using namespace pz;

// multiplicative inverse of an odd p modulo 2n (n a power of two): Newton iteration doubles the valid bits
static inline unsigned inv_mod_2n(long long p, long long n) {
    const unsigned long long mask = 2ull * (unsigned long long)n - 1ull;
    const unsigned long long a = (unsigned long long)p & mask;
    unsigned long long x = a;  // correct to 3 bits
    for (int i = 0; i < 6; ++i) x *= 2ull - a * x;
    return (unsigned)(x & mask);
}
static inline int ensure_w2n(pz_module* M) {
    if (M->w2n) return PZ_OK;
    const long long two_n = 2 * (long long)M->n;
    std::vector<cplx> h((size_t)two_n);
    double c, s;
    for (long long t = 0; t < two_n; ++t) { root_of_unity(t, two_n, c, s); h[(size_t)t] = make_double2(c, s); }
    return upload_table(&M->w2n, h);
}

// ------------------------------------------------------------------------------
// device-level operations (device pointers, batch strides in scalars)
// ------------------------------------------------------------------------------
// vec_znx_dft_apply on `ncs` consecutive columns (res_col.., a_col..)  [vec_znx_dft.rs:160-200]
static inline int dev_dft_apply(pz_module* M, int batch, int step, int offset, DV res, int res_col, DV a, int a_col, int ncs,
                         const cplx* mul, cplx* T) {
    const long long n = (long long)M->n;
    const int steps = (a.size + step - 1) / step;
    const int min_steps = std::min(res.size, steps);
    int nv = 0;
    if (offset < a.size) nv = std::min(min_steps, (a.size - offset + step - 1) / step);
    if (nv > 0) {
        PolyMap sm{nv, ncs, a.bs, (long long)step * a.cols * n, n, n * ((long long)offset * a.cols + a_col)};
        PolyMap dm{nv, ncs, res.bs, (long long)res.cols * n, n, n * res_col};
        const int npolys = batch * nv * ncs;
        if (small_transform_supported(M)) {   // N <= 4096: the whole transform in LDS, one kernel, the spectrum's only trip through HBM
            PZ_TRY(launch_small_fwd(M, npolys, (const long long*)a.p, sm, (cplx*)res.p, true, &dm, mul));
        } else {
            PZ_TRY(launch_fwd_pass1(M, npolys, (const long long*)a.p, sm, T));
            PZ_TRY(launch_fwd_pass2(M, npolys, T, (double*)res.p, dm, mul));
        }
    }
    // limbs [nv, min_steps) are left untouched (vec_znx_dft.rs:191-194); the rest is zeroed
    for (int c = 0; c < ncs; ++c)
        PZ_TRY(launch_ew(M, EW_ZERO, poly_ptr(M, res, res_col + c, min_steps), res.bs, limb_stride(M, res), nullptr, 0, 0,
                         nullptr, 0, 0, res.size - min_steps, batch));
    return PZ_OK;
}

// inverse transform of `nlimbs` limbs x `ncs` columns: a (f64) -> res (i64)
static inline int dev_idft(pz_module* M, int batch, DV res, int res_col, DV a, int a_col, int ncs, int nlimbs, cplx* T) {
    const long long n = (long long)M->n;
    if (nlimbs <= 0) return PZ_OK;
    PolyMap sm{nlimbs, ncs, a.bs, (long long)a.cols * n, n, n * a_col};
    PolyMap dm{nlimbs, ncs, res.bs, (long long)res.cols * n, n, n * res_col};
    const int npolys = batch * nlimbs * ncs;
    if (small_transform_supported(M)) return launch_small_idft(M, npolys, (const double*)a.p, sm, (long long*)res.p, dm);
    PZ_TRY(launch_inv_pass2(M, npolys, (const double*)a.p, sm, T));
    PZ_TRY(launch_inv_pass1(M, npolys, T, (long long*)res.p, dm));
    return PZ_OK;
}

// ------------------------------------------------------------------------------
// host/device pointer resolution
// ------------------------------------------------------------------------------
static inline bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear sticky error for unregistered host memory
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// A staged argument: device view of a (possibly host) container of `bytes` bytes.
// Host buffers are copied into the module's staging arena; outputs are copied back by
// finish_call() after the stream has drained (blocking hipMemcpy: no reliance on the ordering
// of asynchronous copies into pageable memory).
struct Stage {
    pz_module* M = nullptr;
    void* host = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    bool owned = false, out = false;
    int in(const void* p, size_t nbytes, bool copy_in, bool copy_out, pz_module* mod) {
        M = mod; bytes = nbytes; out = copy_out;
        if (nbytes == 0) { dev = (void*)p; return PZ_OK; }
        if (is_device_ptr(p)) { dev = (void*)p; owned = false; return PZ_OK; }
        host = (void*)p;
        owned = true;
        PZ_TRY(arena_alloc(M, nbytes, &dev));
        if (copy_in) PZ_HIP(hipMemcpyAsync(dev, host, nbytes, hipMemcpyHostToDevice, M->stream));
        return PZ_OK;
    }
    int finish() {
        if (owned && out) M->pending_out.push_back({host, dev, bytes});
        owned = false;
        return PZ_OK;
    }
};

static inline int finish_call(pz_module* M, bool any_host) {
    if (any_host || !M->pending_out.empty()) {
        PZ_HIP(hipStreamSynchronize(M->stream));
        for (auto& po : M->pending_out) PZ_HIP(hipMemcpy(po.host, po.dev, po.bytes, hipMemcpyDeviceToHost));
        M->pending_out.clear();
    }
    return PZ_OK;
}

// verifies the workspace guards of the call when it returns, by whatever path (POULPY_DBG_CANARY; no-op otherwise)
struct CanaryScope {
    pz_module* M;
    const char* fn;
    CanaryScope(pz_module* M_, const char* fn_) : M(M_), fn(fn_) { if (canary_mode()) M->guards.clear(); }
    ~CanaryScope() { if (canary_mode()) canary_verify(M, fn); }
};
// api.hip: drops this module's device mirrors of host keys whose range was published (re-prepared, zeroed, forgotten, freed) since its
// last sweep - one atomic load when nothing was
void mirror_sweep_on_enter(pz_module* M);
#define PZ_ENTER(M)                                              \
    if (!(M)) return fail(PZ_ERR_INVALID, "null module");        \
    std::lock_guard<std::mutex> lock_((M)->mu);                  \
    PZ_HIP(hipSetDevice((M)->device));                           \
    CanaryScope canary_((M), __func__);                          \
    arena_reset(M);                                              \
    mirror_sweep_on_enter(M);

static inline size_t vbytes(const pz_module* M, size_t cols, size_t size) { return (size_t)M->n * cols * size * 8; }


static inline int need_T(pz_module* M, size_t npolys, cplx** T) {
    PZ_TRY(ws_reserve(M, npolys * (size_t)M->m * sizeof(cplx)));
    *T = (cplx*)M->ws;
    return PZ_OK;
}

#define PZ_CHECK_COL(col, cols, what) PZ_REQUIRE((col) < (cols), "%s: col %zu >= cols %zu", what, (size_t)(col), (size_t)(cols))
