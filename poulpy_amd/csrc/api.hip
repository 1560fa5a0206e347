// api.hip — the C ABI of libpoulpy_hip.so (include/poulpy_hip.h) on top of the
// gfx950 kernels (launch_*.hip; interfaces in internal.hpp).
//
// Structure: every public entry point (a) validates shapes the way the reference
// asserts them, (b) resolves each pointer to a device pointer (staging host
// buffers), (c) calls a `dev_*` routine that only sees device pointers and batch
// strides, (d) copies results back for host buffers.  The batched entry points
// call the same `dev_*` routines with batch > 1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "api_common.hpp"
#include <unordered_map>

#include "api_glwe.hpp"

using namespace pz;

// ------------------------------------------------------------------------------
// host containers at the batched GLWE entry points (what the Rust shim's CoreImpl overrides pass: poulpy-hal buffers are host
// addressable by contract).  Ciphertexts are staged like any per-op argument; prepared keys get a device mirror.
// ------------------------------------------------------------------------------
static uint64_t host_fingerprint(const void* p, size_t bytes) {
    // FNV-1a over the first / last 4 KiB and ~8192 words strided over the rest: any (re)preparation of the key changes it with
    // overwhelming probability; a caller that patches a prepared matrix in place behind the backend's back calls
    // pz_module_forget_host_key (include/poulpy_hip.h)
    const uint64_t* w = (const uint64_t*)p;
    const size_t nw = bytes / 8;
    uint64_t h = 1469598103934665603ull ^ (uint64_t)bytes;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
    const size_t edge = std::min<size_t>(nw, 512);
    for (size_t i = 0; i < edge; ++i) mix(w[i]);
    for (size_t i = nw - edge; i < nw; ++i) mix(w[i]);
    const size_t stride = std::max<size_t>(1, nw / 8192) | 1;
    for (size_t i = 0; i < nw; i += stride) mix(w[i]);
    return h;
}
// Process-wide invalidation of host-resident prepared keys: a sibling module on another thread may mirror the same host
// buffer, and the sampled fingerprint can miss an in-place change.  Writers (pz_vmp_prepare, pz_vmp_zero, pz_module_forget_host_key,
// pz_free_bytes) publish the host range under one small lock; a mirror is valid only if no range published after its validation
// overlaps it.  The ring keeps the last kInvalRing ranges; a mirror older than the ring's horizon is revalidated conservatively.
static std::mutex g_inval_mu;
struct HostInval { const char* lo; const char* hi; uint64_t epoch; };
static constexpr size_t kInvalRing = 1024;
static std::vector<HostInval> g_inval;       // ring, oldest overwritten
static size_t g_inval_next = 0;
static uint64_t g_inval_epoch = 0;           // epoch of the newest published range
static uint64_t g_inval_horizon = 0;         // ranges with epoch <= horizon have left the ring
void host_key_invalidate(const void* p, size_t bytes) {
    if (!p || is_device_ptr(p)) return;
    std::lock_guard<std::mutex> g(g_inval_mu);
    HostInval e{(const char*)p, (const char*)p + std::max<size_t>(bytes, 1), ++g_inval_epoch};
    if (g_inval.size() < kInvalRing) g_inval.push_back(e);
    else { g_inval_horizon = g_inval[g_inval_next].epoch; g_inval[g_inval_next] = e; g_inval_next = (g_inval_next + 1) % kInvalRing; }
}
// (current epoch, whether [p, p + bytes) was published after `since`)
static bool host_key_stale(const void* p, size_t bytes, uint64_t since, uint64_t* now) {
    std::lock_guard<std::mutex> g(g_inval_mu);
    *now = g_inval_epoch;
    if (since < g_inval_horizon) return true;   // older than the ring remembers
    const char* lo = (const char*)p; const char* hi = lo + bytes;
    for (const auto& e : g_inval) if (e.epoch > since && e.lo < hi && lo < e.hi) return true;
    return false;
}
static void drop_mirror_at(pz_module* M, size_t i) {
    auto& mr = M->mirrors[i];
    for (size_t k = 0; k < M->pinned.size(); ++k)
        if (M->pinned[k].key == mr.dev) {
            if (M->pinned[k].sliced) (void)hipFree(M->pinned[k].sliced);
            M->pinned.erase(M->pinned.begin() + (long)k);
            break;
        }
    if (mr.dev) (void)hipFree(mr.dev);
    M->mirrors.erase(M->mirrors.begin() + (long)i);
    M->graph_epoch++;
}
int forget_host_key(pz_module* M, const void* host) {
    host_key_invalidate(host, 1);   // every module's mirror of this buffer, not only the caller's
    for (size_t i = 0; i < M->mirrors.size(); ++i)
        if (M->mirrors[i].host == host) {
            PZ_HIP(hipStreamSynchronize(M->stream));
            drop_mirror_at(M, i);
            return PZ_OK;
        }
    return PZ_OK;
}
// device pointer of a prepared key: itself when it is one, else its (validated, possibly refreshed) mirror; the mirror also gets
// the row-sliced copy of the fused pipeline (as pz_module_pin_key would build it), valid for as long as the mirror is
// the backend-private copy of a prepared key that the batched GLWE calls read: row-sliced for the fused pipeline (m = m1 x 128 / 256
// plans), or P'[q1][poly][q2] for the small-ring pipeline at N = 1024 / 2048 (device_small.hpp) - those rings permuted their key on
// every call until round 3 (5 - 9 % of a 1024-ciphertext call)
static bool key_copy_applies(const pz_module* M) {
    return ((M->plan.m2 == 256 || M->plan.m2 == 128) && (M->plan.m1 % 16) == 0) || small_transform_supported(M);
}
static int build_key_copy(pz_module* M, const double* dev, cplx* sliced, size_t npolys) {
    if ((M->plan.m2 == 256 || M->plan.m2 == 128) && (M->plan.m1 % 16) == 0) return launch_permute_pmat(M, dev, sliced, (int)npolys);
    return launch_small_permute(M, dev, sliced, (int)npolys);
}
static int resolve_key(pz_module* M, const double* pmat, size_t bytes, const double** out) {
    if (is_device_ptr(pmat)) { *out = pmat; return PZ_OK; }
    // mirrors whose host range was (re)prepared, zeroed, forgotten or freed since their validation - by any module - go first
    for (size_t i = M->mirrors.size(); i-- > 0;) {
        uint64_t now = 0;
        if (host_key_stale(M->mirrors[i].host, M->mirrors[i].bytes, M->mirrors[i].epoch, &now)) {
            PZ_HIP(hipStreamSynchronize(M->stream));
            drop_mirror_at(M, i);
        } else M->mirrors[i].epoch = now;
    }
    const uint64_t fp = host_fingerprint(pmat, bytes);
    for (size_t i = 0; i < M->mirrors.size(); ++i) {
        auto& mr = M->mirrors[i];
        if (mr.host != (const void*)pmat) continue;
        if (mr.bytes == bytes && mr.fp == fp) { mr.stamp = ++M->mirror_clock; *out = (const double*)mr.dev; return PZ_OK; }
        PZ_HIP(hipStreamSynchronize(M->stream));
        drop_mirror_at(M, i);
        break;
    }
    uint64_t epoch_now = 0;
    (void)host_key_stale(pmat, bytes, ~0ull >> 1, &epoch_now);   // (only reads the current epoch)
    size_t total = bytes;
    for (auto& mr : M->mirrors) total += mr.bytes;
    while (!M->mirrors.empty() && (M->mirrors.size() >= 64 || total > ((size_t)48 << 30))) {   // LRU: at most 64 keys / 48 GiB mirrored
        size_t lru = 0;
        for (size_t i = 1; i < M->mirrors.size(); ++i) if (M->mirrors[i].stamp < M->mirrors[lru].stamp) lru = i;
        total -= M->mirrors[lru].bytes;
        PZ_HIP(hipStreamSynchronize(M->stream));
        drop_mirror_at(M, lru);
    }
    void* dev = nullptr;
    PZ_HIP(hipMalloc(&dev, bytes));
    if (hipMemcpyAsync(dev, pmat, bytes, hipMemcpyHostToDevice, M->stream) != hipSuccess) {
        (void)hipFree(dev);
        return fail(PZ_ERR_HIP, "upload of a host-resident prepared key failed");
    }
    M->mirrors.push_back({(const void*)pmat, bytes, dev, fp, ++M->mirror_clock, epoch_now});
    M->graph_epoch++;
    if (key_copy_applies(M)) {
        const size_t npolys = bytes / ((size_t)M->n * 8);
        cplx* sliced = nullptr;
        if (hipMalloc(&sliced, bytes) == hipSuccess) {
            if (build_key_copy(M, (const double*)dev, sliced, npolys) == PZ_OK) M->pinned.push_back({(const void*)dev, sliced, bytes});
            else (void)hipFree(sliced);
        } else {
            (void)hipGetLastError();   // no room for the sliced copy: the pipeline rebuilds it per call
        }
    }
    // the caller may change or free the host key as soon as this call returns (with device-resident ciphertexts nothing else waits):
    // a fresh mirror's upload is complete before it does
    PZ_HIP(hipStreamSynchronize(M->stream));
    *out = (const double*)dev;
    return PZ_OK;
}
// a batched GLWE op whose ciphertext arguments may be host containers
struct GlweArgs {
    Stage sa, sr;
    int64_t* res = nullptr;
    const int64_t* a = nullptr;
    const double* key = nullptr;
    bool host = false;
};
static int glwe_args_in(pz_module* M, GlweArgs& g, int64_t* res, const int64_t* a, const double* pmat, size_t res_bytes, size_t a_bytes,
                        size_t key_bytes) {
    PZ_REQUIRE(res != nullptr && a != nullptr && pmat != nullptr, "null argument");
    PZ_TRY(resolve_key(M, pmat, key_bytes, &g.key));
    PZ_TRY(g.sa.in(a, a_bytes, true, false, M));
    if ((const void*)res == (const void*)a) {   // *_assign forms
        PZ_REQUIRE(res_bytes == a_bytes, "in-place call with different layouts for a and res");
        g.sr.M = M; g.sr.dev = g.sa.dev; g.sa.out = true;
    } else {
        PZ_TRY(g.sr.in(res, res_bytes, false, true, M));
    }
    g.res = (int64_t*)g.sr.dev; g.a = (const int64_t*)g.sa.dev;
    g.host = g.sa.owned || g.sr.owned;
    return PZ_OK;
}
static int glwe_args_out(pz_module* M, GlweArgs& g) {
    PZ_TRY(g.sr.finish());
    PZ_TRY(g.sa.finish());
    return finish_call(M, g.host);
}

// live modules, so that pz_free_bytes can drop the device mirror of a prepared key whose pinned host buffer is being released (the
// Rust shim's `PinnedBuf::drop`): a later allocation at the same address then never meets a stale mirror, fingerprint or not
static std::mutex g_modules_mu;
static std::vector<pz_module*> g_modules;
static std::unordered_map<void*, size_t> g_host_allocs;   // pz_alloc_bytes blocks (a prepared key may sit inside one)

// ------------------------------------------------------------------------------
// public: misc
// ------------------------------------------------------------------------------
extern "C" {

const char* pz_last_error(void) { return last_error_ref().c_str(); }
uint32_t pz_abi_version(void) { return PZ_ABI_VERSION; }   // 3: + pz_module_set_phase_tuning / _phase_tuning_state, pz_debug_workspace_overrun

int pz_module_new_on_device(uint64_t n, int device, pz_module** out) {
    if (!out) return fail(PZ_ERR_INVALID, "null out");
    *out = nullptr;
    FftPlan pl;
    if (n < 2 || (n & (n - 1))) return fail(PZ_ERR_INVALID, "n must be a power of two but is %llu", (unsigned long long)n);
    if (!make_plan(n, pl)) return fail(PZ_ERR_UNSUPPORTED, "n=%llu unsupported (need 32 <= n <= 131072)", (unsigned long long)n);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(PZ_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(PZ_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    PZ_HIP(hipSetDevice(device));
    pz_module* M = new pz_module();
    M->n = n; M->m = n >> 1; M->device = device; M->plan = pl;
    int r = PZ_OK;
    do {
        if (const char* cm = getenv("POULPY_DBG_CU_MASK")) {
            // diagnostic: restrict the module stream to N CUs ("N" or "N,mode": mode 0 = the first N mask bits, 1 = spread evenly)
            int ncus = atoi(cm), mode = 0;
            if (const char* c = strchr(cm, ',')) mode = atoi(c + 1);
            uint32_t mask[8] = {0};
            int on = 0;
            for (int i = 0; i < 256; ++i) {
                const bool en = mode == 0 ? i < ncus : ((long long)(i + 1) * ncus / 256 > (long long)i * ncus / 256);
                if (en) { mask[i >> 5] |= 1u << (i & 31); ++on; }
            }
            if (hipExtStreamCreateWithCUMask(&M->stream, 8, mask) != hipSuccess) { r = fail(PZ_ERR_HIP, "masked stream create failed"); break; }
            M->cu_count = on;
        } else
        if (hipStreamCreateWithFlags(&M->stream, hipStreamNonBlocking) != hipSuccess) { r = fail(PZ_ERR_HIP, "stream create failed"); break; }
        if ((r = build_tables(M)) != PZ_OK) break;
        if (hipMalloc(&M->margin, 8) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin alloc failed"); break; }
        if (hipMemset(M->margin, 0, 8) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin memset failed"); break; }
    } while (0);
    if (r != PZ_OK) { pz_module_free(M); return r; }
    M->tables_ref = new std::atomic<int>(1);
    {
        std::lock_guard<std::mutex> g(g_modules_mu);
        g_modules.push_back(M);
    }
    *out = M;
    return PZ_OK;
}
// A sibling for another host thread (concurrent calls on one `&Module` happen, poulpy-bin-fhe bdd_arithmetic/eval.rs:210-221):
// it shares the immutable device tables and owns everything a call mutates — stream, workspaces, staging arena, pinned-key list, key
// mirrors, graph cache, lock — so calls on different siblings run concurrently instead of queueing on one mutex.
int pz_module_clone(pz_module* P, pz_module** out) {
    if (!P || !out) return fail(PZ_ERR_INVALID, "null argument");
    *out = nullptr;
    pz_module* M = nullptr;
    {
    // (P's lock is released before the sibling is registered: pz_free_bytes takes the registry lock first, then module locks)
    std::lock_guard<std::mutex> lock_(P->mu);
    PZ_HIP(hipSetDevice(P->device));
    PZ_TRY(ensure_w2n(P));   // built lazily otherwise: the siblings must agree on who owns it
    M = new pz_module();
    M->n = P->n; M->m = P->m; M->device = P->device; M->plan = P->plan;
    M->tw1 = P->tw1; M->tw1inv = P->tw1inv; M->wL1 = P->wL1; M->wL2 = P->wL2; M->tw12 = P->tw12; M->tw12t = P->tw12t; M->w2n = P->w2n;
    M->tables_ref = P->tables_ref;
    M->tables_ref->fetch_add(1);
    M->fuse_tail = P->fuse_tail; M->fuse_mid = P->fuse_mid; M->small_path = P->small_path; M->chunk = P->chunk; M->graphs_on = P->graphs_on;
    int r = PZ_OK;
    do {
        if (hipStreamCreateWithFlags(&M->stream, hipStreamNonBlocking) != hipSuccess) { r = fail(PZ_ERR_HIP, "stream create failed"); break; }
        if (hipMalloc(&M->margin, 8) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin alloc failed"); break; }
        if (hipMemset(M->margin, 0, 8) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin memset failed"); break; }
    } while (0);
    if (r != PZ_OK) { M->tables_ref->fetch_sub(1); M->tables_ref = nullptr; M->tw1 = M->tw1inv = M->wL1 = M->wL2 = M->tw12 = M->tw12t = M->w2n = nullptr; }
    if (r != PZ_OK) { pz_module* dead = M; M = nullptr; pz_module_free(dead); return r; }
    }
    {
        std::lock_guard<std::mutex> g(g_modules_mu);
        g_modules.push_back(M);
    }
    *out = M;
    return PZ_OK;
}
int pz_module_new(uint64_t n, pz_module** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return pz_module_new_on_device(n, dev, out);
}
void pz_module_free(pz_module* M) {
    if (!M) return;
    {
        std::lock_guard<std::mutex> g(g_modules_mu);
        g_modules.erase(std::remove(g_modules.begin(), g_modules.end(), M), g_modules.end());
    }
    (void)hipSetDevice(M->device);
    if (M->stream) (void)hipStreamSynchronize(M->stream);
    // the tables go with the last sibling (a module whose construction failed early owns them alone)
    const bool last = !M->tables_ref || M->tables_ref->fetch_sub(1) == 1;
    if (last) {
        for (void* p : {(void*)M->tw1, (void*)M->tw1inv, (void*)M->wL1, (void*)M->wL2, (void*)M->tw12, (void*)M->tw12t, (void*)M->w2n})
            if (p) (void)hipFree(p);
        delete M->tables_ref;
    }
    for (void* p : {M->ws, M->ws2, (void*)M->margin})
        if (p) (void)hipFree(p);
    if (M->s_owned)
        for (void* p : {(void*)M->s_tw1, (void*)M->s_tw1inv, (void*)M->s_tw12t, (void*)M->s_wL2})
            if (p) (void)hipFree(p);
    for (auto& c : M->arena) (void)hipFree(c.p);
    for (auto& k : M->pinned) if (k.sliced) (void)hipFree(k.sliced);
    for (auto& mr : M->mirrors) if (mr.dev) (void)hipFree(mr.dev);
    if (M->comm) (void)pz_comm_destroy(M);
    for (auto& t : M->timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    for (auto e : M->event_pool) (void)hipEventDestroy(e);
    for (auto& pt : M->phase_tune) for (int i = 0; i < 8; ++i) { (void)hipEventDestroy(pt.e0[i]); (void)hipEventDestroy(pt.e1[i]); }
    for (auto& ge : M->graphs) {
        if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
    }
    if (M->stream) (void)hipStreamDestroy(M->stream);
    delete M;
}
uint64_t pz_module_n(const pz_module* M) { return M ? M->n : 0; }
int pz_module_device(const pz_module* M) { return M ? M->device : -1; }
int pz_module_sync(pz_module* M) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    PZ_HIP(hipSetDevice(M->device));
    PZ_HIP(hipStreamSynchronize(M->stream));
    return PZ_OK;
}
void* pz_module_stream(pz_module* M) { return M ? (void*)M->stream : nullptr; }
int pz_module_set_ws_shift(pz_module* M, size_t bytes) {   // diagnostic (not in the header): extra padding in front of T2'
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    M->ws_shift = bytes;
    M->graph_epoch++;
    return PZ_OK;
}
int pz_module_set_chunk(pz_module* M, size_t c) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    M->graph_epoch++;
    M->chunk = c;
    return PZ_OK;
}
int pz_module_set_small_path(pz_module* M, int enable) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    M->small_path = enable != 0;
    M->graph_epoch++;
    return PZ_OK;
}
int pz_module_set_fusion(pz_module* M, int fuse_tail, int fuse_mid) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    M->graph_epoch++;
    M->fuse_tail = fuse_tail != 0;
    M->fuse_mid = fuse_mid != 0;
    return PZ_OK;
}
// A prepared key that the caller promises not to modify while pinned: its row-sliced copy for the fused pipeline is built
// once here instead of on every batched call (saves 2 x key bytes of HBM traffic per call, ~3 % at the metric shape).
int pz_module_pin_key(pz_module* M, const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    PZ_ENTER(M);
    M->graph_epoch++;
    PZ_REQUIRE(is_device_ptr(pmat), "pz_module_pin_key takes a device pointer");
    PZ_REQUIRE(rows >= 1 && cols_in >= 1 && cols_out >= 1 && size >= 1, "pz_module_pin_key: empty shape");
    for (auto& k : M->pinned) PZ_REQUIRE(k.key != (const void*)pmat, "pz_module_pin_key: key already pinned");
    if (!key_copy_applies(M)) {                                                  // no fused / small-ring pipeline at this N: nothing to cache,
        M->pinned.push_back({(const void*)pmat, nullptr, 0});                      // but the pin is remembered so that unpin succeeds
        return PZ_OK;
    }
    const size_t npolys = rows * cols_in * cols_out * size;
    const size_t bytes = npolys * (size_t)M->n * 8;
    cplx* sliced = nullptr;
    PZ_HIP(hipMalloc(&sliced, bytes));
    const int st = build_key_copy(M, pmat, sliced, npolys);
    if (st != PZ_OK) { (void)hipFree(sliced); return st; }
    M->pinned.push_back({(const void*)pmat, sliced, bytes});
    return PZ_OK;
}
int pz_module_unpin_key(pz_module* M, const double* pmat) {
    PZ_ENTER(M);
    M->graph_epoch++;
    for (size_t i = 0; i < M->pinned.size(); ++i)
        if (M->pinned[i].key == (const void*)pmat) {
            PZ_HIP(hipStreamSynchronize(M->stream));
            if (M->pinned[i].sliced) (void)hipFree(M->pinned[i].sliced);
            M->pinned.erase(M->pinned.begin() + (long)i);
            return PZ_OK;
        }
    return fail(PZ_ERR_INVALID, "pz_module_unpin_key: key is not pinned");
}
// HIP-graph replay of the composite calls (on by default); launches: number of calls served by a graph so far
int pz_module_set_graphs(pz_module* M, int enable) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    M->graphs_on = enable != 0;
    return PZ_OK;
}
uint64_t pz_module_graph_launches(const pz_module* M) { return M ? (uint64_t)M->graph_launches : 0; }
int pz_module_set_debug_stages(pz_module* M, int mask) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    M->dbg_stages = mask;
    return PZ_OK;
}
int pz_module_set_margin_probe(pz_module* M, int enable) {
    PZ_ENTER(M);
    M->probe = enable != 0;
    PZ_HIP(hipMemsetAsync(M->margin, 0, 8, M->stream));
    PZ_HIP(hipStreamSynchronize(M->stream));
    return PZ_OK;
}
static int drain_timers(pz_module* M) {
    PZ_HIP(hipStreamSynchronize(M->stream));
    for (auto& t : M->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
            M->cls_ms[t.cls] += ms;
            M->cls_count[t.cls] += 1;
        }
        M->event_pool.push_back(t.e0);
        M->event_pool.push_back(t.e1);
    }
    M->timed.clear();
    return PZ_OK;
}
int pz_module_set_kernel_timing(pz_module* M, int enable) {
    PZ_ENTER(M);
    PZ_TRY(drain_timers(M));
    M->timing = enable != 0;
    if (enable) for (int i = 0; i < PZ_KCLASS_COUNT; ++i) { M->cls_ms[i] = 0; M->cls_count[i] = 0; }
    return PZ_OK;
}
int pz_module_set_phase_tuning(pz_module* M, int enable) {
    PZ_ENTER(M);
    M->phase_tuning = enable != 0;
    return PZ_OK;
}
int pz_module_phase_tuning_state(pz_module* M, int* shapes_tuned, int* shapes_measuring) {
    PZ_ENTER(M);
    int t = 0, u = 0;
    for (auto& e : M->phase_tune) {
        if (e.calls > 8 && e.pending == 0u && e.best >= 0) ++t; else ++u;
    }
    if (shapes_tuned) *shapes_tuned = t;
    if (shapes_measuring) *shapes_measuring = u;
    return PZ_OK;
}
int pz_module_dispatch_notes(pz_module* M, char* buf, size_t len, int reset) {
    PZ_ENTER(M);
    if (buf && len) {
        std::string all;
        for (auto& s : M->notes) { if (!all.empty()) all += "; "; all += s; }
        snprintf(buf, len, "%s", all.c_str());
    }
    if (reset) M->notes.clear();
    return PZ_OK;
}
int pz_debug_workspace_overrun(pz_module* M, size_t bytes, size_t overrun) {
    PZ_ENTER(M);
    const size_t seg = align256(bytes);
    PZ_TRY(ws_reserve(M, 2 * seg));
    char* base = (char*)M->ws;
    char* s0; char* s1;
    PZ_TRY(ws_take(M, base, seg, &s0));
    PZ_TRY(ws_take(M, base, seg, &s1));
    PZ_HIP(hipMemsetAsync(s0, 0x11, seg + overrun, M->stream));   // `overrun` bytes land behind the first segment
    PZ_HIP(hipMemsetAsync(s1, 0x22, seg, M->stream));
    return PZ_OK;   // the scope object of PZ_ENTER verifies the guards on the way out
}
int pz_module_get_kernel_stats(pz_module* M, int kclass, uint64_t* launches, double* total_ms) {
    PZ_ENTER(M);
    PZ_REQUIRE(kclass >= 0 && kclass < PZ_KCLASS_COUNT, "kernel class out of range");
    PZ_TRY(drain_timers(M));
    if (launches) *launches = M->cls_count[kclass];
    if (total_ms) *total_ms = M->cls_ms[kclass];
    return PZ_OK;
}
const char* pz_kernel_class_name(int k) {
    static const char* names[PZ_KCLASS_COUNT] = {"fwd_pass1", "fwd_pass2", "vmp", "inv_pass2", "inv_pass1", "normalize",
                                                 "elementwise", "fused_mid", "fused_tail"};
    return (k >= 0 && k < PZ_KCLASS_COUNT) ? names[k] : "?";
}
int pz_module_get_margin(pz_module* M, double* max_frac) {
    PZ_ENTER(M);
    unsigned long long bits = 0;
    PZ_HIP(hipStreamSynchronize(M->stream));
    PZ_HIP(hipMemcpy(&bits, M->margin, 8, hipMemcpyDeviceToHost));
    double d;
    memcpy(&d, &bits, 8);
    if (max_frac) *max_frac = d;
    return PZ_OK;
}

void* pz_alloc_bytes(size_t len) {
    void* p = nullptr;
    if (len == 0) len = 64;
    if (hipHostMalloc(&p, len, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    memset(p, 0, len);
    {
        std::lock_guard<std::mutex> g(g_modules_mu);
        g_host_allocs[p] = len;
    }
    return p;
}
void pz_free_bytes(void* p) {
    if (!p) return;
    size_t len = 1;
    {
        std::lock_guard<std::mutex> g(g_modules_mu);
        auto it = g_host_allocs.find(p);
        if (it != g_host_allocs.end()) { len = it->second; g_host_allocs.erase(it); }
    }
    // a prepared key may sit inside the block: publish the range instead of locking every live module (a buffer drop on one thread
    // used to wait for whatever GPU call was in flight on every other thread's sibling module); the mirrors are dropped by their
    // owners at their next key lookup
    host_key_invalidate(p, len);
    (void)hipHostFree(p);
}
int pz_device_alloc(pz_module* M, size_t len, void** out) {
    if (!M || !out) return fail(PZ_ERR_INVALID, "null argument");
    PZ_HIP(hipSetDevice(M->device));
    PZ_HIP(hipMalloc(out, len ? len : 64));
    return PZ_OK;
}
int pz_device_free(pz_module* M, void* p) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    PZ_HIP(hipSetDevice(M->device));
    PZ_HIP(hipStreamSynchronize(M->stream));
    if (p) PZ_HIP(hipFree(p));
    return PZ_OK;
}
int pz_memcpy_h2d(pz_module* M, void* d, const void* s, size_t len) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    PZ_HIP(hipSetDevice(M->device));
    PZ_HIP(hipMemcpyAsync(d, s, len, hipMemcpyHostToDevice, M->stream));
    PZ_HIP(hipStreamSynchronize(M->stream));
    return PZ_OK;
}
int pz_memcpy_d2h(pz_module* M, void* d, const void* s, size_t len) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    PZ_HIP(hipSetDevice(M->device));
    PZ_HIP(hipMemcpyAsync(d, s, len, hipMemcpyDeviceToHost, M->stream));
    PZ_HIP(hipStreamSynchronize(M->stream));
    return PZ_OK;
}
int pz_memset_d(pz_module* M, void* d, int v, size_t len) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    PZ_HIP(hipSetDevice(M->device));
    PZ_HIP(hipMemsetAsync(d, v, len, M->stream));
    return PZ_OK;
}

size_t pz_bytes_of_vec_znx(uint64_t n, size_t cols, size_t size) { return (size_t)n * cols * size * 8; }
size_t pz_bytes_of_vec_znx_dft(uint64_t n, size_t cols, size_t size) { return (size_t)n * cols * size * 8; }
size_t pz_bytes_of_vec_znx_big(uint64_t n, size_t cols, size_t size) { return (size_t)n * cols * size * 8; }
size_t pz_bytes_of_svp_ppol(uint64_t n, size_t cols) { return (size_t)n * cols * 8; }
size_t pz_bytes_of_vmp_pmat(uint64_t n, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    return (size_t)n * rows * cols_in * cols_out * size * 8;
}

int pz_event_create(void** ev) {
    hipEvent_t e;
    PZ_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return PZ_OK;
}
int pz_event_destroy(void* ev) {
    PZ_HIP(hipEventDestroy((hipEvent_t)ev));
    return PZ_OK;
}
int pz_event_record(pz_module* M, void* ev) {
    if (!M) return fail(PZ_ERR_INVALID, "null module");
    std::lock_guard<std::mutex> lock_(M->mu);
    PZ_HIP(hipEventRecord((hipEvent_t)ev, M->stream));
    return PZ_OK;
}
int pz_event_elapsed_ms(void* e0, void* e1, float* ms) {
    PZ_HIP(hipEventSynchronize((hipEvent_t)e1));
    PZ_HIP(hipEventElapsedTime(ms, (hipEvent_t)e0, (hipEvent_t)e1));
    return PZ_OK;
}

// ------------------------------------------------------------------------------
// public: batched GLWE operations (device-resident)
// ------------------------------------------------------------------------------
struct OpShape {
    int cols_a, cols_in, cols_out;  // columns of `a`, VMP input columns, output columns
    int a_col0;                     // first column of `a` that enters the product
    int a_size_eff;                 // limbs of `a` in the key's base (after optional conversion)
    bool convert;
};
// kind: 0 external product, 1 key switch (mask columns 1.. of a GLWE), 2 tensor relinearization (operations/glwe.rs:541-607: `a` is
// a GLWETensor of cols + pairs columns, the pairs = rank (rank + 1) / 2 columns behind the first cols = rank + 1 are key-switched
// and the first cols are added to every column of the big value)
static OpShape op_shape(const pz_glwe_op_params* p, bool ks, bool tensor = false) {
    OpShape s;
    if (tensor) {
        const int cols = (int)p->rank + 1, pairs = (int)(p->rank * (p->rank + 1) / 2);
        s.cols_a = cols + pairs; s.cols_in = pairs; s.cols_out = cols; s.a_col0 = cols;
        s.convert = p->a_base2k != p->key_base2k;
        s.a_size_eff = s.convert ? (int)((p->a_size * p->a_base2k + p->key_base2k - 1) / p->key_base2k) : (int)p->a_size;
        return s;
    }
    s.a_col0 = ks ? 1 : 0;
    s.cols_a = (int)p->rank + 1;
    s.cols_in = ks ? (int)p->rank : (int)p->rank + 1;
    s.cols_out = ks ? (int)p->rank_out + 1 : (int)p->rank + 1;
    s.convert = p->a_base2k != p->key_base2k;
    s.a_size_eff = s.convert ? (int)((p->a_size * p->a_base2k + p->key_base2k - 1) / p->key_base2k) : (int)p->a_size;
    return s;
}

struct OpWs {
    size_t a_conv, a_dft, res_dft, tmp_dft, T, res_tmp, total;
};
static OpWs op_ws(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, size_t chunk, bool ks, bool au = false) {
    OpWs w;
    const size_t n8 = (size_t)M->n * 8;
    const size_t dsz = p->dsize;
    w.a_conv = s.convert ? align256(chunk * n8 * s.cols_a * s.a_size_eff) : 0;
    w.a_dft = align256(chunk * n8 * s.cols_in * (size_t)s.a_size_eff);
    w.res_dft = align256(chunk * n8 * s.cols_out * p->key_size);
    w.tmp_dft = dsz > 1 ? align256(chunk * n8 * (s.cols_out * p->key_size + (ks ? s.cols_in * (size_t)s.a_size_eff : 0))) : 0;
    const size_t tp = std::max((size_t)s.cols_in * s.a_size_eff, (size_t)s.cols_out * p->key_size);
    w.T = align256(chunk * tp * (size_t)M->m * sizeof(cplx));
    w.res_tmp = au ? align256(chunk * n8 * s.cols_out * p->res_size) : 0;  // normalized result before the final permutation
    w.total = w.a_conv + w.a_dft + w.res_dft + w.tmp_dft + w.T + w.res_tmp;
    return w;
}
static size_t pick_chunk(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, size_t batch) {
    if (M->chunk) return std::min(M->chunk, batch);
    // Measured on MI355X (profiles/r01_chunk_sweep.txt, r01_batch_sweep.txt): the intermediates do not stay in the Infinity
    // Cache anyway and every wave re-streams the key and pays the pipeline fill of the persistent middle kernel, so larger
    // waves win (128 -> 1024 ciphertexts per wave: +13 %); cap the workspace at ~24 GiB of the 288 GB.
    const size_t per_ct = (size_t)M->n * 8 * ((size_t)s.cols_in * s.a_size_eff + 2 * (size_t)s.cols_out * p->key_size);
    size_t c = ((size_t)24 << 30) / std::max<size_t>(per_ct, 1);
    c = std::max<size_t>(c & ~(size_t)7, 8);
    return std::min(c, batch);
}

// which pipeline glwe_op takes for a shape, and what it reserves there (one definition for the call and for the workspace query)
static bool fused_applies(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, bool ks, bool tensor, bool au) {
    const int npi = s.cols_in * s.a_size_eff, npo = s.cols_out * (int)p->key_size;
    const bool digits = p->dsize > 1, cross_out = p->res_base2k != p->key_base2k;
    (void)ks;
    return M->fuse_mid && M->fuse_tail && tail_supported(M) && mid_supported(M, npi, npo) && !(tensor && s.convert) &&
           (!(digits || cross_out) || (M->plan.m2 == 128 && !au && (int)p->dnum * s.cols_in <= 255 && npo <= 255));
}
struct FusedWs {
    size_t key, conv, t, t2, rtmp, small2, total;
};
static FusedWs fused_ws(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, size_t chunk, bool au) {
    FusedWs w;
    const size_t n8 = (size_t)M->n * 8, ksz = p->key_size;
    const size_t npi = (size_t)s.cols_in * s.a_size_eff, npo = (size_t)s.cols_out * ksz;
    w.key = align256((size_t)p->dnum * s.cols_in * npo * n8);
    w.conv = s.convert ? align256(chunk * n8 * s.cols_a * s.a_size_eff) : 0;
    w.t = align256(chunk * npi * M->m * sizeof(cplx));
    w.t2 = align256(chunk * npo * M->m * sizeof(cplx));
    // res_tmp holds the normalized result before the final permutation (mode 0 / gather scheme) OR, in the spectral form, the
    // body-column operand (min(a_size, key_size) limbs of one column): sized for the larger of the two
    const size_t body_limbs = std::min<size_t>((size_t)s.a_size_eff, ksz);
    w.rtmp = au ? align256(chunk * n8 * std::max((size_t)s.cols_out * p->res_size, body_limbs)) : 0;
    // cross-base output: the tail's key-base digits (cols_out x key_size limbs per ciphertext) before the cross-base pass
    w.small2 = p->res_base2k != p->key_base2k ? align256(chunk * n8 * s.cols_out * ksz) : 0;
    w.total = w.key + w.conv + w.t + w.t2 + w.rtmp + w.small2 + kMidDummyBytes + align256(M->ws_shift) + ((size_t)4 << 20);
    return w;
}
// keyswitch: 0 external product, 1 key switch, 2 automorphism family, 3 tensor relinearization.  The figure is what the call reserves
// in the module's grow-only workspace (+ the 12.5 % growth slack of its first allocation); a key that is neither pinned nor mirrored
// costs its row-sliced copy, which is included.
// N = 1024 / 2048: the two-kernel pipeline of device_small.hpp (plain products, key switches and the automorphism family; dsize 1, one
// base2k, <= 4 key limbs); `packed` = no OpLayout (the automorphism family needs it)
static bool small_ring_applies(const pz_module* M, const pz_glwe_op_params* p, const OpShape& s, bool ks, bool tensor, bool au, bool packed) {
    static const int small_env = getenv("POULPY_DBG_SMALL") ? atoi(getenv("POULPY_DBG_SMALL")) : 1;
    static const int small_au = getenv("POULPY_DBG_SMALL_AUTO") ? atoi(getenv("POULPY_DBG_SMALL_AUTO")) : 1;
    const bool cross_out = p->res_base2k != p->key_base2k;   // (with an automorphism: phi and the cross-base pass do not commute)
    return small_env && M->small_path && M->fuse_mid && M->fuse_tail && M->n < 4096 && (!au || (small_au && ks && packed && !cross_out)) &&
           !tensor && p->dsize == 1 && !M->probe && M->dbg_stages == 7 && small_supported(M, s.cols_in * s.a_size_eff, (int)p->key_size);
}

size_t pz_glwe_op_workspace_bytes(const pz_module* M, const pz_glwe_op_params* p, size_t batch, int keyswitch) {
    if (!M || !p || p->key_size == 0 || p->a_size == 0) return 0;
    const bool tensor = keyswitch == 3, ks = keyswitch != 0, au = keyswitch == 2;
    const OpShape s = op_shape(p, ks, tensor);
    const size_t chunk = pick_chunk(M, p, s, batch);
    size_t bytes;
    if (fused_applies(M, p, s, ks, tensor, au)) bytes = fused_ws(M, p, s, chunk, au).total;
    else if (small_ring_applies(M, p, s, ks, tensor, au, true))   // the key re-sliced + the spectra of one wave
        bytes = align256((size_t)p->dnum * s.cols_in * s.cols_out * p->key_size * (size_t)M->n * 8) +
                align256(chunk * (size_t)(s.cols_in * s.a_size_eff) * (size_t)M->m * sizeof(cplx)) +
                (s.convert ? align256(chunk * (size_t)M->n * 8 * s.cols_a * s.a_size_eff) : 0) +
                (p->res_base2k != p->key_base2k ? align256(chunk * (size_t)M->n * 8 * s.cols_out * p->key_size) : 0);
    else bytes = op_ws(M, p, s, chunk, ks, au).total;
    return bytes + (bytes >> 3);
}

// Automorphism family on top of the key switch (poulpy-core automorphism/glwe_ct.rs:51-275).  With phi = X -> X^p:
//   mode 0  res = phi(normalize(big))                       (:65-71)
//   mode 1  res = normalize(phi(big) + a)   (add, :133-138)   2: phi(big) - a (:222-227)   3: a - phi(big) (:268-273)
// where big is the key-switch value including the body (keyswitching/glwe.rs:236-237).  Normalization acts per
// coefficient, so modes 1-3 are computed as  phi(normalize'(s .* (big + small)))  with small = -+phi^-1(a) (+ body) built
// by one gather kernel, s(n) the sign phi gives coefficient n (applied inside the tail before the carry chain; flipped
// for mode 3) and a final sign-free permutation; mode 0 is the plain key switch followed by the signed permutation.
// (Round 2 experiment, removed — git history has it: a CU-partitioned, overlapped form of the fused pipeline.  With a CU mask spread
//  over the 8 XCDs (hipExtStreamCreateWithCUMask; POULPY_DBG_CU_MASK still runs the whole pipeline under one) pass 1 and the tail
//  keep their full rate down to 64 CUs while the middle kernel scales with its CU count (profiles/r02_cu_mask_scaling.txt), so chunk
//  c+1's pass 1, chunk c's middle kernel and chunk c-1's tail were run concurrently on disjoint CU sets, chained by events.
//  Bit-exact, but slower in every split tried (best 73 500/s with 8 + 8 CUs per XCD for the two streams against 88 700/s back to
//  back, profiles/r02_overlap_sweep.txt): under concurrency the three kernels share HBM at ~4.7 TB/s aggregate — no better than
//  the 4.7 TB/s the back-to-back sequence averages — and the three-deep chunk pipeline adds its fill / drain per call.)

int glwe_op(pz_module* M, bool ks, int64_t* res, const int64_t* a, const double* pmat, const pz_glwe_op_params* p, size_t batch,
                   const AutoSpec* au, const OpLayout* lay, bool tensor, bool* post_rsh) {
    const bool want_rsh = post_rsh && *post_rsh;
    if (post_rsh) *post_rsh = false;
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->dsize >= 1 && p->dnum >= 1 && p->key_size >= 1 && p->a_size >= 1 && p->res_size >= 1, "glwe op: empty shape");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a) && is_device_ptr(pmat), "batched entry points take device pointers");
    if (batch == 0) return PZ_OK;
    if (tensor) ks = true;   // the product is gglwe_product_dft, as for a key switch
    const OpShape s = op_shape(p, ks, tensor);
    const size_t chunk = pick_chunk(M, p, s, batch);
    const long long n = (long long)M->n;
    PZ_REQUIRE(!(tensor && (au || lay)), "glwe_tensor_relinearize: packed tensors, no automorphism");
    const int dsize = (int)p->dsize, dnum = (int)p->dnum, ksz = (int)p->key_size;
    const long long a_ct = n * s.cols_a * (long long)p->a_size;
    const long long res_ct = n * s.cols_out * (long long)p->res_size;
    const int npi = s.cols_in * s.a_size_eff, npo = s.cols_out * ksz;
    const int nrows = dnum * s.cols_in, ncols = s.cols_out * ksz;
    const bool au_big = au && au->mode != 0;
    const unsigned au_p = au ? (unsigned)((unsigned long long)au->p & (2ull * (unsigned long long)n - 1ull)) : 0u;
    const unsigned au_g = au ? inv_mod_2n(au->p, n) : 0u;
    const long long a_bs = lay ? lay->a_stride : a_ct, res_bs = lay ? lay->res_stride : res_ct;
    const int body_col = lay ? lay->body_col : 0;
    PZ_REQUIRE(!(au && lay), "glwe_automorphism: packed ciphertexts only");
    PZ_REQUIRE(body_col >= 0 && body_col < s.cols_out, "body column out of range");
    if (au) {
        PZ_REQUIRE(ks && s.cols_a == s.cols_out, "glwe_automorphism: the key must map rank -> rank");
        PZ_REQUIRE((au->p & 1) != 0, "glwe_automorphism: the Galois element must be odd");
        PZ_REQUIRE(au->mode >= 0 && au->mode <= 3, "glwe_automorphism: unknown mode");
    }

    // ---- fully fused pipeline: pass 1 (row-major) | row pass + VMP + inverse row pass | tail ----
    // dsize > 1 (digit-selected product inside the middle kernel) and res_base2k != key_base2k (the tail normalizes into the key's base,
    // one cross-base pass follows) ride on the same three kernels; both need the 128-point-row plans and no automorphism
    const bool digits = dsize > 1, cross_out = p->res_base2k != p->key_base2k;
    if (fused_applies(M, p, s, ks, tensor, au != nullptr)) {
        MidDigits dg;
        if (digits) {
            // external_product/glwe.rs:235-267, keyswitching/glwe.rs:332-379: limb l of `a` is digit di = (dsize - 1 - l) mod dsize, element
            // k = (l - (dsize - 1 - di)) / dsize of that digit's vector (vec_znx_dft_apply with step dsize, offset dsize - 1 - di); the
            // vector has (a_size + di) / dsize elements (at most dnum for a key switch) and multiplies key rows k (all input columns) with
            // limb_offset di, into a result of key_size - max(dsize - di - 2, 0) limbs (zero-tail semantics of SURVEY.md A.2)
            for (int l = 0; l < s.a_size_eff; ++l) {
                const int di = ((dsize - 1 - l) % dsize + dsize) % dsize;
                const int k = (l - (dsize - 1 - di)) / dsize;
                int a_sz = (s.a_size_eff + di) / dsize;
                if (ks) a_sz = std::min(a_sz, dnum);
                if (k < 0 || k >= a_sz || k >= dnum) continue;
                const int r_sz = ksz - std::max(dsize - di - 2, 0);
                const int off = di * s.cols_out;
                const int cb = off < ncols ? std::min(s.cols_out * r_sz, ncols - off) : 0;
                if (cb <= 0) continue;
                for (int c = 0; c < s.cols_in; ++c) {
                    dg.in[dg.n] = (unsigned char)(l * s.cols_in + c);
                    dg.row[dg.n] = (unsigned char)(k * s.cols_in + c);
                    dg.coff[dg.n] = (unsigned char)off;
                    dg.cb[dg.n] = (unsigned char)cb;
                    ++dg.n;
                }
            }
        }
        const FusedWs fw = fused_ws(M, p, s, chunk, au != nullptr);
        const size_t key_bytes = fw.key, conv_bytes = fw.conv, t_bytes = fw.t, t2_bytes = fw.t2, rtmp_bytes = fw.rtmp, small2_bytes = fw.small2;
        PZ_TRY(ws_reserve(M, fw.total));
        char* base = (char*)M->ws;
        cplx* Pp; int64_t* a_conv; cplx* T;
        PZ_TRY(ws_take(M, base, key_bytes, &Pp));
        PZ_TRY(ws_take(M, base, conv_bytes, &a_conv));
        PZ_TRY(ws_take(M, base, t_bytes, &T));
        // Placement of T2' relative to the result.  The tail of ciphertext b reads T2' + X and writes res + X and res + X + N*4 bytes
        // (the two coefficient halves), the same X for every workgroup; with both buffers on the same 1 MiB phase (large allocations
        // are 2 MiB aligned) the read and the two write streams of every workgroup meet on the same HBM channels: tail 3.55 ms per
        // 1024 ciphertexts in most processes, 3.14 in some, depending on the physical pages (profiles/r02_t2_placement.txt); the
        // middle kernel (T' -> T2') shows a smaller effect of the same kind.  Which phase is best depends on the placement too, so it is
        // MEASURED: after a warm-up call, the next kPhaseCount calls with a given argument set each run with one candidate phase and time middle kernel + tail
        // with two events (read at the next call); from then on the best one is used.  Results do not depend on the phase.
        static const size_t kPhase[] = {0xC0000, 0x40000, 0x80000, 0x140000, 0x1C0000, 0x240000, 0x340000, 0x3C0000};
        constexpr int kPhaseCount = (int)(sizeof(kPhase) / sizeof(kPhase[0]));
        int phase_idx = 0;
        pz_module::PhaseTune* tune = nullptr;
        bool tune_measure = false;
        static const int tune_env = getenv("POULPY_DBG_PHASE_TUNE") ? atoi(getenv("POULPY_DBG_PHASE_TUNE")) : 1;
        if (M->n >= 32768 && tune_env && M->phase_tuning) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(M->stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
            if (cs == hipStreamCaptureStatusNone) {
                KeyHash kh;
                kh.add(batch); kh.add(chunk); kh.add(npi); kh.add(npo); kh.add(*p); kh.add(ks);
                for (auto& e : M->phase_tune) if (e.key == kh.h) tune = &e;
                if (!tune) {
                    if (M->phase_tune.size() >= 16) {
                        size_t lru = 0;
                        for (size_t i = 1; i < M->phase_tune.size(); ++i) if (M->phase_tune[i].stamp < M->phase_tune[lru].stamp) lru = i;
                        for (int i = 0; i < 8; ++i) { (void)hipEventDestroy(M->phase_tune[lru].e0[i]); (void)hipEventDestroy(M->phase_tune[lru].e1[i]); }
                        M->phase_tune.erase(M->phase_tune.begin() + (long)lru);
                    }
                    pz_module::PhaseTune e{};
                    e.key = kh.h; e.calls = 0; e.best = -1; e.best_ms = 1e30f; e.pending = 0u; e.stamp = 0;
                    bool ok = true;
                    for (int i = 0; i < 8 && ok; ++i) ok = hipEventCreate(&e.e0[i]) == hipSuccess && hipEventCreate(&e.e1[i]) == hipSuccess;
                    if (ok) {
                        M->phase_tune.push_back(e);
                        tune = &M->phase_tune.back();
                    } else (void)hipGetLastError();
                }
                if (tune) {
                    static_assert(kPhaseCount == 8, "one event pair per candidate");
                    tune->stamp = ++M->phase_clock;
                    // measurements that have completed since the last call (never waited for)
                    for (int i = 0; i < kPhaseCount && tune->pending; ++i) {
                        if (!(tune->pending & (1u << i))) continue;
                        const hipError_t q = hipEventQuery(tune->e1[i]);
                        if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
                        float ms = 0.f;
                        if (q == hipSuccess && hipEventElapsedTime(&ms, tune->e0[i], tune->e1[i]) == hipSuccess) {
                            if (tune_env > 1) fprintf(stderr, "[phase tune] phase %#zx: %.3f ms\n", kPhase[i], ms);
                            if (ms < tune->best_ms) { tune->best_ms = ms; tune->best = i; }
                        } else (void)hipGetLastError();
                        tune->pending &= ~(1u << i);
                    }
                    // call 0 warms up (first touch of the workspace: not representative), calls 1 .. kPhaseCount try the candidates
                    // (under per-launch kernel timing nothing is measured: the instrumented pass uses what the plain calls found)
                    if (M->timing) phase_idx = tune->best >= 0 ? tune->best : 0;
                    else if (tune->calls == 0) { phase_idx = 0; tune->calls = 1; }
                    else if (tune->calls <= kPhaseCount) { phase_idx = tune->calls - 1; tune_measure = true; }
                    else phase_idx = tune->best >= 0 ? tune->best : 0;
                }
            }
        }
        base += ((size_t)kPhase[phase_idx] - (size_t)(((uintptr_t)base - (uintptr_t)res) & 0x3FFFFF)) & 0x3FFFFF;
        base += align256(M->ws_shift);
        cplx* T2; int64_t* res_tmp; int64_t* small2; cplx* mid_dummy;
        PZ_TRY(ws_take(M, base, t2_bytes, &T2));
        PZ_TRY(ws_take(M, base, rtmp_bytes, &res_tmp));
        PZ_TRY(ws_take(M, base, small2_bytes, &small2));
        PZ_TRY(ws_take(M, base, kMidDummyBytes, &mid_dummy));
        // the key arrives in the standard device layout; its row-sliced copy is rebuilt per call (2 x 128 MiB of
        // traffic at the metric shape, ~4 % of a 128-ciphertext call) so that no stale copy can ever be used
        bool pinned = false;
        for (auto& pk : M->pinned)
            if (pk.key == (const void*)pmat && pk.sliced && pk.bytes == (size_t)nrows * ncols * (size_t)M->n * 8) { Pp = pk.sliced; pinned = true; }
        if (!pinned && (M->dbg_stages & 2)) PZ_TRY(launch_permute_pmat(M, pmat, Pp, nrows * ncols));
        if (tune_measure) PZ_HIP(hipEventRecord(tune->e0[phase_idx], M->stream));
        for (size_t b0 = 0; b0 < batch; b0 += chunk) {
            const int nb = (int)std::min(chunk, batch - b0);
            DV av{(void*)(a + (long long)b0 * a_bs), a_bs, s.cols_a, (int)p->a_size};
            if (s.convert) {
                DV cv{a_conv, n * s.cols_a * s.a_size_eff, s.cols_a, s.a_size_eff};
                for (int c = 0; c < s.cols_a; ++c)
                    PZ_TRY(dev_normalize(M, nb, cv, (int)p->key_base2k, 0, c, av, (int)p->a_base2k, c));
                av = cv;
            }
            const int a_size = av.size;
            const int a_col0 = s.a_col0;
            PolyMap sm{a_size, s.cols_in, av.bs, (long long)av.cols * n, n, n * a_col0};
            // N = 4096, plain external product / key switch with <= 4 key limbs: two kernels, the spectra cross HBM once (device_small.hpp)
            static const int small_env = getenv("POULPY_DBG_SMALL") ? atoi(getenv("POULPY_DBG_SMALL")) : 1;
            static const int small_au4 = getenv("POULPY_DBG_SMALL_AUTO") ? atoi(getenv("POULPY_DBG_SMALL_AUTO")) : 1;
            // (round 3: the 8-slot tile of k_mid128r - 8 polynomials in, 8 out, 8 product rows: the external product with 4 limbs, BASELINE
            //  configs[1] - now beats the two-kernel form, 3.25 vs 3.16 M/s, profiles/r03_ab_small_vs_pipeline.txt; POULPY_DBG_SMALL=2 forces
            //  the two-kernel form there too)
            const bool mid8 = !ks && !au && npi == 8 && npo == 8 && std::min(nrows, npi) == 8 && small_env != 2;
            if (small_env && M->small_path && (!au || (small_au4 && ks && !lay)) && !tensor && !digits && !cross_out && !M->probe && M->dbg_stages == 7 &&
                small_supported(M, npi, ksz) && !mid8) {
                const bool rsh4 = want_rsh && au && au->mode != 0 && p->res_base2k <= 29;
                PZ_TRY(launch_small_fwd(M, nb * npi, (const long long*)av.p, sm, T));
                PZ_TRY(launch_small_inv(M, nb, T, Pp, npi, nrows, ncols, s.cols_out, ksz, (long long*)(res + (long long)b0 * res_bs), res_bs,
                                        s.cols_out, (int)p->res_size, ks ? (const long long*)av.p : nullptr, av.bs, s.cols_a, a_size,
                                        (int)p->res_base2k, body_col, false, nullptr, 0, au != nullptr, au_p, au ? au->mode : 0, rsh4));
                if (rsh4) *post_rsh = true;
                continue;
            }
            if (M->dbg_stages & 1) PZ_TRY(launch_fwd_pass1(M, nb * npi, (const long long*)av.p, sm, T, true));
            // X -> X^p with p = 1 mod 4 on the big value (the add / sub / sub_negate forms): DFT(phi(a))[q] = DFT(a)[p q + (p-1)/4 mod m]
            // is an affine map of the spectrum index that sends rows of the four-step layout to rows, so the middle kernel writes
            // its product at the permuted position (k_mid128<.., PERM>) and the tail's inverse transform is phi(big) itself.  The
            // tail then adds ONE operand stream per column at the natural index -- +-a[col], and on the body column
            // +-(phi(body) +- a0) prepared by one k_automorphism pass over that column into the (cache-resident) workspace -- and
            // writes the final result: no permutation pass over the result, no gathers in the tail, in-place forms safe.
            static const int au_spec = getenv("POULPY_DBG_AUTO_SPECTRAL") ? atoi(getenv("POULPY_DBG_AUTO_SPECTRAL")) : 1;
            // (mode 0, phi(normalize(big)), rides on the same form: the tail undoes phi's signs in front of the carry chain and puts them
            //  back on the digits; POULPY_DBG_AUTO_SPECTRAL=2 keeps the key switch + signed permutation pass for it)
            // (round 3) p = 3 mod 4 too - X -> X^-1, the first step of every trace, among them: the spectrum of phi(a) is then the CONJUGATE of
            // a permuted spectrum (MidArgs::perm_ysign); POULPY_DBG_AUTO_SPECTRAL=3 keeps those Galois elements on the older path
            const bool spec = au_spec && au && (au_big || au_spec == 1 || au_spec == 3) && ((au_p & 3u) == 1u || au_spec != 3) && M->plan.m2 == 128 &&
                              M->dbg_stages == 7;
            unsigned perm_mul = 0, perm_add = 0;
            bool perm_conj = false;
            if (spec) {
                const unsigned mm = (unsigned)M->m;
                if ((au_p & 3u) == 1u) {
                    perm_mul = au_g & (mm - 1u);
                    const unsigned long long c0 = (unsigned long long)(((au_p - 1u) >> 2) & (mm - 1u));
                    perm_add = (unsigned)((mm - (unsigned)(((unsigned long long)perm_mul * c0) & (unsigned long long)(mm - 1u))) & (mm - 1u));
                } else {
                    perm_conj = true;
                    perm_mul = (mm - (au_g & (mm - 1u))) & (mm - 1u);                                   // (-p)^-1 mod m
                    const unsigned long long c0 = (unsigned long long)((((unsigned long long)au_p + 1ull) >> 2) & (unsigned long long)(mm - 1u));
                    perm_add = (unsigned)(((unsigned long long)perm_mul * c0) & (unsigned long long)(mm - 1u));   // (-p)^-1 (p + 1)/4
                }
            }
            if (digits && dg.n == 0) {   // nothing reaches the product (e.g. dsize > a.size): the big value is the body alone
                PZ_HIP(hipMemsetAsync(T2, 0, (size_t)nb * npo * M->m * sizeof(cplx), M->stream));
            } else if (M->dbg_stages & 2)
                PZ_TRY(launch_mid(M, nb, T, T2, Pp, npi, npo, nrows, ncols, mid_dummy, perm_mul, perm_add, digits ? &dg : nullptr, nullptr, perm_conj));
            int64_t* res_b = res + (long long)b0 * res_bs;
            if (spec) {
                // (the tail reads operand limbs j < min(key_size, a_size) only: the pre-pass covers exactly those)
                const int bl = std::min(a_size, ksz);
                PolyMap bsm{bl, 1, av.bs, (long long)av.cols * n, 0, 0}, bdm{bl, 1, (long long)bl * n, n, 0, 0};
                if (!au_big) {   // plain form: the body column's operand is phi(body); no other operand
                    PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, (long long*)res_tmp, bdm, au_g, 1));
                    PZ_TRY(launch_inv_tail(M, nb, T2, ksz, s.cols_out, (long long*)res_b, res_bs, s.cols_out, (int)p->res_size,
                                           (const long long*)av.p, av.bs, s.cols_a, a_size, (int)p->res_base2k, true, true, au_g, false, 0u, false,
                                           body_col, (const long long*)res_tmp, (long long)bl * n, n, false, false, true, true));
                    continue;
                }
                // operand of the body column, one stream: phi(body) + a0 (add) or -phi(body) + a0 (sub forms: the tail negates every operand)
                PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, (long long*)res_tmp, bdm, au_g, au->mode == 1 ? 1 : 3,
                                           (const long long*)av.p, bsm));
                const bool rsh = want_rsh && tail_rsh_supported(M) && !cross_out && p->res_base2k <= 29;   // (32-bit shift steps: device_fft.hpp)
                PZ_TRY(launch_inv_tail(M, nb, T2, ksz, s.cols_out, (long long*)res_b, res_bs, s.cols_out, (int)p->res_size,
                                       (const long long*)av.p, av.bs, s.cols_a, a_size, (int)p->res_base2k, true, true,
                                       au->mode == 3 ? 2u * (unsigned)n : 0u, au->mode == 3, 0u, false, body_col, (const long long*)res_tmp,
                                       (long long)bl * n, n, au->mode != 1, rsh));
                if (rsh) *post_rsh = true;
                continue;
            }
            const long long* small = ks ? (const long long*)av.p : nullptr;
            long long small_bs = av.bs;
            // au_big: the operand -+phi^-1(a) (+ body for column 0) is gathered from `a` inside the tail (TailArgs::gather_mul)
            (void)small2;
            if (cross_out) {
                // vec_znx_big_normalize(res_base2k <- key_base2k) in two exact steps: the tail's carry chain writes balanced key-base digits
                // (all key_size limbs: nothing is dropped), the cross-base kernel converts them.  Both steps are functions of the torus
                // value only, so the result equals the reference's single cross-base pass over the big value (checked on the oracle over
                // thousands of random shapes / edge digits, and by the parity tests).
                const long long tmp_ct = n * s.cols_out * (long long)ksz;
                PZ_TRY(launch_inv_tail(M, nb, T2, ksz, s.cols_out, (long long*)small2, tmp_ct, s.cols_out, ksz, small, small_bs, s.cols_a, a_size,
                                       (int)p->key_base2k, true, tensor, 0u, false, 0u, false, body_col));
                DV tv{small2, tmp_ct, s.cols_out, ksz}, rv{res_b, res_bs, s.cols_out, (int)p->res_size};
                for (int c = 0; c < s.cols_out; ++c)
                    PZ_TRY(dev_normalize(M, nb, rv, (int)p->res_base2k, 0, c, tv, (int)p->key_base2k, c));
                continue;
            }
            if (M->dbg_stages & 4) PZ_TRY(launch_inv_tail(M, nb, T2, ksz, s.cols_out, (long long*)(au ? res_tmp : res_b), au ? res_ct : res_bs, s.cols_out, (int)p->res_size,
                                   small, small_bs, s.cols_a, a_size, (int)p->res_base2k, true, au_big || tensor, au_big ? au_p : 0u, au && au->mode == 3,
                                   au_big ? au_p : 0u, au_big && au->mode != 1, body_col));
            if (au) {
                PolyMap tm{(int)p->res_size, s.cols_out, res_ct, (long long)s.cols_out * n, n, 0};
                PZ_TRY(launch_automorphism(M, nb * (int)p->res_size * s.cols_out, (const long long*)res_tmp, tm, (long long*)res_b, tm, au_g,
                                           au->mode == 0 ? 1 : 0));
            }
        }
        if (tune_measure) {
            PZ_HIP(hipEventRecord(tune->e1[phase_idx], M->stream));
            tune->pending |= 1u << phase_idx;
            tune->calls++;
        }
        return PZ_OK;
    }

    // ---- N = 1024 / 2048: no pipeline plan (their per-op split is 16 x 32 / 32 x 32), but whole polynomials fit LDS: the two-kernel
    // pipeline of device_small.hpp with its own m = M1 x 128 tables.  Plain external product / key switch, dsize 1, one base2k, <= 4 key
    // limbs; anything else stays on the five-kernel path below ----
    {
        // (the automorphism family too: phi is an index / sign map inside the inverse kernel's carry-chain stage)
        if (small_ring_applies(M, p, s, ks, tensor, au != nullptr, lay == nullptr)) {
            // mixed bases as in the fused pipeline: `a` re-expressed in the key's base first (external_product/glwe.rs:124-132); a result in
            // another base = balanced key-base digits from the inverse kernel (all key limbs), then one cross-base pass (same two exact
            // steps as the three-kernel tail)
            const size_t n8 = (size_t)M->n * 8;
            const size_t key_bytes = align256((size_t)nrows * ncols * n8), s_bytes = align256(chunk * npi * (size_t)M->m * sizeof(cplx));
            const size_t conv_bytes = s.convert ? align256(chunk * n8 * s.cols_a * s.a_size_eff) : 0;
            const size_t tmp_bytes = cross_out ? align256(chunk * n8 * s.cols_out * ksz) : 0;
            PZ_TRY(ws_reserve(M, key_bytes + s_bytes + conv_bytes + tmp_bytes));
            char* sbase = (char*)M->ws;
            cplx* Pp; cplx* S; int64_t* a_conv; int64_t* key_digits;
            PZ_TRY(ws_take(M, sbase, key_bytes, &Pp));
            PZ_TRY(ws_take(M, sbase, s_bytes, &S));
            PZ_TRY(ws_take(M, sbase, conv_bytes, &a_conv));
            PZ_TRY(ws_take(M, sbase, tmp_bytes, &key_digits));
            {   // a pinned (or mirrored) key brings its permuted copy along
                bool pinned = false;
                for (auto& pk : M->pinned)
                    if (pk.key == (const void*)pmat && pk.sliced && pk.bytes == (size_t)nrows * ncols * n8) { Pp = pk.sliced; pinned = true; }
                if (!pinned) PZ_TRY(launch_small_permute(M, pmat, Pp, nrows * ncols));
            }
            const bool small_rsh = want_rsh && au && au->mode != 0 && !cross_out && p->res_base2k <= 29;
            for (size_t b0 = 0; b0 < batch; b0 += chunk) {
                const int nb = (int)std::min(chunk, batch - b0);
                DV av{(void*)(a + (long long)b0 * a_bs), a_bs, s.cols_a, (int)p->a_size};
                if (s.convert) {
                    DV cv{a_conv, n * s.cols_a * s.a_size_eff, s.cols_a, s.a_size_eff};
                    for (int c = 0; c < s.cols_a; ++c) PZ_TRY(dev_normalize(M, nb, cv, (int)p->key_base2k, 0, c, av, (int)p->a_base2k, c));
                    av = cv;
                }
                PolyMap sm{av.size, s.cols_in, av.bs, (long long)av.cols * n, n, n * s.a_col0};
                PZ_TRY(launch_small_fwd(M, nb * npi, (const long long*)av.p, sm, S));
                int64_t* res_b = res + (long long)b0 * res_bs;
                if (cross_out) {
                    const long long tmp_ct = n * s.cols_out * (long long)ksz;
                    PZ_TRY(launch_small_inv(M, nb, S, Pp, npi, nrows, ncols, s.cols_out, ksz, (long long*)key_digits, tmp_ct, s.cols_out, ksz,
                                            ks ? (const long long*)av.p : nullptr, av.bs, s.cols_a, av.size, (int)p->key_base2k, body_col));
                    DV tv{key_digits, tmp_ct, s.cols_out, ksz}, rv{res_b, res_bs, s.cols_out, (int)p->res_size};
                    for (int c = 0; c < s.cols_out; ++c) PZ_TRY(dev_normalize(M, nb, rv, (int)p->res_base2k, 0, c, tv, (int)p->key_base2k, c));
                    continue;
                }
                PZ_TRY(launch_small_inv(M, nb, S, Pp, npi, nrows, ncols, s.cols_out, ksz, (long long*)res_b, res_bs, s.cols_out, (int)p->res_size,
                                        ks ? (const long long*)av.p : nullptr, av.bs, s.cols_a, av.size, (int)p->res_base2k, body_col, false, nullptr,
                                        0, au != nullptr, au_p, au ? au->mode : 0, small_rsh));
            }
            if (small_rsh) *post_rsh = true;
            return PZ_OK;
        }
    }

    const OpWs w = op_ws(M, p, s, chunk, ks, au != nullptr);
    PZ_TRY(ws_reserve(M, w.total));
    char* base = (char*)M->ws;
    int64_t* a_conv; double* a_dft; double* res_dft; double* tmp_dft; cplx* T; int64_t* res_tmp;
    PZ_TRY(ws_take(M, base, w.a_conv, &a_conv));
    PZ_TRY(ws_take(M, base, w.a_dft, &a_dft));
    PZ_TRY(ws_take(M, base, w.res_dft, &res_dft));
    PZ_TRY(ws_take(M, base, w.tmp_dft, &tmp_dft));
    PZ_TRY(ws_take(M, base, w.T, &T));
    PZ_TRY(ws_take(M, base, w.res_tmp, &res_tmp));

    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = (int)std::min(chunk, batch - b0);
        DV av{(void*)(a + (long long)b0 * a_bs), a_bs, s.cols_a, (int)p->a_size};
        if (s.convert) {  // glwe_normalize into the key's base (external_product/glwe.rs:124-132)
            DV cv{a_conv, n * s.cols_a * s.a_size_eff, s.cols_a, s.a_size_eff};
            for (int c = 0; c < s.cols_a; ++c)
                PZ_TRY(dev_normalize(M, nb, cv, (int)p->key_base2k, 0, c, av, (int)p->a_base2k, c));
            av = cv;
        }
        const DV raw_av{(void*)(a + (long long)b0 * a_bs), a_bs, s.cols_a, (int)p->a_size};
        const int a_size = av.size;
        const int a_col0 = s.a_col0;  // key-switch transforms the mask columns only (keyswitching/glwe.rs:231-234)
        DV rd{res_dft, n * s.cols_out * ksz, s.cols_out, ksz};
        int res_dft_size = ksz;
        if (dsize == 1) {
            DV ad{a_dft, n * s.cols_in * a_size, s.cols_in, a_size};
            PZ_TRY(dev_dft_apply(M, nb, 1, 0, ad, 0, av, a_col0, s.cols_in, nullptr, T));
            PZ_TRY(dev_vmp(M, nb, rd, ad, pmat, dnum, s.cols_in, s.cols_out, ksz, 0));
        } else {
            // external_product/glwe.rs:235-267 ; keyswitching/glwe.rs:332-379
            // res_dft starts zeroed (glwe.rs:122): limbs skipped by the first iterations are only ever added to
            PZ_HIP(hipMemsetAsync(res_dft, 0, (size_t)nb * rd.bs * 8, M->stream));
            DV td{tmp_dft, n * s.cols_out * ksz, s.cols_out, ksz};
            for (int di = 0; di < dsize; ++di) {
                int a_sz = (a_size + di) / dsize;
                if (ks) a_sz = std::min(a_sz, dnum);
                const int drop = std::max(dsize - di - 2, 0);
                res_dft_size = ksz - drop;
                DV ad{a_dft, n * s.cols_in * a_sz, s.cols_in, a_sz};
                PZ_TRY(dev_dft_apply(M, nb, dsize, dsize - 1 - di, ad, 0, av, a_col0, s.cols_in, nullptr, T));
                DV rdi{res_dft, rd.bs, s.cols_out, res_dft_size};
                if (di == 0) {
                    PZ_TRY(dev_vmp(M, nb, rdi, ad, pmat, dnum, s.cols_in, s.cols_out, ksz, 0));
                } else {
                    DV tdi{tmp_dft, td.bs, s.cols_out, res_dft_size};
                    PZ_TRY(dev_vmp(M, nb, tdi, ad, pmat, dnum, s.cols_in, s.cols_out, ksz, di));
                    PZ_TRY(launch_ew(M, EW_ADD, res_dft, rd.bs, n, res_dft, rd.bs, n, tmp_dft, td.bs, n, s.cols_out * res_dft_size, nb));
                }
            }
            if (ks) res_dft_size = ksz;  // keyswitching/glwe.rs:378 res.set_size(res.max_size())
            if (ks && dsize > 2) {
                // limbs dropped by the last iterations keep the value of the earlier ones (reference behaviour); nothing to do
            }
        }
        DV rb{res_dft, rd.bs, s.cols_out, res_dft_size};
        DV rv{(void*)(res + (long long)b0 * res_bs), res_bs, s.cols_out, (int)p->res_size};
        if (au) {
            // op-by-op, as the reference: big value, body, [automorphism of the big value, +- a], normalize, [automorphism]
            PZ_TRY(dev_idft(M, nb, rb, 0, rb, 0, s.cols_out, res_dft_size, T));
            const long long big_ls = (long long)s.cols_out * n, a_ls = (long long)av.cols * n;
            PZ_TRY(launch_ew(M, EW_ADD_I64, res_dft, rb.bs, big_ls, res_dft, rb.bs, big_ls, av.p, av.bs, a_ls, std::min(res_dft_size, a_size), nb));
            DV nsrc = rb;
            if (au_big) {
                int64_t* big2 = (int64_t*)T;  // free again: same bytes as the big value
                PolyMap bm{res_dft_size, s.cols_out, rb.bs, big_ls, n, 0};
                PZ_TRY(launch_automorphism(M, nb * res_dft_size * s.cols_out, (const long long*)res_dft, bm, (long long*)big2, bm, au_g, 1));
                const int sum = std::min(res_dft_size, a_size);
                for (int c = 0; c < s.cols_out; ++c) {
                    int64_t* bc = big2 + (long long)c * n;
                    const int64_t* ac = (const int64_t*)av.p + (long long)c * n;
                    if (au->mode == 1) PZ_TRY(launch_ew(M, EW_ADD_I64, bc, rb.bs, big_ls, bc, rb.bs, big_ls, ac, av.bs, a_ls, sum, nb));
                    else if (au->mode == 2) PZ_TRY(launch_ew(M, EW_SUB_I64, bc, rb.bs, big_ls, bc, rb.bs, big_ls, ac, av.bs, a_ls, sum, nb));
                    else {  // a - big, and -big where a has no limb (vec_znx/sub.rs:84-110)
                        PZ_TRY(launch_ew(M, EW_SUB_I64, bc, rb.bs, big_ls, ac, av.bs, a_ls, bc, rb.bs, big_ls, sum, nb));
                        PZ_TRY(launch_ew(M, EW_NEG_I64, bc + (long long)sum * big_ls, rb.bs, big_ls, bc + (long long)sum * big_ls, rb.bs, big_ls,
                                         nullptr, 0, 0, res_dft_size - sum, nb));
                    }
                }
                nsrc = DV{big2, rb.bs, s.cols_out, res_dft_size};
            }
            DV nd = au->mode == 0 ? DV{res_tmp, res_ct, s.cols_out, (int)p->res_size} : rv;
            for (int c = 0; c < s.cols_out; ++c)
                PZ_TRY(dev_normalize(M, nb, nd, (int)p->res_base2k, 0, c, nsrc, (int)p->key_base2k, c));
            if (au->mode == 0) {
                PolyMap tm{(int)p->res_size, s.cols_out, res_ct, (long long)s.cols_out * n, n, 0};
                PZ_TRY(launch_automorphism(M, nb * (int)p->res_size * s.cols_out, (const long long*)res_tmp, tm, (long long*)rv.p, tm, au_g, 1));
            }
        } else if (p->res_base2k == p->key_base2k && M->fuse_tail && tail_supported(M)) {
            // inverse pass 2, then the fused tail: inverse pass 1 + body add + carry chain, no VecZnxBig in HBM
            PolyMap sm{res_dft_size, s.cols_out, rb.bs, (long long)s.cols_out * n, n, 0};
            PZ_TRY(launch_inv_pass2(M, nb * res_dft_size * s.cols_out, res_dft, sm, T));
            // (tensor: every column receives its operand; with a conversion the reference still adds the UN-normalized a when
            //  res_base2k == key_base2k, operations/glwe.rs:588-592)
            const DV& sv = tensor ? raw_av : av;
            PZ_TRY(launch_inv_tail(M, nb, T, res_dft_size, s.cols_out, (long long*)rv.p, rv.bs, rv.cols, rv.size,
                                   ks ? (const long long*)sv.p : nullptr, sv.bs, sv.cols, sv.size, (int)p->res_base2k, false, tensor, 0, false, 0,
                                   false, body_col));
        } else {
            PZ_TRY(dev_idft(M, nb, rb, 0, rb, 0, s.cols_out, res_dft_size, T));
            if (tensor) {  // operations/glwe.rs:588-598: + a[col] on every column (raw a when res_base2k == key_base2k, else the converted one)
                const DV& sv = p->res_base2k == p->key_base2k ? raw_av : av;
                for (int c = 0; c < s.cols_out; ++c)
                    PZ_TRY(launch_ew(M, EW_ADD_I64, res_dft + (long long)c * n, rb.bs, (long long)s.cols_out * n, res_dft + (long long)c * n, rb.bs,
                                     (long long)s.cols_out * n, (const int64_t*)sv.p + (long long)c * n, sv.bs, (long long)sv.cols * n,
                                     std::min(res_dft_size, sv.size), nb));
            } else if (ks)  // body column added after the inverse transform (keyswitching/glwe.rs:237)
                PZ_TRY(launch_ew(M, EW_ADD_I64, res_dft + (long long)body_col * n, rb.bs, (long long)s.cols_out * n,
                                 res_dft + (long long)body_col * n, rb.bs, (long long)s.cols_out * n, av.p, av.bs, (long long)av.cols * n,
                                 std::min(res_dft_size, a_size), nb));
            for (int c = 0; c < s.cols_out; ++c)
                PZ_TRY(dev_normalize(M, nb, rv, (int)p->res_base2k, 0, c, rb, (int)p->key_base2k, c));
        }
    }
    return PZ_OK;
}

// The four GLWE-level entry points accept device pointers (batched, device-resident: the measured path) or HOST containers
// (what a CoreImpl override of the Rust shim passes): host ciphertexts are staged, a host-resident prepared key is mirrored on
// the device (resolve_key); the call is then logically synchronous like every host-pointer call.
static int glwe_entry(pz_module* M, bool ks, bool tensor, int64_t* res, const int64_t* a, const double* pmat, const pz_glwe_op_params* p,
                      size_t batch, const AutoSpec* au) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->dsize >= 1 && p->dnum >= 1 && p->key_size >= 1 && p->a_size >= 1 && p->res_size >= 1, "glwe op: empty shape");
    const OpShape s = op_shape(p, ks || tensor, tensor);
    const size_t n8 = (size_t)M->n * 8;
    GlweArgs g;
    PZ_TRY(glwe_args_in(M, g, res, a, pmat, batch * n8 * s.cols_out * p->res_size, batch * n8 * s.cols_a * p->a_size,
                        n8 * p->dnum * s.cols_in * s.cols_out * p->key_size));
    PZ_TRY(glwe_op(M, ks, g.res, g.a, g.key, p, batch, au, nullptr, tensor));
    return glwe_args_out(M, g);
}
int pz_glwe_external_product_batched(pz_module* M, int64_t* res, const int64_t* a, const double* ggsw_pmat,
                                     const pz_glwe_op_params* p, size_t batch) {
    PZ_ENTER(M);
    return glwe_entry(M, false, false, res, a, ggsw_pmat, p, batch, nullptr);
}
int pz_glwe_keyswitch_batched(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p,
                              size_t batch) {
    PZ_ENTER(M);
    return glwe_entry(M, true, false, res, a, key_pmat, p, batch, nullptr);
}
int pz_glwe_automorphism_batched(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p,
                                 int64_t gal, int mode, size_t batch) {
    PZ_ENTER(M);
    AutoSpec au{(long long)gal, mode};
    return glwe_entry(M, true, false, res, a, key_pmat, p, batch, &au);
}
// glwe_tensor_relinearize (poulpy-core/src/operations/glwe.rs:541-607) on `batch` GLWETensors sharing one prepared tensor key
int pz_glwe_tensor_relinearize_batched(pz_module* M, int64_t* res, const int64_t* a, const double* tsk_pmat, const pz_glwe_op_params* p,
                                       size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->rank >= 1 && p->rank_out == p->rank, "glwe_tensor_relinearize: the tensor key maps rank (rank + 1) / 2 -> rank");
    return glwe_entry(M, true, true, res, a, tsk_pmat, p, batch, nullptr);
}
int pz_module_forget_host_key(pz_module* M, const double* host_pmat) {
    PZ_ENTER(M);
    return forget_host_key(M, (const void*)host_pmat);
}
size_t pz_module_host_key_mirrors(pz_module* M) {
    if (!M) return 0;
    std::lock_guard<std::mutex> lock_(M->mu);
    // mirrors whose host range has been invalidated since (possibly by another module or by pz_free_bytes) are released now
    (void)hipSetDevice(M->device);
    for (size_t i = M->mirrors.size(); i-- > 0;) {
        uint64_t now = 0;
        if (host_key_stale(M->mirrors[i].host, M->mirrors[i].bytes, M->mirrors[i].epoch, &now)) {
            (void)hipStreamSynchronize(M->stream);
            drop_mirror_at(M, i);
        } else M->mirrors[i].epoch = now;
    }
    return M->mirrors.size();
}
// ggsw_external_product (external_product/ggsw.rs:54-58): every (row, column) entry of the GGSW `a` is a GLWE and the entries
// are contiguous in the MatZnx layout, so the operation is one batched external product over dnum_a * (rank+1) ciphertexts
int pz_ggsw_external_product(pz_module* M, int64_t* res, const int64_t* a, size_t a_dnum, const double* ggsw_pmat,
                             const pz_glwe_op_params* p) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    return glwe_op(M, false, res, a, ggsw_pmat, p, a_dnum * (p->rank + 1));
}

// ggsw_expand_row (conversion/gglwe_to_ggsw.rs:116-268): column `col` >= 1 of every row is the key switch of the mask of
// res.at(row, 0) by tsk.at(col - 1), with the body of res.at(row, 0) added to column `col` of the big value before the
// normalization.  The entries (row, 0) of `count` contiguous GGSWs are `count * dnum` ciphertexts at a fixed stride, so
// each column is one batched key switch; column 0 is left untouched.
int ggsw_expand_row(pz_module* M, int64_t* ggsw, size_t dnum, const double* const* tsk_pmat, const pz_glwe_op_params* p, size_t count) {
    PZ_REQUIRE(p != nullptr && tsk_pmat != nullptr, "null params");
    PZ_REQUIRE(p->a_size == p->res_size && p->a_base2k == p->res_base2k, "ggsw_expand_row: a and res describe the same GGSW");
    PZ_REQUIRE(dnum >= 1, "ggsw_expand_row: empty GGSW");
    const size_t cols = p->rank + 1;
    const long long ct = (long long)M->n * (long long)cols * (long long)p->res_size;
    for (size_t col = 1; col < cols; ++col) {
        PZ_REQUIRE(tsk_pmat[col - 1] != nullptr, "ggsw_expand_row: null tensor key");
        OpLayout lay{ct * (long long)cols, ct * (long long)cols, (int)col};
        PZ_TRY(glwe_op(M, true, ggsw + (long long)col * ct, ggsw, tsk_pmat[col - 1], p, count * dnum, nullptr, &lay));
    }
    return PZ_OK;
}
int pz_ggsw_expand_row_batched(pz_module* M, int64_t* ggsw, size_t dnum, const double* const* tsk_pmat, const pz_glwe_op_params* p,
                               size_t count) {
    PZ_ENTER(M);
    return ggsw_expand_row(M, ggsw, dnum, tsk_pmat, p, count);
}

// ggsw_from_gglwe (conversion/gglwe_to_ggsw.rs:32-61): entries (row, 0) of the GGSW are copies of the entries (row, 0) of
// the GGLWE `a` (glwe_copy), then ggsw_expand_row.  `count` contiguous GGLWEs -> `count` contiguous GGSWs, one strided copy.
int pz_ggsw_from_gglwe_batched(pz_module* M, int64_t* ggsw, const int64_t* a, size_t a_cols_in, size_t dnum,
                               const double* const* tsk_pmat, const pz_glwe_op_params* p, size_t count) {
    PZ_ENTER(M);
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(is_device_ptr(ggsw) && is_device_ptr(a), "batched entry points take device pointers");
    PZ_REQUIRE(a_cols_in >= 1 && dnum >= 1, "ggsw_from_gglwe: empty GGLWE");
    PZ_REQUIRE((const void*)ggsw != (const void*)a, "ggsw_from_gglwe: res must not alias a");
    const size_t cols = p->rank + 1;
    const long long n = (long long)M->n, ct = n * (long long)cols * (long long)p->res_size;
    PZ_TRY(launch_ew(M, EW_COPY, ggsw, (long long)cols * ct, n, a, (long long)a_cols_in * ct, n, nullptr, 0, 0, (int)(cols * p->res_size),
                     (int)(count * dnum)));
    return ggsw_expand_row(M, ggsw, dnum, tsk_pmat, p, count);  // (the module lock is not recursive)
}

// glwe_trace_assign (poulpy-core/src/glwe_trace.rs:129-176) on `batch` ciphertexts:
//   for every step s:  res = rsh(res, 1 bit) on every column (operations/glwe.rs:1096-1112);  res = glwe_automorphism_add_assign(res, key_s)
// res in another base than the keys (:153-163; test_suite/trace.rs:36-39): (a_size, a_base2k = key_base2k) describe res re-expressed in
// the keys' base (a_size = ceil(res.max_k / key_base2k)); normalize into a temporary of that layout, trace there, normalize back.
int glwe_trace(pz_module* M, int64_t* res, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                      const pz_glwe_op_params* p, size_t batch) {
    PZ_REQUIRE(p != nullptr && (nsteps == 0 || (gals != nullptr && key_pmats != nullptr)), "glwe_trace: null argument");
    PZ_REQUIRE(p->rank_out == p->rank, "glwe_trace: rank_out != rank");
    PZ_REQUIRE(is_device_ptr(res), "batched entry points take device pointers");
    if (p->res_base2k != p->key_base2k) {
        PZ_REQUIRE(p->a_base2k == p->key_base2k && p->a_size >= 1 && p->res_size >= 1,
                   "glwe_trace: with res in another base than the keys, (a_size, a_base2k) is its layout in the keys' base");
        if (batch == 0) return PZ_OK;
        const long long n = (long long)M->n;
        const int cols = (int)p->rank + 1, B = (int)batch;
        const long long ct_c = n * cols * (long long)p->a_size, ct_r = n * cols * (long long)p->res_size;
        PZ_TRY(ws2_reserve(M, (size_t)B * ct_c * 8));
        int64_t* conv = (int64_t*)M->ws2;
        DV cv{conv, ct_c, cols, (int)p->a_size}, rv{res, ct_r, cols, (int)p->res_size};
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, cv, (int)p->key_base2k, 0, c, rv, (int)p->res_base2k, c));
        pz_glwe_op_params q = *p;
        q.res_size = p->a_size; q.res_base2k = p->key_base2k;
        PZ_TRY(glwe_trace(M, conv, nsteps, gals, key_pmats, &q, batch));
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, rv, (int)p->res_base2k, 0, c, cv, (int)p->key_base2k, c));
        return PZ_OK;
    }
    PZ_REQUIRE(p->a_size == p->res_size && p->a_base2k == p->res_base2k,
               "glwe_trace: a and res describe the same ciphertexts when res is in the keys' base");
    if (batch == 0) return PZ_OK;
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1;
    const long long ct = n * cols * (long long)p->res_size;
    // the one-bit shift in front of step s + 1 rides on the tail of step s where that path has the shifted-store variant
    // (POULPY_DBG_TRACE_RSH=0: always the separate pass)
    static const int fuse_rsh = getenv("POULPY_DBG_TRACE_RSH") ? atoi(getenv("POULPY_DBG_TRACE_RSH")) : 1;
    bool shifted = false;
    for (size_t s = 0; s < nsteps; ++s) {
        PZ_REQUIRE((gals[s] & 1) != 0, "glwe_trace: Galois elements must be odd");
        if (!shifted) PZ_TRY(launch_rsh(M, (int)batch, (long long*)res, ct, cols, (int)p->res_size, 0, cols, (int)p->res_base2k, 1));
        AutoSpec au{(long long)gals[s], 1};
        bool rsh = fuse_rsh && s + 1 < nsteps;
        PZ_TRY(glwe_op(M, true, res, res, key_pmats[s], p, batch, &au, nullptr, false, &rsh));
        shifted = rsh;
    }
    return PZ_OK;
}

int pz_glwe_trace_batched(pz_module* M, int64_t* res, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                          const pz_glwe_op_params* p, size_t batch) {
    PZ_ENTER(M);
    KeyHash k;
    k.add((int)1); k.add(res); k.add(nsteps); k.add(batch);
    if (p) k.add(*p);
    for (size_t s = 0; s < nsteps && gals && key_pmats; ++s) { k.add(gals[s]); k.add(key_pmats[s]); }
    graph_key_module(M, k);
    return with_graph(M, k.h, [&]() { return glwe_trace(M, res, nsteps, gals, key_pmats, p, batch); });
}

}  // extern "C"

// api_lwe.hip composes the LWE <-> GLWE conversions around the batched key switch while holding the module lock
namespace pz {
int glwe_keyswitch_nolock(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p, size_t batch) {
    return glwe_entry(M, true, false, res, a, key_pmat, p, batch, nullptr);
}
}  // namespace pz

