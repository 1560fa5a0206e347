// api.hip — the C ABI of libpoulpy_hip.so (include/poulpy_hip.h) on top of the
// gfx950 kernels (launch_*.hip; interfaces in internal.hpp): module life cycle, memory, knobs, pinned keys and the device
// mirrors of host-resident prepared keys.  The batched GLWE product lives in api_glwe.hip.
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
// live modules, so that pz_free_bytes can drop the device mirror of a prepared key whose pinned host buffer is being released (the
// Rust shim's `PinnedBuf::drop`): a later allocation at the same address then never meets a stale mirror, fingerprint or not
static std::mutex g_modules_mu;
static std::vector<pz_module*> g_modules;
static std::mutex g_host_allocs_mu;   // its own lock: pz_alloc_bytes / pz_free_bytes never wait behind a mirror sweep's device syncs
static std::unordered_map<void*, size_t> g_host_allocs;   // pz_alloc_bytes blocks (a prepared key may sit inside one)
static std::mutex g_inval_mu;
struct HostInval { const char* lo; const char* hi; uint64_t epoch; };
static constexpr size_t kInvalRing = 1024;
static std::vector<HostInval> g_inval;       // ring, oldest overwritten
static size_t g_inval_next = 0;
static uint64_t g_inval_epoch = 0;           // epoch of the newest published range
static std::atomic<uint64_t> g_inval_epoch_pub{0};   // the same, readable without the lock (PZ_ENTER's one-load check)
static uint64_t g_inval_horizon = 0;         // ranges with epoch <= horizon have left the ring
void host_key_invalidate(const void* p, size_t bytes) {
    if (!p || is_device_ptr(p)) return;
    std::lock_guard<std::mutex> g(g_inval_mu);
    HostInval e{(const char*)p, (const char*)p + std::max<size_t>(bytes, 1), ++g_inval_epoch};
    if (g_inval.size() < kInvalRing) g_inval.push_back(e);
    else { g_inval_horizon = g_inval[g_inval_next].epoch; g_inval[g_inval_next] = e; g_inval_next = (g_inval_next + 1) % kInvalRing; }
    g_inval_epoch_pub.store(g_inval_epoch, std::memory_order_release);
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
// ---- releasing dead mirrors -------------------------------------------------------------------------------
// Publishing a range only marks mirrors stale; their device memory (the mirror + its row-sliced copy: 2 x the key bytes) used to stay
// allocated until the OWNING module next looked a key up.  A sibling parked in the Rust shim's pool, or a thread that moved on to
// device-resident keys, never does - up to 64 keys / 48 GiB per module held for good.  Now: (a) every API call sweeps its own module's
// stale mirrors on entry (one atomic load when nothing was published); (b) whoever publishes a range also sweeps every other live module
// that is idle at that moment (try_lock: never waits for a call in flight); (c) a failed hipMalloc sweeps everything reachable and
// retries once.
// caller holds M->mu
static size_t sweep_own_mirrors(pz_module* M) {
    size_t dropped = 0;
    for (size_t i = M->mirrors.size(); i-- > 0;) {
        uint64_t now = 0;
        if (host_key_stale(M->mirrors[i].host, M->mirrors[i].bytes, M->mirrors[i].epoch, &now)) {
            (void)hipStreamSynchronize(M->stream);
            drop_mirror_at(M, i);
            ++dropped;
        } else M->mirrors[i].epoch = now;
    }
    M->mirror_seen_epoch = g_inval_epoch_pub.load(std::memory_order_acquire);
    return dropped;
}
void mirror_sweep_on_enter(pz_module* M) {
    if (M->mirrors.empty() || M->mirror_seen_epoch == g_inval_epoch_pub.load(std::memory_order_acquire)) return;
    (void)sweep_own_mirrors(M);
}
// every live module other than `self` that no call is running on right now.  Lock order: registry, then module locks by try_lock only
// (a thread that holds a module lock may call this; nobody blocks on a module lock while holding the registry).
size_t sweep_idle_modules(pz_module* self) {
    size_t dropped = 0;
    int dev0 = -1;
    (void)hipGetDevice(&dev0);
    std::lock_guard<std::mutex> g(g_modules_mu);
    for (pz_module* O : g_modules) {
        if (O == self || !O->mu.try_lock()) continue;
        if (!O->mirrors.empty()) {
            if (dev0 < 0) { O->mu.unlock(); continue; }   // cannot restore the caller's device afterwards: leave the sibling alone
            (void)hipSetDevice(O->device);
            dropped += sweep_own_mirrors(O);
        }
        O->mu.unlock();
    }
    if (dev0 >= 0) (void)hipSetDevice(dev0);
    (void)hipGetLastError();
    return dropped;
}
namespace pz {
// hipMalloc that gives dead key mirrors back before it gives up (the caller holds M->mu)
int device_malloc_retry(pz_module* M, void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) return PZ_OK;
    (void)hipGetLastError();
    const size_t freed = sweep_own_mirrors(M) + sweep_idle_modules(M);
    (void)hipSetDevice(M->device);
    if (freed) e = hipMalloc(out, bytes);
    if (e == hipSuccess) return PZ_OK;
    (void)hipGetLastError();
    *out = nullptr;
    return fail(PZ_ERR_HIP, "hipMalloc of %zu bytes failed: %s (stale key mirrors released first: %zu)", bytes, hipGetErrorString(e), freed);
}
}  // namespace pz

int forget_host_key(pz_module* M, const void* host) {
    host_key_invalidate(host, 1);   // every module's mirror of this buffer, not only the caller's
    for (size_t i = 0; i < M->mirrors.size(); ++i)
        if (M->mirrors[i].host == host) {
            PZ_HIP(hipStreamSynchronize(M->stream));
            drop_mirror_at(M, i);
            break;
        }
    (void)sweep_idle_modules(M);   // the siblings that mirror this buffer and are idle let go of it now
    PZ_HIP(hipSetDevice(M->device));
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
int resolve_key(pz_module* M, const double* pmat, size_t bytes, const double** out) {
    if (is_device_ptr(pmat)) { *out = pmat; return PZ_OK; }
    // mirrors whose host range was (re)prepared, zeroed, forgotten or freed since their validation - by any module - go first
    (void)sweep_own_mirrors(M);
    // the epoch a fresh mirror is stamped with is read BEFORE its bytes are fingerprinted and uploaded: a range published
    // while they are being read is then newer than the mirror and kills it at the next lookup
    uint64_t epoch_now = 0;
    (void)host_key_stale(pmat, bytes, ~0ull >> 1, &epoch_now);   // (only reads the current epoch)
    const uint64_t fp = host_fingerprint(pmat, bytes);
    for (size_t i = 0; i < M->mirrors.size(); ++i) {
        auto& mr = M->mirrors[i];
        if (mr.host != (const void*)pmat) continue;
        if (mr.bytes == bytes && mr.fp == fp) { mr.stamp = ++M->mirror_clock; *out = (const double*)mr.dev; return PZ_OK; }
        PZ_HIP(hipStreamSynchronize(M->stream));
        drop_mirror_at(M, i);
        break;
    }
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
    PZ_TRY(device_malloc_retry(M, &dev, bytes));
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

// ------------------------------------------------------------------------------
// public: misc
// ------------------------------------------------------------------------------
extern "C" {

const char* pz_last_error(void) { return last_error_ref().c_str(); }
uint32_t pz_abi_version(void) { return PZ_ABI_VERSION; }   // 4: - pz_module_set_phase_tuning / _phase_tuning_state (the placement tuner measured nothing, NOTEBOOK.md 13)

int pz_module_new_on_device(uint64_t n, int device, pz_module** out) {
    if (!out) return fail(PZ_ERR_INVALID, "null out");
    *out = nullptr;
    FftPlan pl;
    if (n < 2 || (n & (n - 1))) return fail(PZ_ERR_INVALID, "n must be a power of two but is %llu", (unsigned long long)n);
    if (!make_plan(n, pl)) return fail(PZ_ERR_UNSUPPORTED, "n=%llu unsupported (need 8 <= n <= 131072)", (unsigned long long)n);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(PZ_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(PZ_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    PZ_HIP(hipSetDevice(device));
    pz_module* M = new pz_module();
    M->n = n; M->m = n >> 1; M->device = device; M->plan = pl;
    int r = PZ_OK;
    do {
#ifdef PZ_EXPERIMENT
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
#endif
        if (hipStreamCreateWithFlags(&M->stream, hipStreamNonBlocking) != hipSuccess) { r = fail(PZ_ERR_HIP, "stream create failed"); break; }
        if ((r = build_tables(M)) != PZ_OK) break;
        if (hipMalloc(&M->margin, 16) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin alloc failed"); break; }
        if (hipMemset(M->margin, 0, 16) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin memset failed"); break; }
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
        if (hipMalloc(&M->margin, 16) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin alloc failed"); break; }
        if (hipMemset(M->margin, 0, 16) != hipSuccess) { r = fail(PZ_ERR_HIP, "margin memset failed"); break; }
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
    for (auto& ge : M->graphs) {
        if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
    }
    if (M->stream2) { (void)hipStreamSynchronize(M->stream2); (void)hipStreamDestroy(M->stream2); }
    if (M->stream_out) { (void)hipStreamSynchronize(M->stream_out); (void)hipStreamDestroy(M->stream_out); }
    if (M->ev_fork) (void)hipEventDestroy(M->ev_fork);
    if (M->ev_join) (void)hipEventDestroy(M->ev_join);
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
        std::lock_guard<std::mutex> g(g_host_allocs_mu);
        g_host_allocs[p] = len;
    }
    return p;
}
void pz_free_bytes(void* p) {
    if (!p) return;
    size_t len = 1;
    {
        std::lock_guard<std::mutex> g(g_host_allocs_mu);
        auto it = g_host_allocs.find(p);
        if (it != g_host_allocs.end()) { len = it->second; g_host_allocs.erase(it); }
    }
    // a prepared key may sit inside the block: publish the range instead of locking every live module (a buffer drop on one thread
    // used to wait for whatever GPU call was in flight on every other thread's sibling module); the mirrors are dropped by their
    // owners at their next key lookup
    host_key_invalidate(p, len);
    // ... and every module that is idle right now releases its mirrors of the range at once (try_lock: a call in flight on another thread's
    // sibling is never waited for; that module sweeps at its next call, or when an allocation fails)
    if (!is_device_ptr(p)) (void)sweep_idle_modules(nullptr);
    (void)hipHostFree(p);
}
int pz_device_alloc(pz_module* M, size_t len, void** out) {
    if (!M || !out) return fail(PZ_ERR_INVALID, "null argument");
    PZ_HIP(hipSetDevice(M->device));
    if (hipMalloc(out, len ? len : 64) == hipSuccess) return PZ_OK;
    (void)hipGetLastError();
    (void)sweep_idle_modules(nullptr);   // dead key mirrors of every idle module (this one included) go first, then once more
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

int pz_module_forget_host_key(pz_module* M, const double* host_pmat) {
    PZ_ENTER(M);
    return forget_host_key(M, (const void*)host_pmat);
}
size_t pz_module_host_key_mirrors(pz_module* M) {
    if (!M) return 0;
    std::lock_guard<std::mutex> lock_(M->mu);
    // mirrors whose host range has been invalidated since (possibly by another module or by pz_free_bytes) are released now
    (void)hipSetDevice(M->device);
    (void)sweep_own_mirrors(M);
    return M->mirrors.size();
}
}  // extern "C"
