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

using namespace pz;

// ------------------------------------------------------------------------------
// HIP graphs for the launch-bound composite calls.  A blind rotation on the composed path is 5 launches per LWE block
// (hundreds per call), a trace 6-8 per step: at small batches the kernels are shorter than their launch cost.  The first
// call with a given argument set runs normally (it sizes the workspaces and builds tables: nothing that allocates or
// synchronizes may happen under capture); the second one is captured on the module's stream and instantiated; from then on
// the call is one hipGraphLaunch.  The key covers every value a kernel argument is derived from (pointers, shapes, the
// module's workspaces and knobs); entries are evicted least-recently-used.  Capture failures fall back to plain launches.
// ------------------------------------------------------------------------------
struct KeyHash {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void* p, size_t len) {
        const unsigned char* c = (const unsigned char*)p;
        for (size_t i = 0; i < len; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    }
    template <typename T> void add(const T& v) { bytes(&v, sizeof(T)); }
};
static void graph_key_module(const pz_module* M, KeyHash& k) {
    k.add(M->ws); k.add(M->ws2); k.add(M->ws_bytes); k.add(M->ws2_bytes); k.add(M->fuse_mid); k.add(M->fuse_tail); k.add(M->small_path); k.add(M->chunk);
    k.add(M->dbg_stages); k.add(M->probe); k.add(M->graph_epoch); k.add(M->w2n);
}
static void graph_drop(pz_module::GraphEntry& e) {
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (e.graph) (void)hipGraphDestroy(e.graph);
    e.exec = nullptr; e.graph = nullptr;
}
template <typename F>
static int with_graph(pz_module* M, uint64_t key, F&& body) {
    static const int env_on = getenv("POULPY_DBG_GRAPHS") ? atoi(getenv("POULPY_DBG_GRAPHS")) : 1;
    if (!env_on || !M->graphs_on || M->timing) return body();
    pz_module::GraphEntry* e = nullptr;
    for (auto& ge : M->graphs) if (ge.key == key) e = &ge;
    if (e && e->exec) {
        e->stamp = ++M->graph_clock;
        PZ_HIP(hipGraphLaunch(e->exec, M->stream));
        M->graph_launches++;
        return PZ_OK;
    }
    if (!e) {  // first sight: plain run, remember the key
        const int rc = body();
        if (rc != PZ_OK) return rc;
        if (M->graphs.size() >= 16) {
            size_t lru = 0;
            for (size_t i = 1; i < M->graphs.size(); ++i) if (M->graphs[i].stamp < M->graphs[lru].stamp) lru = i;
            graph_drop(M->graphs[lru]);
            M->graphs.erase(M->graphs.begin() + (long)lru);
        }
        M->graphs.push_back({key, nullptr, nullptr, ++M->graph_clock, false});
        return PZ_OK;
    }
    if (e->failed) return body();
    e->stamp = ++M->graph_clock;
    if (hipStreamBeginCapture(M->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        e->failed = true;
        return body();
    }
    const int rc = body();
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(M->stream, &g);
    hipGraphExec_t ex = nullptr;
    if (rc == PZ_OK && ce == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex) {
        // (e may dangle if body() touched M->graphs: it does not — nested calls never go through with_graph)
        e->graph = g; e->exec = ex;
        PZ_HIP(hipGraphLaunch(ex, M->stream));
        M->graph_launches++;
        return PZ_OK;
    }
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    e->failed = true;
    // nothing ran under the failed / invalidated capture (launches issued while capturing only record nodes): run the call plainly,
    // also when body() reported an error that the broken capture itself produced
    return body();
}

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
static int forget_host_key(pz_module* M, const void* host) {
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
static int resolve_key(pz_module* M, const double* pmat, size_t bytes, const double** out) {
    if (is_device_ptr(pmat)) { *out = pmat; return PZ_OK; }
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
    PZ_HIP(hipMalloc(&dev, bytes));
    if (hipMemcpyAsync(dev, pmat, bytes, hipMemcpyHostToDevice, M->stream) != hipSuccess) {
        (void)hipFree(dev);
        return fail(PZ_ERR_HIP, "upload of a host-resident prepared key failed");
    }
    M->mirrors.push_back({(const void*)pmat, bytes, dev, fp, ++M->mirror_clock});
    M->graph_epoch++;
    if ((M->plan.m2 == 256 || M->plan.m2 == 128) && (M->plan.m1 % 16) == 0) {
        const size_t npolys = bytes / ((size_t)M->n * 8);
        cplx* sliced = nullptr;
        if (hipMalloc(&sliced, bytes) == hipSuccess) {
            if (launch_permute_pmat(M, (const double*)dev, sliced, (int)npolys) == PZ_OK) M->pinned.push_back({(const void*)dev, sliced, bytes});
            else (void)hipFree(sliced);
        } else {
            (void)hipGetLastError();   // no room for the sliced copy: the pipeline rebuilds it per call
        }
    }
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
static std::vector<std::pair<void*, size_t>> g_host_allocs;   // pz_alloc_bytes blocks (a prepared key may sit inside one)

// ------------------------------------------------------------------------------
// public: misc
// ------------------------------------------------------------------------------
extern "C" {

const char* pz_last_error(void) { return last_error_ref().c_str(); }
uint32_t pz_abi_version(void) { return 2; }

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
    for (void* p : {(void*)M->tw1, (void*)M->tw1inv, (void*)M->wL1, (void*)M->wL2, (void*)M->tw12, (void*)M->tw12t, (void*)M->w2n, M->ws, M->ws2, (void*)M->margin})
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
    if (!(M->plan.m2 == 256 || M->plan.m2 == 128) || (M->plan.m1 % 16) != 0) {  // no fused pipeline at this N: nothing to cache,
        M->pinned.push_back({(const void*)pmat, nullptr, 0});                      // but the pin is remembered so that unpin succeeds
        return PZ_OK;
    }
    const size_t npolys = rows * cols_in * cols_out * size;
    const size_t bytes = npolys * (size_t)M->n * 8;
    cplx* sliced = nullptr;
    PZ_HIP(hipMalloc(&sliced, bytes));
    const int st = launch_permute_pmat(M, pmat, sliced, (int)npolys);
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
        g_host_allocs.emplace_back(p, len);
    }
    return p;
}
void pz_free_bytes(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_modules_mu);
        size_t len = 1;
        for (size_t i = 0; i < g_host_allocs.size(); ++i)
            if (g_host_allocs[i].first == p) {
                len = g_host_allocs[i].second;
                g_host_allocs[i] = g_host_allocs.back();
                g_host_allocs.pop_back();
                break;
            }
        for (pz_module* M : g_modules) {
            std::lock_guard<std::mutex> lock_(M->mu);
            for (size_t i = M->mirrors.size(); i-- > 0;) {
                const char* h = (const char*)M->mirrors[i].host;
                if (h < (const char*)p || h >= (const char*)p + len) continue;
                (void)hipSetDevice(M->device);
                (void)hipStreamSynchronize(M->stream);
                drop_mirror_at(M, i);
            }
        }
    }
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
// public: VecZnxDft
// ------------------------------------------------------------------------------


int pz_vec_znx_dft_apply(pz_module* M, size_t step, size_t offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                         const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(step > 0, "vec_znx_dft_apply: step must be > 0");
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_apply(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_apply(a)");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));  // inout: untouched limbs / other columns survive
    cplx* T;
    PZ_TRY(need_T(M, std::min(res_size, a_size), &T));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_dft_apply(M, 1, (int)step, (int)offset, dr, (int)res_col, da, (int)a_col, 1, nullptr, T));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_vec_znx_dft_apply_batched(pz_module* M, size_t batch, size_t step, size_t offset, double* res, size_t res_cols,
                                 size_t res_size, size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(step > 0, "vec_znx_dft_apply: step must be > 0");
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_apply(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_apply(a)");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a), "batched entry points take device pointers");
    cplx* T;
    PZ_TRY(need_T(M, batch * std::min(res_size, a_size), &T));
    DV dr{res, (long long)(M->n * res_cols * res_size), (int)res_cols, (int)res_size};
    DV da{(void*)a, (long long)(M->n * a_cols * a_size), (int)a_cols, (int)a_size};
    return dev_dft_apply(M, (int)batch, (int)step, (int)offset, dr, (int)res_col, da, (int)a_col, 1, nullptr, T);
}

size_t pz_vec_znx_idft_apply_tmp_bytes(const pz_module*) { return 0; }  // hal_defaults/vec_znx_dft.rs:68-73

static int idft_common(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                       size_t a_cols, size_t a_size, size_t a_col) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_idft_apply(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_idft_apply(a)");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));
    const int min_size = (int)std::min(res_size, a_size);
    cplx* T;
    PZ_TRY(need_T(M, min_size, &T));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_idft(M, 1, dr, (int)res_col, da, (int)a_col, 1, min_size, T));
    PZ_TRY(launch_ew(M, EW_ZERO, poly_ptr(M, dr, (int)res_col, min_size), 0, limb_stride(M, dr), nullptr, 0, 0, nullptr, 0, 0,
                     (int)res_size - min_size, 1));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_vec_znx_idft_apply(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                          size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return idft_common(M, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_idft_apply_tmpa(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, double* a,
                               size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);  // `a` may be used as scratch by the reference; this backend leaves it intact
    return idft_common(M, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}

static int consume_common(pz_module* M, size_t batch, void* data, size_t cols, size_t size, bool require_dev) {
    if (require_dev) PZ_REQUIRE(is_device_ptr(data), "batched entry points take device pointers");
    Stage sd;
    if (require_dev) { sd.M = M; sd.dev = data; }
    else PZ_TRY(sd.in(data, vbytes(M, cols, size), true, true, M));
    cplx* T;
    PZ_TRY(need_T(M, batch * cols * size, &T));
    DV d{sd.dev, (long long)(M->n * cols * size), (int)cols, (int)size};
    PZ_TRY(dev_idft(M, (int)batch, d, 0, d, 0, (int)cols, (int)size, T));
    const bool host = sd.owned;
    PZ_TRY(sd.finish());
    return finish_call(M, host);
}
int pz_vec_znx_idft_apply_consume(pz_module* M, void* data, size_t cols, size_t size) {
    PZ_ENTER(M);
    return consume_common(M, 1, data, cols, size, false);
}
int pz_vec_znx_idft_apply_consume_batched(pz_module* M, size_t batch, void* data, size_t cols, size_t size) {
    PZ_ENTER(M);
    return consume_common(M, batch, data, cols, size, true);
}

// generic staged three-operand limb-range op helper
struct Tri {
    Stage sr, sa, sb;
    DV dr, da, db;
    bool host = false;
};
static int tri_in(pz_module* M, Tri& t, double* res, size_t rc, size_t rs, const double* a, size_t ac, size_t as_, const double* b,
                  size_t bc, size_t bs_) {
    PZ_TRY(t.sa.in(a, a ? vbytes(M, ac, as_) : 0, true, false, M));
    if (b) PZ_TRY(t.sb.in(b, vbytes(M, bc, bs_), true, false, M));
    // res aliasing a or b (assign forms pass res as operand): reuse the same staging
    if ((const void*)res == (const void*)a) { t.sr.M = M; t.sr.dev = t.sa.dev; t.sa.out = true; }
    else PZ_TRY(t.sr.in(res, vbytes(M, rc, rs), true, true, M));
    t.dr = DV{t.sr.dev, 0, (int)rc, (int)rs};
    t.da = DV{t.sa.dev, 0, (int)ac, (int)as_};
    t.db = DV{t.sb.dev, 0, (int)bc, (int)bs_};
    t.host = t.sr.owned || t.sa.owned || t.sb.owned;
    return PZ_OK;
}
static int tri_out(pz_module* M, Tri& t) {
    PZ_TRY(t.sr.finish());
    PZ_TRY(t.sa.finish());
    PZ_TRY(t.sb.finish());
    return finish_call(M, t.host);
}
static int ew_limbs(pz_module* M, int op, const DV& r, int rcol, int rl0, const DV* a, int acol, int al0, const DV* b, int bcol,
                    int bl0, int nl) {
    return launch_ew(M, op, poly_ptr(M, r, rcol, rl0), 0, limb_stride(M, r), a ? poly_ptr(M, *a, acol, al0) : nullptr, 0,
                     a ? limb_stride(M, *a) : 0, b ? poly_ptr(M, *b, bcol, bl0) : nullptr, 0, b ? limb_stride(M, *b) : 0, nl, 1);
}

// i64 = true: the same limb-range logic on i64 containers (reference/vec_znx/add.rs:6-65, sub.rs:6-58), wrapping arithmetic
static inline int ew_for(int op, bool i64) {
    if (!i64) return op;
    return op == EW_ADD ? EW_ADD_I64 : op == EW_SUB ? EW_SUB_I64 : op == EW_NEG ? EW_NEG_I64 : op;
}
static int add_sub_into(pz_module* M, bool sub, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                        size_t a_cols, size_t a_size, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col,
                        bool i64 = false) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_add/sub(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_add/sub(a)");
    PZ_CHECK_COL(b_col, b_cols, "vec_znx_dft_add/sub(b)");
    PZ_REQUIRE((const void*)res != (const void*)b, "vec_znx_dft_add/sub: res must not alias b (use the *_assign form)");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, a, a_cols, a_size, b, b_cols, b_size));
    const bool a_le_b = a_size <= b_size;
    const int sum = (int)std::min(a_le_b ? a_size : b_size, res_size);
    const int cpy = (int)std::min(a_le_b ? b_size : a_size, res_size);
    PZ_TRY(ew_limbs(M, ew_for(sub ? EW_SUB : EW_ADD, i64), t.dr, (int)res_col, 0, &t.da, (int)a_col, 0, &t.db, (int)b_col, 0, sum));
    if (a_le_b) PZ_TRY(ew_limbs(M, ew_for(sub ? EW_NEG : EW_COPY, i64), t.dr, (int)res_col, sum, &t.db, (int)b_col, sum, nullptr, 0, 0, cpy - sum));
    else PZ_TRY(ew_limbs(M, EW_COPY, t.dr, (int)res_col, sum, &t.da, (int)a_col, sum, nullptr, 0, 0, cpy - sum));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, cpy, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - cpy));
    return tri_out(M, t);
}
int pz_vec_znx_dft_add_into(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                            size_t a_cols, size_t a_size, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, false, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}
int pz_vec_znx_dft_sub(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a, size_t a_cols,
                       size_t a_size, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, true, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, b, b_cols, b_size, b_col);
}

// res (op)= a over limb ranges; shifts express add_scaled_assign
static int assign_op(pz_module* M, int op_res_a /*EW_ADD: res+a, EW_SUB: res-a, -EW_SUB: a-res*/, double* res, size_t res_cols,
                     size_t res_size, size_t res_col, const double* a, size_t a_cols, size_t a_size, size_t a_col, int res_shift,
                     int a_shift, int nl, bool negate_tail, bool i64 = false) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_*_assign(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_*_assign(a)");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, a, a_cols, a_size, nullptr, 0, 0));
    if (op_res_a == EW_ADD || op_res_a == EW_SUB)
        PZ_TRY(ew_limbs(M, ew_for(op_res_a, i64), t.dr, (int)res_col, res_shift, &t.dr, (int)res_col, res_shift, &t.da, (int)a_col, a_shift, nl));
    else
        PZ_TRY(ew_limbs(M, ew_for(EW_SUB, i64), t.dr, (int)res_col, res_shift, &t.da, (int)a_col, a_shift, &t.dr, (int)res_col, res_shift, nl));
    if (negate_tail)
        PZ_TRY(ew_limbs(M, ew_for(EW_NEG, i64), t.dr, (int)res_col, nl, &t.dr, (int)res_col, nl, nullptr, 0, 0, (int)res_size - nl));
    return tri_out(M, t);
}
int pz_vec_znx_dft_add_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                              size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_ADD, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, 0, 0, (int)std::min(a_size, res_size), false);
}
int pz_vec_znx_dft_sub_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                              size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_SUB, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, 0, 0, (int)std::min(a_size, res_size), false);
}
int pz_vec_znx_dft_sub_negate_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                                     size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, -EW_SUB, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, 0, 0, (int)std::min(a_size, res_size), true);
}
// ---- i64 VecZnx limb-wise family (hal_impl.rs:59 add_into, :65 add_assign, :90 sub, :96 sub_assign, :101 sub_negate_assign,
//      :126 negate, :131 negate_assign, :289 copy, :34 zero): SURVEY.md 8f rank 3, so that ciphertexts stay on the device
//      between the hot-path operations.  Same limb-range rules as the DFT-domain family above, wrapping i64 arithmetic.
int pz_vec_znx_add_into(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                        size_t a_size, size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, false, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, (const double*)b, b_cols,
                        b_size, b_col, true);
}
int pz_vec_znx_sub(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                   size_t a_size, size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return add_sub_into(M, true, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, (const double*)b, b_cols,
                        b_size, b_col, true);
}
int pz_vec_znx_add_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                          size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_ADD, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, 0, 0,
                     (int)std::min(a_size, res_size), false, true);
}
int pz_vec_znx_sub_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                          size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, EW_SUB, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, 0, 0,
                     (int)std::min(a_size, res_size), false, true);
}
int pz_vec_znx_sub_negate_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                                 size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return assign_op(M, -EW_SUB, (double*)res, res_cols, res_size, res_col, (const double*)a, a_cols, a_size, a_col, 0, 0,
                     (int)std::min(a_size, res_size), true, true);
}
// res = -a over the common limbs, zero beyond (negate.rs:6-29); copy: res = a, zero beyond (copy.rs)
static int negate_or_copy(pz_module* M, int op, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                          size_t a_cols, size_t a_size, size_t a_col) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_negate/copy(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_negate/copy(a)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    const int mn = (int)std::min(res_size, a_size);
    PZ_TRY(ew_limbs(M, op, t.dr, (int)res_col, 0, &t.da, (int)a_col, 0, nullptr, 0, 0, mn));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, mn, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - mn));
    return tri_out(M, t);
}
int pz_vec_znx_negate(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                      size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return negate_or_copy(M, EW_NEG_I64, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_copy(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a, size_t a_cols,
                    size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(!((const void*)res == (const void*)a && res_col != a_col), "vec_znx_copy: column-to-column copy inside one container is not supported");
    if ((const void*)res == (const void*)a) return PZ_OK;
    return negate_or_copy(M, EW_COPY, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_negate_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_negate_assign(res)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, nullptr, 0, 0, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_NEG_I64, t.dr, (int)res_col, 0, &t.dr, (int)res_col, 0, nullptr, 0, 0, (int)res_size));
    return tri_out(M, t);
}
int pz_vec_znx_zero(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_zero(res)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, nullptr, 0, 0, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, 0, nullptr, 0, 0, nullptr, 0, 0, (int)res_size));
    return tri_out(M, t);
}
int pz_vec_znx_dft_add_scaled_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* a,
                                     size_t a_cols, size_t a_size, size_t a_col, int64_t a_scale) {
    PZ_ENTER(M);
    int rs = 0, as_ = 0, nl;  // vec_znx_dft.rs:93-128
    if (a_scale > 0) {
        size_t shift = std::min<size_t>((size_t)a_scale, a_size);
        size_t mn = std::min(a_size, res_size);
        nl = (int)(mn > shift ? mn - shift : 0);
        as_ = (int)shift;
    } else if (a_scale < 0) {
        size_t shift = std::min<size_t>((size_t)(-a_scale), res_size);
        nl = (int)std::min(a_size, res_size - shift);
        rs = (int)shift;
    } else {
        nl = (int)std::min(a_size, res_size);
    }
    return assign_op(M, EW_ADD, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col, rs, as_, nl, false);
}

int pz_vec_znx_dft_copy(pz_module* M, size_t step, size_t offset, double* res, size_t res_cols, size_t res_size, size_t res_col,
                        const double* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(step > 0, "vec_znx_dft_copy: step must be > 0");
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_copy(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_dft_copy(a)");
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_dft_copy: res must not alias a");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, a, a_cols, a_size, nullptr, 0, 0));
    const int steps = (int)((a_size + step - 1) / step);
    const int min_steps = std::min((int)res_size, steps);
    int nv = 0;
    if (offset < a_size) nv = std::min(min_steps, (int)((a_size - offset + step - 1) / step));
    // strided source limbs: limb stride of the source is step*cols*n
    PZ_TRY(launch_ew(M, EW_COPY, poly_ptr(M, t.dr, (int)res_col, 0), 0, limb_stride(M, t.dr), poly_ptr(M, t.da, (int)a_col, (int)offset),
                     0, (long long)step * limb_stride(M, t.da), nullptr, 0, 0, nv, 1));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, nv, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - nv));
    return tri_out(M, t);
}
int pz_vec_znx_dft_zero(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_dft_zero(res)");
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, nullptr, 0, 0, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, 0, nullptr, 0, 0, nullptr, 0, 0, (int)res_size));
    return tri_out(M, t);
}

// ------------------------------------------------------------------------------
// public: SVP
// ------------------------------------------------------------------------------
int pz_svp_prepare(pz_module* M, double* res, size_t res_cols, size_t res_col, const int64_t* a, size_t a_cols, size_t a_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "svp_prepare(res)");
    PZ_CHECK_COL(a_col, a_cols, "svp_prepare(a)");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, 1), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, 1), true, true, M));
    cplx* T;
    PZ_TRY(need_T(M, 1, &T));
    DV dr{sr.dev, 0, (int)res_cols, 1}, da{sa.dev, 0, (int)a_cols, 1};
    PZ_TRY(dev_dft_apply(M, 1, 1, 0, dr, (int)res_col, da, (int)a_col, 1, nullptr, T));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_svp_apply_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol, size_t a_cols,
                     size_t a_col, const int64_t* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "svp_apply_dft(res)");
    PZ_CHECK_COL(a_col, a_cols, "svp_apply_dft(ppol)");
    PZ_CHECK_COL(b_col, b_cols, "svp_apply_dft(b)");
    Stage sr, sp, sb;
    PZ_TRY(sp.in(ppol, vbytes(M, a_cols, 1), true, false, M));
    PZ_TRY(sb.in(b, vbytes(M, b_cols, b_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));
    const int min_size = (int)std::min(res_size, b_size);
    cplx* T;
    PZ_TRY(need_T(M, min_size, &T));
    // svp.rs:21-54: FFT of limbs < min_size times ppol, the rest zero
    DV dr{sr.dev, 0, (int)res_cols, min_size}, db{sb.dev, 0, (int)b_cols, (int)b_size};
    const cplx* mul = reinterpret_cast<const cplx*>((const double*)sp.dev + (size_t)M->n * a_col);
    PZ_TRY(dev_dft_apply(M, 1, 1, 0, dr, (int)res_col, db, (int)b_col, 1, mul, T));
    DV drf{sr.dev, 0, (int)res_cols, (int)res_size};
    PZ_TRY(ew_limbs(M, EW_ZERO, drf, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));
    const bool host = sr.owned || sp.owned || sb.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sp.finish());
    PZ_TRY(sb.finish());
    return finish_call(M, host);
}

static int svp_dft_to_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol,
                          size_t a_cols, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_CHECK_COL(res_col, res_cols, "svp_apply_dft_to_dft(res)");
    PZ_CHECK_COL(a_col, a_cols, "svp_apply_dft_to_dft(ppol)");
    PZ_CHECK_COL(b_col, b_cols, "svp_apply_dft_to_dft(b)");
    Stage sp;
    PZ_TRY(sp.in(ppol, vbytes(M, a_cols, 1), true, false, M));
    Tri t;
    PZ_TRY(tri_in(M, t, res, res_cols, res_size, b, b_cols, b_size, nullptr, 0, 0));
    const int min_size = (int)std::min(res_size, b_size);
    // res[j] = ppol * b[j]: the prepared polynomial is the same for every limb (limb stride 0)
    PZ_TRY(launch_ew(M, EW_CMUL, poly_ptr(M, t.dr, (int)res_col, 0), 0, limb_stride(M, t.dr),
                     (const double*)sp.dev + (size_t)M->n * a_col, 0, 0, poly_ptr(M, t.da, (int)b_col, 0), 0, limb_stride(M, t.da),
                     min_size, 1));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));
    t.host = t.host || sp.owned;
    PZ_TRY(sp.finish());
    return tri_out(M, t);
}
int pz_svp_apply_dft_to_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol,
                            size_t a_cols, size_t a_col, const double* b, size_t b_cols, size_t b_size, size_t b_col) {
    PZ_ENTER(M);
    return svp_dft_to_dft(M, res, res_cols, res_size, res_col, ppol, a_cols, a_col, b, b_cols, b_size, b_col);
}
int pz_svp_apply_dft_to_dft_assign(pz_module* M, double* res, size_t res_cols, size_t res_size, size_t res_col, const double* ppol,
                                   size_t a_cols, size_t a_col) {
    PZ_ENTER(M);
    return svp_dft_to_dft(M, res, res_cols, res_size, res_col, ppol, a_cols, a_col, res, res_cols, res_size, res_col);
}

// ------------------------------------------------------------------------------
// public: VMP
// ------------------------------------------------------------------------------
size_t pz_vmp_prepare_tmp_bytes(const pz_module* M, size_t, size_t, size_t, size_t) { return M ? (size_t)M->n * 8 : 0; }
size_t pz_vmp_apply_dft_to_dft_tmp_bytes(const pz_module*, size_t, size_t a_size, size_t b_rows, size_t b_cols_in, size_t, size_t) {
    return (16 + 8 * std::min(a_size, b_rows) * b_cols_in) * 8;  // vmp.rs:132-135
}
size_t pz_vmp_apply_dft_tmp_bytes(const pz_module* M, size_t res_size, size_t a_size, size_t b_rows, size_t b_cols_in,
                                  size_t b_cols_out, size_t b_size) {
    // hal_impl/family_common.rs:3-15
    return pz_bytes_of_vec_znx_dft(M ? M->n : 0, b_cols_in, std::min(a_size, b_rows)) +
           pz_vmp_apply_dft_to_dft_tmp_bytes(M, res_size, a_size, b_rows, b_cols_in, b_cols_out, b_size);
}

int pz_vmp_prepare(pz_module* M, double* pmat, const int64_t* mat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    PZ_ENTER(M);
    PZ_TRY(forget_host_key(M, (const void*)pmat));   // (a device mirror of this host buffer would be stale)
    const size_t npolys = rows * cols_in * cols_out * size;
    Stage sp, sm;
    PZ_TRY(sm.in(mat, npolys * M->n * 8, true, false, M));
    PZ_TRY(sp.in(pmat, npolys * M->n * 8, false, true, M));
    // Device VmpPMat = spectra of the MatZnx polynomials in MatZnx order (entry (r, c) at (r*ncols + c)*n):
    // one FFT per matrix entry (vmp.rs:52-93) and no block re-layout.
    const size_t group = 256;
    cplx* T;
    PZ_TRY(need_T(M, std::min(npolys, group), &T));
    const long long n = (long long)M->n;
    for (size_t p0 = 0; p0 < npolys; p0 += group) {
        const int cnt = (int)std::min(group, npolys - p0);
        PolyMap sm_{cnt, 1, 0, n, 0, (long long)p0 * n};
        PolyMap dm_{cnt, 1, 0, n, 0, (long long)p0 * n};
        PZ_TRY(launch_fwd_pass1(M, cnt, (const long long*)sm.dev, sm_, T));
        PZ_TRY(launch_fwd_pass2(M, cnt, T, (double*)sp.dev, dm_, nullptr));
    }
    const bool host = sp.owned || sm.owned;
    PZ_TRY(sp.finish());
    PZ_TRY(sm.finish());
    return finish_call(M, host);
}

int pz_vmp_zero(pz_module* M, double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    PZ_ENTER(M);
    PZ_TRY(forget_host_key(M, (const void*)pmat));
    const size_t bytes = rows * cols_in * cols_out * size * M->n * 8;
    if (is_device_ptr(pmat)) PZ_HIP(hipMemsetAsync(pmat, 0, bytes, M->stream));
    else memset(pmat, 0, bytes);
    return PZ_OK;
}

static int vmp_checks(pz_module* M, size_t res_cols, size_t a_cols, size_t cols_in, size_t cols_out) {
    (void)M;
    PZ_REQUIRE(res_cols == cols_out, "vmp_apply: res.cols %zu != pmat.cols_out %zu", res_cols, cols_out);
    PZ_REQUIRE(a_cols == cols_in, "vmp_apply: a.cols %zu != pmat.cols_in %zu", a_cols, cols_in);
    return PZ_OK;
}

int pz_vmp_apply_dft_to_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, const double* a, size_t a_cols, size_t a_size,
                            const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size, size_t limb_offset) {
    PZ_ENTER(M);
    PZ_TRY(vmp_checks(M, res_cols, a_cols, cols_in, cols_out));
    PZ_REQUIRE((const void*)res != (const void*)a, "vmp_apply_dft_to_dft: res must not alias a");
    Stage sr, sa, sp;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sp.in(pmat, rows * cols_in * cols_out * size * M->n * 8, true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), false, true, M));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_vmp(M, 1, dr, da, (const double*)sp.dev, (int)rows, (int)cols_in, (int)cols_out, (int)size, (int)limb_offset));
    const bool host = sr.owned || sa.owned || sp.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    PZ_TRY(sp.finish());
    return finish_call(M, host);
}

int pz_vmp_apply_dft_to_dft_batched(pz_module* M, size_t batch, double* res, size_t res_cols, size_t res_size, const double* a,
                                    size_t a_cols, size_t a_size, const double* pmat, size_t rows, size_t cols_in, size_t cols_out,
                                    size_t size, size_t limb_offset) {
    PZ_ENTER(M);
    PZ_TRY(vmp_checks(M, res_cols, a_cols, cols_in, cols_out));
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a) && is_device_ptr(pmat), "batched entry points take device pointers");
    DV dr{res, (long long)(M->n * res_cols * res_size), (int)res_cols, (int)res_size};
    DV da{(void*)a, (long long)(M->n * a_cols * a_size), (int)a_cols, (int)a_size};
    return dev_vmp(M, (int)batch, dr, da, pmat, (int)rows, (int)cols_in, (int)cols_out, (int)size, (int)limb_offset);
}

int pz_vmp_apply_dft(pz_module* M, double* res, size_t res_cols, size_t res_size, const int64_t* a, size_t a_cols, size_t a_size,
                     const double* pmat, size_t rows, size_t cols_in, size_t cols_out, size_t size) {
    PZ_ENTER(M);
    PZ_REQUIRE(res_cols == cols_out, "vmp_apply_dft: res.cols %zu != pmat.cols_out %zu", res_cols, cols_out);
    PZ_REQUIRE(a_cols <= cols_in, "vmp_apply_dft: a.cols %zu > pmat.cols_in %zu", a_cols, cols_in);
    Stage sr, sa, sp;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sp.in(pmat, rows * cols_in * cols_out * size * M->n * 8, true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), false, true, M));
    // family_common.rs:17-54: DFT of a right-aligned into cols_in columns (leading columns zero), then the product
    const size_t sz = std::min(a_size, rows);
    const size_t adft_bytes = vbytes(M, cols_in, sz);
    PZ_TRY(ws_reserve(M, adft_bytes + sz * a_cols * M->m * sizeof(cplx)));
    double* adft = (double*)M->ws;
    cplx* T = (cplx*)((char*)M->ws + adft_bytes);
    PZ_HIP(hipMemsetAsync(adft, 0, adft_bytes, M->stream));
    DV dad{adft, 0, (int)cols_in, (int)sz}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_dft_apply(M, 1, 1, 0, dad, (int)(cols_in - a_cols), da, 0, (int)a_cols, nullptr, T));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size};
    PZ_TRY(dev_vmp(M, 1, dr, dad, (const double*)sp.dev, (int)rows, (int)cols_in, (int)cols_out, (int)size, 0));
    const bool host = sr.owned || sa.owned || sp.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    PZ_TRY(sp.finish());
    return finish_call(M, host);
}

// ------------------------------------------------------------------------------
// public: VecZnxBig
// ------------------------------------------------------------------------------
size_t pz_vec_znx_big_normalize_tmp_bytes(const pz_module* M) { return M ? 3 * (size_t)M->n * 8 : 0; }  // normalize.rs:13-15

static int normalize_checks(size_t res_col, size_t res_cols, size_t a_col, size_t a_cols, size_t res_base2k, size_t a_base2k) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_big_normalize(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_big_normalize(a)");
    PZ_REQUIRE(res_base2k >= 1 && res_base2k <= 63 && a_base2k >= 1 && a_base2k <= 63, "vec_znx_big_normalize: base2k out of range");
    return PZ_OK;
}

static int normalize_impl(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                          size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    PZ_TRY(normalize_checks(res_col, res_cols, a_col, a_cols, res_base2k, a_base2k));
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_big_normalize: res must not alias a");
    Stage sr, sa;
    PZ_TRY(sa.in(a, vbytes(M, a_cols, a_size), true, false, M));
    PZ_TRY(sr.in(res, vbytes(M, res_cols, res_size), true, true, M));
    DV dr{sr.dev, 0, (int)res_cols, (int)res_size}, da{sa.dev, 0, (int)a_cols, (int)a_size};
    PZ_TRY(dev_normalize(M, 1, dr, (int)res_base2k, res_offset, (int)res_col, da, (int)a_base2k, (int)a_col));
    const bool host = sr.owned || sa.owned;
    PZ_TRY(sr.finish());
    PZ_TRY(sa.finish());
    return finish_call(M, host);
}

int pz_vec_znx_big_normalize(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                             size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    PZ_ENTER(M);
    return normalize_impl(M, res, res_cols, res_size, res_base2k, res_offset, res_col, a, a_cols, a_size, a_base2k, a_col);
}
// vec_znx_normalize (hal_impl.rs:41): with ScalarBig = i64 (poulpy-cpu-ref/src/fft64/module.rs:40-43) it is the function
// vec_znx_big_normalize forwards to (reference/fft64/vec_znx_big.rs:241-278 -> vec_znx/normalize.rs:18-48)
size_t pz_vec_znx_normalize_tmp_bytes(const pz_module* M) { return pz_vec_znx_big_normalize_tmp_bytes(M); }
int pz_vec_znx_normalize(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k, int64_t res_offset,
                         size_t res_col, const int64_t* a, size_t a_cols, size_t a_size, size_t a_base2k, size_t a_col) {
    PZ_ENTER(M);
    return normalize_impl(M, res, res_cols, res_size, res_base2k, res_offset, res_col, a, a_cols, a_size, a_base2k, a_col);
}
// vec_znx_normalize_assign (hal_impl.rs:55; reference/vec_znx/normalize.rs:403-425): in place, same base == the out-of-place
// same-base normalization of a copy of the column
int pz_vec_znx_normalize_assign(pz_module* M, size_t base2k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_normalize_assign(res)");
    PZ_REQUIRE(base2k >= 1 && base2k <= 63, "vec_znx_normalize_assign: base2k out of range");
    if (size == 0) return PZ_OK;
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
    DV dr{sr.dev, 0, (int)cols, (int)size};
    PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
    DV tv{M->ws, 0, 1, (int)size};
    PZ_TRY(dev_normalize(M, 1, dr, (int)base2k, 0, (int)col, tv, (int)base2k, 0));
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

// vec_znx_lsh (hal_impl.rs:165), vec_znx_rsh (:137), vec_znx_lsh_assign (:221): reference/vec_znx/shift.rs:68-135, :245-342,
// :16-66 walk the limbs exactly as vec_znx_normalize does at equal bases with res_offset = +k / -k (same step functions, same
// ranges; pinned on the literal restatement by tests/test_oracle_exact.py P10), so they run on the normalize kernels.
size_t pz_vec_znx_lsh_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }  // shift.rs:12-14
int pz_vec_znx_lsh(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                   size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(k <= ((size_t)1 << 40), "vec_znx_lsh: shift out of range");
    return normalize_impl(M, res, res_cols, res_size, base2k, (int64_t)k, res_col, a, a_cols, a_size, base2k, a_col);
}
int pz_vec_znx_rsh(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                   size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_REQUIRE(k <= ((size_t)1 << 40), "vec_znx_rsh: shift out of range");
    return normalize_impl(M, res, res_cols, res_size, base2k, -(int64_t)k, res_col, a, a_cols, a_size, base2k, a_col);
}
int pz_vec_znx_lsh_assign(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_lsh_assign(res)");
    PZ_REQUIRE(base2k >= 1 && base2k <= 63, "vec_znx_lsh_assign: base2k out of range");
    PZ_REQUIRE(k <= ((size_t)1 << 40), "vec_znx_lsh_assign: shift out of range");
    if (size == 0) return PZ_OK;
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
    DV dr{sr.dev, 0, (int)cols, (int)size};
    PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
    DV tv{M->ws, 0, 1, (int)size};
    PZ_TRY(dev_normalize(M, 1, dr, (int)base2k, (long long)k, (int)col, tv, (int)base2k, 0));
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

int pz_vec_znx_big_normalize_batched(pz_module* M, size_t batch, int64_t* res, size_t res_cols, size_t res_size, size_t res_base2k,
                                     int64_t res_offset, size_t res_col, const int64_t* a, size_t a_cols, size_t a_size,
                                     size_t a_base2k, size_t a_col) {
    PZ_ENTER(M);
    PZ_TRY(normalize_checks(res_col, res_cols, a_col, a_cols, res_base2k, a_base2k));
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(a), "batched entry points take device pointers");
    DV dr{res, (long long)(M->n * res_cols * res_size), (int)res_cols, (int)res_size};
    DV da{(void*)a, (long long)(M->n * a_cols * a_size), (int)a_cols, (int)a_size};
    return dev_normalize(M, (int)batch, dr, (int)res_base2k, res_offset, (int)res_col, da, (int)a_base2k, (int)a_col);
}

int pz_vec_znx_big_add_small_assign(pz_module* M, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                                    size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_big_add_small_assign(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_big_add_small_assign(a)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    PZ_TRY(ew_limbs(M, EW_ADD_I64, t.dr, (int)res_col, 0, &t.dr, (int)res_col, 0, &t.da, (int)a_col, 0, (int)std::min(a_size, res_size)));
    return tri_out(M, t);
}

// vec_znx_automorphism (hal_impl.rs:236) and vec_znx_big_automorphism (:517) are the same operation on i64 containers
// (fft64/vec_znx_big.rs:144-170 re-types the big container and calls the VecZnx function)
static int automorphism_into(pz_module* M, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                             const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_automorphism(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_automorphism(a)");
    PZ_REQUIRE((p & 1) != 0, "vec_znx_automorphism: the Galois element must be odd");
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_automorphism: res must not alias a (use the *_assign form)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    const int min_size = (int)std::min(res_size, a_size);
    const long long n = (long long)M->n;
    PolyMap sm{std::max(min_size, 1), 1, 0, (long long)a_cols * n, 0, n * (long long)a_col};
    PolyMap dm{std::max(min_size, 1), 1, 0, (long long)res_cols * n, 0, n * (long long)res_col};
    PZ_TRY(launch_automorphism(M, min_size, (const long long*)t.da.p, sm, (long long*)t.dr.p, dm, inv_mod_2n(p, n), 1));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));  // automorphism.rs:32-34
    return tri_out(M, t);
}
static int automorphism_assign(pz_module* M, int64_t p, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_CHECK_COL(col, cols, "vec_znx_automorphism_assign(res)");
    PZ_REQUIRE((p & 1) != 0, "vec_znx_automorphism_assign: the Galois element must be odd");
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    if (size > 0) {
        // the reference permutes through one polynomial of scratch (automorphism.rs:37-51); here: the column's limbs are
        // copied to the workspace and gathered back
        PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
        DV dr{sr.dev, 0, (int)cols, (int)size};
        PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
        PolyMap sm{(int)size, 1, 0, n, 0, 0};
        PolyMap dm{(int)size, 1, 0, (long long)cols * n, 0, n * (long long)col};
        PZ_TRY(launch_automorphism(M, (int)size, (const long long*)M->ws, sm, (long long*)sr.dev, dm, inv_mod_2n(p, n), 1));
    }
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}
size_t pz_vec_znx_automorphism_assign_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }      // automorphism.rs:6-8
size_t pz_vec_znx_big_automorphism_assign_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }  // vec_znx_big.rs:140-142
int pz_vec_znx_automorphism(pz_module* M, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                            size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return automorphism_into(M, p, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_automorphism_assign(pz_module* M, int64_t p, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    return automorphism_assign(M, p, res, cols, size, col);
}
int pz_vec_znx_big_automorphism(pz_module* M, int64_t p, int64_t* res, size_t res_cols, size_t res_size, size_t res_col,
                                const int64_t* a, size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    return automorphism_into(M, p, res, res_cols, res_size, res_col, a, a_cols, a_size, a_col);
}
int pz_vec_znx_big_automorphism_assign(pz_module* M, int64_t p, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    return automorphism_assign(M, p, res, cols, size, col);
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
    w.total = w.key + w.conv + w.t + w.t2 + w.rtmp + w.small2 + kMidDummyBytes;
    return w;
}
// keyswitch: 0 external product, 1 key switch, 2 automorphism family, 3 tensor relinearization.  The figure is what the call reserves
// in the module's grow-only workspace (+ the 12.5 % growth slack of its first allocation); a key that is neither pinned nor mirrored
// costs its row-sliced copy, which is included.
size_t pz_glwe_op_workspace_bytes(const pz_module* M, const pz_glwe_op_params* p, size_t batch, int keyswitch) {
    if (!M || !p || p->key_size == 0 || p->a_size == 0) return 0;
    const bool tensor = keyswitch == 3, ks = keyswitch != 0, au = keyswitch == 2;
    const OpShape s = op_shape(p, ks, tensor);
    const size_t chunk = pick_chunk(M, p, s, batch);
    const size_t bytes = fused_applies(M, p, s, ks, tensor, au) ? fused_ws(M, p, s, chunk, au).total : op_ws(M, p, s, chunk, ks, au).total;
    return bytes + (bytes >> 3);
}

// Automorphism family on top of the key switch (poulpy-core automorphism/glwe_ct.rs:51-275).  With phi = X -> X^p:
//   mode 0  res = phi(normalize(big))                       (:65-71)
//   mode 1  res = normalize(phi(big) + a)   (add, :133-138)   2: phi(big) - a (:222-227)   3: a - phi(big) (:268-273)
// where big is the key-switch value including the body (keyswitching/glwe.rs:236-237).  Normalization acts per
// coefficient, so modes 1-3 are computed as  phi(normalize'(s .* (big + small)))  with small = -+phi^-1(a) (+ body) built
// by one gather kernel, s(n) the sign phi gives coefficient n (applied inside the tail before the carry chain; flipped
// for mode 3) and a final sign-free permutation; mode 0 is the plain key switch followed by the signed permutation.
struct AutoSpec {
    long long p;
    int mode;
};
// ciphertexts that are not tightly packed (the entries of one column of a GGSW) and a body that lands in another column
struct OpLayout {
    long long a_stride, res_stride;  // in i64 elements between consecutive ciphertexts
    int body_col;
};
// (Round 2 experiment, removed — git history has it: a CU-partitioned, overlapped form of the fused pipeline.  With a CU mask spread
//  over the 8 XCDs (hipExtStreamCreateWithCUMask; POULPY_DBG_CU_MASK still runs the whole pipeline under one) pass 1 and the tail
//  keep their full rate down to 64 CUs while the middle kernel scales with its CU count (profiles/r02_cu_mask_scaling.txt), so chunk
//  c+1's pass 1, chunk c's middle kernel and chunk c-1's tail were run concurrently on disjoint CU sets, chained by events.
//  Bit-exact, but slower in every split tried (best 73 500/s with 8 + 8 CUs per XCD for the two streams against 88 700/s back to
//  back, profiles/r02_overlap_sweep.txt): under concurrency the three kernels share HBM at ~4.7 TB/s aggregate — no better than
//  the 4.7 TB/s the back-to-back sequence averages — and the three-deep chunk pipeline adds its fill / drain per call.)

static int glwe_op(pz_module* M, bool ks, int64_t* res, const int64_t* a, const double* pmat, const pz_glwe_op_params* p, size_t batch,
                   const AutoSpec* au = nullptr, const OpLayout* lay = nullptr, bool tensor = false) {
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
        PZ_TRY(ws_reserve(M, key_bytes + conv_bytes + t_bytes + t2_bytes + rtmp_bytes + small2_bytes + kMidDummyBytes));
        char* base = (char*)M->ws;
        cplx* Pp = (cplx*)base; base += key_bytes;
        int64_t* a_conv = (int64_t*)base; base += conv_bytes;
        cplx* T = (cplx*)base; base += t_bytes;
        cplx* T2 = (cplx*)base; base += t2_bytes;
        int64_t* res_tmp = (int64_t*)base; base += rtmp_bytes;
        int64_t* small2 = (int64_t*)base; base += small2_bytes;
        cplx* mid_dummy = (cplx*)base;
        // the key arrives in the standard device layout; its row-sliced copy is rebuilt per call (2 x 128 MiB of
        // traffic at the metric shape, ~4 % of a 128-ciphertext call) so that no stale copy can ever be used
        bool pinned = false;
        for (auto& pk : M->pinned)
            if (pk.key == (const void*)pmat && pk.sliced && pk.bytes == (size_t)nrows * ncols * (size_t)M->n * 8) { Pp = pk.sliced; pinned = true; }
        if (!pinned && (M->dbg_stages & 2)) PZ_TRY(launch_permute_pmat(M, pmat, Pp, nrows * ncols));
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
            if (small_env && M->small_path && !au && !tensor && !digits && !cross_out && !M->probe && M->dbg_stages == 7 &&
                small_supported(M, npi, ksz)) {
                PZ_TRY(launch_small_fwd(M, nb * npi, (const long long*)av.p, sm, T));
                PZ_TRY(launch_small_inv(M, nb, T, Pp, npi, nrows, ncols, s.cols_out, ksz, (long long*)(res + (long long)b0 * res_bs), res_bs,
                                        s.cols_out, (int)p->res_size, ks ? (const long long*)av.p : nullptr, av.bs, s.cols_a, a_size,
                                        (int)p->res_base2k, body_col));
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
            const bool spec = au_spec && au_big && (au_p & 3u) == 1u && M->plan.m2 == 128 && M->dbg_stages == 7;
            unsigned perm_mul = 0, perm_add = 0;
            if (spec) {
                const unsigned mm = (unsigned)M->m;
                perm_mul = au_g & (mm - 1u);
                const unsigned long long c0 = (unsigned long long)(((au_p - 1u) >> 2) & (mm - 1u));
                perm_add = (unsigned)((mm - (unsigned)(((unsigned long long)perm_mul * c0) & (unsigned long long)(mm - 1u))) & (mm - 1u));
            }
            if (digits && dg.n == 0) {   // nothing reaches the product (e.g. dsize > a.size): the big value is the body alone
                PZ_HIP(hipMemsetAsync(T2, 0, (size_t)nb * npo * M->m * sizeof(cplx), M->stream));
            } else if (M->dbg_stages & 2)
                PZ_TRY(launch_mid(M, nb, T, T2, Pp, npi, npo, nrows, ncols, mid_dummy, perm_mul, perm_add, digits ? &dg : nullptr));
            int64_t* res_b = res + (long long)b0 * res_bs;
            if (spec) {
                // (the tail reads operand limbs j < min(key_size, a_size) only: the pre-pass covers exactly those)
                const int bl = std::min(a_size, ksz);
                PolyMap bsm{bl, 1, av.bs, (long long)av.cols * n, 0, 0}, bdm{bl, 1, (long long)bl * n, n, 0, 0};
                // operand of the body column, one stream: phi(body) + a0 (add) or -phi(body) + a0 (sub forms: the tail negates every operand)
                PZ_TRY(launch_automorphism(M, nb * bl, (const long long*)av.p, bsm, (long long*)res_tmp, bdm, au_g, au->mode == 1 ? 1 : 3,
                                           (const long long*)av.p, bsm));
                PZ_TRY(launch_inv_tail(M, nb, T2, ksz, s.cols_out, (long long*)res_b, res_bs, s.cols_out, (int)p->res_size,
                                       (const long long*)av.p, av.bs, s.cols_a, a_size, (int)p->res_base2k, true, true,
                                       au->mode == 3 ? 2u * (unsigned)n : 0u, au->mode == 3, 0u, false, body_col, (const long long*)res_tmp,
                                       (long long)bl * n, n, au->mode != 1));
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
        return PZ_OK;
    }

    const OpWs w = op_ws(M, p, s, chunk, ks, au != nullptr);
    PZ_TRY(ws_reserve(M, w.total));
    char* base = (char*)M->ws;
    int64_t* a_conv = (int64_t*)base; base += w.a_conv;
    double* a_dft = (double*)base; base += w.a_dft;
    double* res_dft = (double*)base; base += w.res_dft;
    double* tmp_dft = (double*)base; base += w.tmp_dft;
    cplx* T = (cplx*)base; base += w.T;
    int64_t* res_tmp = (int64_t*)base;

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
static int ggsw_expand_row(pz_module* M, int64_t* ggsw, size_t dnum, const double* const* tsk_pmat, const pz_glwe_op_params* p, size_t count) {
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

// vec_znx_rotate (hal_impl.rs:225) / vec_znx_rotate_assign (:232): res = X^k * a (reference/znx/rotate.rs:3-27), limbs of res
// beyond a.size zeroed (vec_znx/rotate.rs:33-35)
size_t pz_vec_znx_rotate_assign_tmp_bytes(const pz_module* M) { return M ? (size_t)M->n * 8 : 0; }
int pz_vec_znx_rotate(pz_module* M, int64_t k, int64_t* res, size_t res_cols, size_t res_size, size_t res_col, const int64_t* a,
                      size_t a_cols, size_t a_size, size_t a_col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(res_col, res_cols, "vec_znx_rotate(res)");
    PZ_CHECK_COL(a_col, a_cols, "vec_znx_rotate(a)");
    PZ_REQUIRE((const void*)res != (const void*)a, "vec_znx_rotate: res must not alias a (use the *_assign form)");
    Tri t;
    PZ_TRY(tri_in(M, t, (double*)res, res_cols, res_size, (const double*)a, a_cols, a_size, nullptr, 0, 0));
    const int min_size = (int)std::min(res_size, a_size);
    const long long n = (long long)M->n;
    PolyMap sm{std::max(min_size, 1), 1, 0, (long long)a_cols * n, 0, n * (long long)a_col};
    PolyMap dm{std::max(min_size, 1), 1, 0, (long long)res_cols * n, 0, n * (long long)res_col};
    PZ_TRY(launch_rotate(M, min_size, (const long long*)t.da.p, sm, (long long*)t.dr.p, dm, 0, std::max(min_size, 1), nullptr, 0, 0, (long long)k));
    PZ_TRY(ew_limbs(M, EW_ZERO, t.dr, (int)res_col, min_size, nullptr, 0, 0, nullptr, 0, 0, (int)res_size - min_size));
    return tri_out(M, t);
}
int pz_vec_znx_rotate_assign(pz_module* M, int64_t k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_rotate_assign(res)");
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    const long long n = (long long)M->n;
    if (size > 0) {
        PZ_TRY(ws_reserve(M, (size_t)size * (size_t)n * 8));
        DV dr{sr.dev, 0, (int)cols, (int)size};
        PZ_TRY(launch_ew(M, EW_COPY, M->ws, 0, n, poly_ptr(M, dr, (int)col, 0), 0, limb_stride(M, dr), nullptr, 0, 0, (int)size, 1));
        PolyMap sm{(int)size, 1, 0, n, 0, 0};
        PolyMap dm{(int)size, 1, 0, (long long)cols * n, 0, n * (long long)col};
        PZ_TRY(launch_rotate(M, (int)size, (const long long*)M->ws, sm, (long long*)sr.dev, dm, 0, (int)size, nullptr, 0, 0, (long long)k));
    }
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

// vec_znx_rsh_assign (hal_impl.rs:217; reference/vec_znx/shift.rs:186-243)
size_t pz_vec_znx_rsh_tmp_bytes(const pz_module* M) { return M ? 2 * (size_t)M->n * 8 : 0; }  // shift.rs: carry + one polynomial
int pz_vec_znx_rsh_assign(pz_module* M, size_t base2k, size_t k, int64_t* res, size_t cols, size_t size, size_t col) {
    PZ_ENTER(M);
    PZ_CHECK_COL(col, cols, "vec_znx_rsh_assign(res)");
    PZ_REQUIRE(base2k >= 1 && base2k <= 63, "vec_znx_rsh_assign: base2k out of range");
    PZ_REQUIRE(k <= base2k * size, "vec_znx_rsh_assign: shift beyond the precision of res");
    Stage sr;
    PZ_TRY(sr.in(res, vbytes(M, cols, size), true, true, M));
    PZ_TRY(launch_rsh(M, 1, (long long*)sr.dev, 0, (int)cols, (int)size, (int)col, 1, (int)base2k, (int)k));
    const bool host = sr.owned;
    PZ_TRY(sr.finish());
    return finish_call(M, host);
}

// glwe_trace_assign (poulpy-core/src/glwe_trace.rs:129-176) on `batch` ciphertexts, equal base2k for res and keys:
//   for every step s:  res = rsh(res, 1 bit) on every column (operations/glwe.rs:1096-1112);  res = glwe_automorphism_add_assign(res, key_s)
static int glwe_trace(pz_module* M, int64_t* res, size_t nsteps, const int64_t* gals, const double* const* key_pmats,
                      const pz_glwe_op_params* p, size_t batch) {
    PZ_REQUIRE(p != nullptr && (nsteps == 0 || (gals != nullptr && key_pmats != nullptr)), "glwe_trace: null argument");
    PZ_REQUIRE(p->a_size == p->res_size && p->a_base2k == p->res_base2k && p->res_base2k == p->key_base2k && p->rank_out == p->rank,
               "glwe_trace: res and keys must share base2k, and a/res one layout (the other cases re-normalize around this call)");
    PZ_REQUIRE(is_device_ptr(res), "batched entry points take device pointers");
    if (batch == 0) return PZ_OK;
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1;
    const long long ct = n * cols * (long long)p->res_size;
    for (size_t s = 0; s < nsteps; ++s) {
        PZ_REQUIRE((gals[s] & 1) != 0, "glwe_trace: Galois elements must be odd");
        PZ_TRY(launch_rsh(M, (int)batch, (long long*)res, ct, cols, (int)p->res_size, 0, cols, (int)p->res_base2k, 1));
        AutoSpec au{(long long)gals[s], 1};
        PZ_TRY(glwe_op(M, true, res, res, key_pmats[s], p, batch, &au));
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

// ------------------------------------------------------------------------------
// public: CGGI blind rotation on a batch of LWE ciphertexts (device-resident)
// poulpy-bin-fhe/src/blind_rotation/algorithms/cggi/algorithm.rs:76-118 (dispatch), :265-368 (block binary), :370-440 (standard)
// ------------------------------------------------------------------------------
size_t pz_blind_rotation_workspace_bytes(const pz_module* M, const pz_blind_rotation_params* p, size_t batch) {
    if (!M || !p) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->rank + 1;
    const size_t tp = cols * std::max({(size_t)p->dnum, (size_t)p->brk_size, (size_t)p->res_size});
    const size_t T = align256(batch * tp * (size_t)M->m * sizeof(cplx));
    if (p->block_size > 1) {
        // the composed path, or (plans with 128-point rows) the row-sliced keys of one block + T' + T2' of the three-kernel block step
        const size_t composed = align256(batch * n8 * cols * p->dnum) + 2 * align256(batch * n8 * cols * p->brk_size) + T;
        const size_t mid = align256((size_t)p->block_size * p->dnum * cols * cols * p->brk_size * n8) +
                           align256(batch * n8 * cols * std::min((size_t)p->dnum, (size_t)p->res_size)) + align256(batch * n8 * cols * p->brk_size) +
                           kMidDummyBytes;
        return std::max(composed, mid);
    }
    pz_glwe_op_params ep;
    ep.rank = p->rank; ep.dnum = p->dnum; ep.dsize = 1; ep.key_size = p->brk_size; ep.key_base2k = p->base2k;
    ep.a_size = p->res_size; ep.a_base2k = p->base2k; ep.res_size = p->res_size; ep.res_base2k = p->base2k; ep.rank_out = p->rank;
    return align256(batch * n8 * cols * p->res_size) + pz_glwe_op_workspace_bytes(M, &ep, batch, 0);
}

static int blind_rotation(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                          const pz_blind_rotation_params* p, size_t batch) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->n_lwe >= 1 && p->block_size >= 1 && p->dnum >= 1 && p->brk_size >= 1 && p->res_size >= 1 && p->lut_size >= 1,
               "blind_rotation: empty shape");
    PZ_REQUIRE(p->base2k >= 1 && p->base2k <= 63, "blind_rotation: base2k out of range");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(lwe_2n) && is_device_ptr(lut) && is_device_ptr(brk),
               "batched entry points take device pointers");
    if (batch == 0) return PZ_OK;
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1, dnum = (int)p->dnum, bsz = (int)p->brk_size, rsz = (int)p->res_size;
    const int B = (int)batch, n_lwe = (int)p->n_lwe, blk = (int)p->block_size, k = (int)p->base2k;
    const long long lwe_bs = (long long)n_lwe + 1;
    const size_t pmat_doubles = (size_t)n * dnum * cols * cols * bsz;
    const long long res_ct = n * cols * rsz;
    DV rv{res, res_ct, cols, rsz};

    // acc = X^b * LUT in column 0, zero elsewhere (:298-301 / :413-416)
    PZ_HIP(hipMemsetAsync(res, 0, (size_t)B * res_ct * 8, M->stream));
    {
        const int nl = std::min(rsz, (int)p->lut_size);
        PolyMap sm{nl, 1, 0, n, 0, 0};                     // the LUT is shared: batch stride 0, VecZnx(1, lut_size)
        PolyMap dm{nl, 1, res_ct, (long long)cols * n, 0, 0};
        PZ_TRY(launch_rotate(M, B * nl, (const long long*)lut, sm, (long long*)res, dm, 0, nl, (const long long*)lwe_2n, lwe_bs, 0, 0));
    }

    PZ_TRY(ensure_w2n(M));
    {
        bool launched = false;
        PZ_TRY(br_try_fused(M, res, lwe_2n, lut, brk, p, batch, &launched));
        if (launched) return PZ_OK;
    }
    if (blk > 1) {
        const size_t n8 = (size_t)M->n * 8;
        // plans with 128-point rows (N >= 4096): the block step on the three-kernel pipeline of the GLWE products — pass 1 of the
        // accumulator limbs | k_mid128<.., BR> (row DFT, the block's blk products weighted by DFT(X^a_i - 1), inverse row DFT) | tail
        // (inverse column pass + accumulator + carry chain): the spectra never reach HBM and one launch covers the whole block
        {
            const int npi = cols * std::min(dnum, rsz), npo = cols * bsz, nrows_key = dnum * cols, ncols_key = cols * bsz;
            static const int br_mid = getenv("POULPY_DBG_BR_MID") ? atoi(getenv("POULPY_DBG_BR_MID")) : 1;
            if (br_mid && M->fuse_mid && M->fuse_tail && tail_supported(M) && M->plan.m2 == 128 && mid_supported(M, npi, npo) &&
                npi == nrows_key && blk <= 16) {
                const size_t key_bytes = align256((size_t)blk * nrows_key * ncols_key * n8);
                const size_t t_bytes = align256(batch * npi * (size_t)M->m * sizeof(cplx)), t2_bytes = align256(batch * npo * (size_t)M->m * sizeof(cplx));
                PZ_TRY(ws_reserve(M, key_bytes + t_bytes + t2_bytes + kMidDummyBytes));
                char* base = (char*)M->ws;
                cplx* Pp = (cplx*)base; base += key_bytes;
                cplx* T = (cplx*)base; base += t_bytes;
                cplx* T2 = (cplx*)base; base += t2_bytes;
                cplx* mid_dummy = (cplx*)base;
                PolyMap sm{npi / cols, cols, res_ct, (long long)cols * n, n, 0};
                for (int b0 = 0; b0 + blk <= n_lwe; b0 += blk) {
                    PZ_TRY(launch_permute_pmat(M, brk + (size_t)b0 * pmat_doubles, Pp, blk * nrows_key * ncols_key));
                    PZ_TRY(launch_fwd_pass1(M, B * npi, (const long long*)res, sm, T, true));
                    MidBr mb{(const long long*)lwe_2n, lwe_bs, b0, blk};
                    PZ_TRY(launch_mid(M, B, T, T2, Pp, npi, npo, nrows_key, ncols_key, mid_dummy, 0, 0, nullptr, &mb));
                    PZ_TRY(launch_inv_tail(M, B, T2, bsz, cols, (long long*)res, res_ct, cols, rsz, (const long long*)res, res_ct, cols, rsz, k,
                                           true, true));
                }
                return PZ_OK;
            }
        }
        const size_t acc_dft_bytes = align256(batch * n8 * cols * dnum), vr_bytes = align256(batch * n8 * cols * bsz);
        const size_t tp = (size_t)cols * std::max({dnum, bsz, rsz});
        const size_t t_bytes = align256(batch * tp * (size_t)M->m * sizeof(cplx));
        PZ_TRY(ws_reserve(M, acc_dft_bytes + 2 * vr_bytes + t_bytes));
        char* base = (char*)M->ws;
        double* acc_dft = (double*)base; base += acc_dft_bytes;
        double* vmp_res = (double*)base; base += vr_bytes;
        double* acc_add = (double*)base; base += vr_bytes;
        cplx* T = (cplx*)base;
        DV ad{acc_dft, n * cols * dnum, cols, dnum}, vr{vmp_res, n * cols * bsz, cols, bsz}, aa{acc_add, n * cols * bsz, cols, bsz};
        const bool tail = M->fuse_tail && tail_supported(M);
        for (int b0 = 0; b0 + blk <= n_lwe; b0 += blk) {  // chunks_exact: a trailing partial block is ignored, as in the reference
            PZ_TRY(dev_dft_apply(M, B, 1, 0, ad, 0, rv, 0, cols, nullptr, T));                      // :319-321
            const int row_max = std::min(dnum * cols, cols * std::min(dnum, rsz));
            bool block_done = false;
            if (M->fuse_mid) PZ_TRY(br_block_step(M, acc_dft, ad.bs, acc_add, aa.bs, brk, pmat_doubles, row_max, cols * bsz, B, b0, blk, lwe_2n, lwe_bs, &block_done));
            if (!block_done) {
            PZ_HIP(hipMemsetAsync(acc_add, 0, (size_t)B * aa.bs * 8, M->stream));                     // :321
            for (int i = b0; i < b0 + blk; ++i) {                                                       // :324-337
                PZ_TRY(dev_vmp(M, B, vr, ad, brk + (size_t)i * pmat_doubles, dnum, cols, cols, bsz, 0));
                PZ_TRY(launch_xai_acc(M, acc_add, aa.bs, vmp_res, vr.bs, cols * bsz, B, lwe_2n, lwe_bs, i));
            }
            }
            // acc = normalize(idft(acc_add) + acc)  (:342-346)
            if (tail) {
                PolyMap sm{bsz, cols, aa.bs, (long long)cols * n, n, 0};
                PZ_TRY(launch_inv_pass2(M, B * bsz * cols, acc_add, sm, T));
                PZ_TRY(launch_inv_tail(M, B, T, bsz, cols, (long long*)res, res_ct, cols, rsz, (const long long*)res, res_ct, cols, rsz, k,
                                       false, true));
            } else {
                PZ_TRY(dev_idft(M, B, aa, 0, aa, 0, cols, bsz, T));
                for (int c = 0; c < cols; ++c) {
                    PZ_TRY(launch_ew(M, EW_ADD_I64, (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n,
                                     (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n, res + (long long)c * n, res_ct,
                                     (long long)cols * n, std::min(bsz, rsz), B));
                    PZ_TRY(dev_normalize(M, B, rv, k, 0, c, aa, k, c));
                }
            }
        }
        return PZ_OK;
    }

    // standard: acc += (X^a_i - 1) * (acc (x) BRK_i) per coefficient, one normalization at the end (:423-437)
    pz_glwe_op_params ep;
    ep.rank = p->rank; ep.dnum = p->dnum; ep.dsize = 1; ep.key_size = p->brk_size; ep.key_base2k = p->base2k;
    ep.a_size = p->res_size; ep.a_base2k = p->base2k; ep.res_size = p->res_size; ep.res_base2k = p->base2k; ep.rank_out = p->rank;
    // acc_tmp lives in the module's second workspace: the external product owns the first one
    PZ_TRY(ws2_reserve(M, (size_t)B * res_ct * 8));
    int64_t* acc_tmp = (int64_t*)M->ws2;
    PolyMap pm{rsz, cols, res_ct, (long long)cols * n, n, 0};
    for (int i = 0; i < n_lwe; ++i) {
        PZ_TRY(glwe_op(M, false, acc_tmp, res, brk + (size_t)i * pmat_doubles, &ep, batch));
        PZ_TRY(launch_rotate(M, B * rsz * cols, (const long long*)acc_tmp, pm, (long long*)res, pm, 2, rsz * cols, (const long long*)lwe_2n,
                             lwe_bs, 1 + i, 0));
    }
    // vec_znx_normalize_assign (normalize.rs:403-425) == out-of-place same-base normalize of a copy
    PZ_HIP(hipMemcpyAsync(acc_tmp, res, (size_t)B * res_ct * 8, hipMemcpyDeviceToDevice, M->stream));
    DV tv{acc_tmp, res_ct, cols, rsz};
    for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, rv, k, 0, c, tv, k, c));
    return PZ_OK;
}

// execute_block_binary_extended (algorithm.rs:121-273; extension_factor > 1, block_size > 1): the ext accumulators of a
// ciphertext are one more batch dimension ([b][e]); per LWE block: batched forward DFT | per coefficient: batched VMP, then
// k_xai_ext moves the products between the accumulators as the reference does | inverse DFT + acc + carry chain (fused tail).
size_t pz_blind_rotation_extended_tmp_bytes(const pz_module* M, const pz_blind_rotation_params* p, size_t extension_factor, size_t batch) {
    if (!M || !p) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->rank + 1, be = batch * extension_factor;
    return align256(be * n8 * cols * p->res_size) + align256(be * n8 * cols * p->dnum) + 2 * align256(be * n8 * cols * p->brk_size) +
           align256(be * cols * std::max({(size_t)p->dnum, (size_t)p->brk_size, (size_t)p->res_size}) * (size_t)M->m * sizeof(cplx));
}
static int blind_rotation_extended(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                   const pz_blind_rotation_params* p, size_t extension_factor, void* tmp, size_t tmp_bytes, size_t batch) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->n_lwe >= 1 && p->block_size >= 1 && p->dnum >= 1 && p->brk_size >= 1 && p->res_size >= 1 && p->lut_size >= 1,
               "blind_rotation: empty shape");
    PZ_REQUIRE(p->base2k >= 1 && p->base2k <= 63, "blind_rotation: base2k out of range");
    PZ_REQUIRE(extension_factor >= 1 && (extension_factor & (extension_factor - 1)) == 0 && extension_factor <= 64,
               "blind_rotation: extension_factor must be a power of two");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(lwe_2n) && is_device_ptr(lut) && is_device_ptr(brk) && is_device_ptr(tmp),
               "batched entry points take device pointers");
    PZ_REQUIRE(tmp_bytes >= pz_blind_rotation_extended_tmp_bytes(M, p, extension_factor, batch), "blind_rotation: tmp is too small");
    if (batch == 0) return PZ_OK;
    PZ_TRY(ensure_w2n(M));
    int log_ext = 0;
    while (((size_t)1 << log_ext) < extension_factor) ++log_ext;
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1, dnum = (int)p->dnum, bsz = (int)p->brk_size, rsz = (int)p->res_size, k = (int)p->base2k;
    const int B = (int)batch, BE = B * (int)extension_factor, n_lwe = (int)p->n_lwe, blk = (int)p->block_size;
    const long long lwe_bs = (long long)n_lwe + 1, res_ct = n * cols * rsz;
    const size_t pmat_doubles = (size_t)M->n * dnum * cols * cols * bsz;
    const size_t n8 = (size_t)M->n * 8;
    char* base = (char*)tmp;
    int64_t* acc = (int64_t*)base; base += align256((size_t)BE * n8 * cols * rsz);
    double* acc_dft = (double*)base; base += align256((size_t)BE * n8 * cols * dnum);
    double* vmp_res = (double*)base; base += align256((size_t)BE * n8 * cols * bsz);
    double* acc_add = (double*)base; base += align256((size_t)BE * n8 * cols * bsz);
    cplx* T = (cplx*)base;
    // :159-161 zero, :180-190 rotated table
    PZ_HIP(hipMemsetAsync(acc, 0, (size_t)BE * res_ct * 8, M->stream));
    PZ_REQUIRE(BE <= 65535, "blind_rotation: batch * extension_factor exceeds 65535 (split the batch)");
    PZ_TRY(launch_br_ext_init(M, acc, lut, lwe_2n, lwe_bs, log_ext, cols, rsz, (int)p->lut_size, B));
    DV rv{acc, res_ct, cols, rsz};
    DV ad{acc_dft, n * cols * dnum, cols, dnum}, vr{vmp_res, n * cols * bsz, cols, bsz}, aa{acc_add, n * cols * bsz, cols, bsz};
    const bool tail = M->fuse_tail && tail_supported(M);
    for (int b0 = 0; b0 + blk <= n_lwe; b0 += blk) {
        PZ_TRY(dev_dft_apply(M, BE, 1, 0, ad, 0, rv, 0, cols, nullptr, T));                           // :195-200
        PZ_HIP(hipMemsetAsync(acc_add, 0, (size_t)BE * aa.bs * 8, M->stream));
        for (int i = b0; i < b0 + blk; ++i) {
            PZ_TRY(dev_vmp(M, BE, vr, ad, brk + (size_t)i * pmat_doubles, dnum, cols, cols, bsz, 0));   // :209-211
            PZ_TRY(launch_xai_ext(M, acc_add, vmp_res, cols * bsz, log_ext, B, lwe_2n, lwe_bs, i));
        }
        if (tail) {                                                                                    // :260-266
            PolyMap sm{bsz, cols, aa.bs, (long long)cols * n, n, 0};
            PZ_TRY(launch_inv_pass2(M, BE * bsz * cols, acc_add, sm, T));
            PZ_TRY(launch_inv_tail(M, BE, T, bsz, cols, (long long*)acc, res_ct, cols, rsz, (const long long*)acc, res_ct, cols, rsz, k, false, true));
        } else {
            PZ_TRY(dev_idft(M, BE, aa, 0, aa, 0, cols, bsz, T));
            for (int c = 0; c < cols; ++c) {
                PZ_TRY(launch_ew(M, EW_ADD_I64, (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n,
                                 (int64_t*)acc_add + (long long)c * n, aa.bs, (long long)cols * n, acc + (long long)c * n, res_ct,
                                 (long long)cols * n, std::min(bsz, rsz), BE));
                PZ_TRY(dev_normalize(M, BE, rv, k, 0, c, aa, k, c));
            }
        }
    }
    // :270-272 res = acc[0]
    return launch_ew(M, EW_COPY, res, res_ct, n, acc, (long long)extension_factor * res_ct, n, nullptr, 0, 0, cols * rsz, B);
}
int pz_blind_rotation_execute_extended_batched(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                               const pz_blind_rotation_params* p, size_t extension_factor, void* tmp, size_t tmp_bytes,
                                               size_t batch) {
    PZ_ENTER(M);
    return blind_rotation_extended(M, res, lwe_2n, lut, brk, p, extension_factor, tmp, tmp_bytes, batch);
}

int pz_blind_rotation_execute_batched(pz_module* M, int64_t* res, const int64_t* lwe_2n, const int64_t* lut, const double* brk,
                                      const pz_blind_rotation_params* p, size_t batch) {
    PZ_ENTER(M);
    KeyHash k;
    k.add((int)2); k.add(res); k.add(lwe_2n); k.add(lut); k.add(brk); k.add(batch);
    if (p) k.add(*p);
    graph_key_module(M, k);
    return with_graph(M, k.h, [&]() { return blind_rotation(M, res, lwe_2n, lut, brk, p, batch); });
}

// ------------------------------------------------------------------------------
// public: circuit bootstrapping LWE -> GGSW, constant mode, one base2k for every key and the result
// poulpy-bin-fhe/src/circuit_bootstrapping/circuit.rs:219-370 (circuit_bootstrap_core, to_exponent = false):
//   :321-331  acc = blind_rotation(lwe, lut)                                       (copy into the atk layout: same limbs)
//   :344-366  entry (i, 0) of the GGSW = glwe_trace(X^(-i*gap) * acc, skip 0)       (the reference rotates acc in place between rows)
//   :369      ggsw_expand_row
// The dnum_res traces of one LWE are independent, so all batch * dnum_res of them run as one batched trace.
// ------------------------------------------------------------------------------
static int glwe_pack(pz_module* M, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out,
                     const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p, void* tmp, size_t tmp_bytes,
                     size_t batch);
struct CbtRepack { size_t log_gap_in, log_gap_out, log_domain; };  // exponent mode with log_gap_in != log_gap_out (post_process)
static inline size_t cbt_tmp_size(const pz_circuit_bootstrapping_params* p) { return (size_t)std::max(p->br.res_size, p->res_size); }
static inline size_t cbt_ext(const pz_circuit_bootstrapping_params* p) { return p->extension_factor > 1 ? (size_t)p->extension_factor : 1; }
size_t pz_circuit_bootstrapping_tmp_bytes(const pz_module* M, const pz_circuit_bootstrapping_params* p, size_t batch) {
    if (!M || !p) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->br.rank + 1;
    const size_t ext_bytes = cbt_ext(p) > 1 ? align256(pz_blind_rotation_extended_tmp_bytes(M, &p->br, cbt_ext(p), batch)) : 0;
    return align256(batch * n8 * cols * p->br.res_size) + align256(batch * p->res_dnum * n8 * cols * cbt_tmp_size(p)) + ext_bytes;
}
size_t pz_circuit_bootstrapping_to_exponent_tmp_bytes(const pz_module* M, const pz_circuit_bootstrapping_params* p, size_t log_domain,
                                                      size_t batch) {
    if (!M || !p || log_domain > 20) return 0;
    const size_t n8 = (size_t)M->n * 8, cols = p->br.rank + 1;
    const size_t rows_ct = align256(batch * p->res_dnum * n8 * cols * cbt_tmp_size(p));
    // acc | rotated rows | 2^log_domain shifted copies | packed result | glwe_pack scratch (3 ciphertext arrays)
    return pz_circuit_bootstrapping_tmp_bytes(M, p, batch) + (((size_t)1 << log_domain) + 1 + 3) * rows_ct;
}
static int circuit_bootstrapping(pz_module* M, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut, const double* brk, size_t nsteps,
                                 const int64_t* gals, const double* const* atk_pmats, const double* const* tsk_pmats,
                                 const pz_circuit_bootstrapping_params* p, void* tmp, size_t tmp_bytes, size_t batch,
                                 const CbtRepack* rp = nullptr) {
    PZ_REQUIRE(p != nullptr, "null params");
    PZ_REQUIRE(p->res_dnum >= 1 && p->res_size >= 1 && p->atk_dnum >= 1 && p->atk_size >= 1 && p->tsk_dnum >= 1 && p->tsk_size >= 1,
               "circuit_bootstrapping: empty shape");
    PZ_REQUIRE(is_device_ptr(ggsw) && is_device_ptr(tmp), "batched entry points take device pointers");
    PZ_REQUIRE(tmp_bytes >= (rp ? pz_circuit_bootstrapping_to_exponent_tmp_bytes(M, p, rp->log_domain, batch)
                                : pz_circuit_bootstrapping_tmp_bytes(M, p, batch)),
               "circuit_bootstrapping: tmp is smaller than the *_tmp_bytes of this call");
    if (batch == 0) return PZ_OK;
    const long long n = (long long)M->n;
    const int cols = (int)p->br.rank + 1, gsz = (int)p->br.res_size, rsz = (int)p->res_size, tsz = (int)cbt_tmp_size(p);
    const int rows = (int)p->res_dnum, B = (int)batch;
    const long long ct_g = n * cols * gsz, ct_t = n * cols * tsz, ct_r = n * cols * rsz;
    int64_t* acc = (int64_t*)tmp;
    int64_t* tr = (int64_t*)((char*)tmp + align256((size_t)B * ct_g * 8));
    if (cbt_ext(p) > 1) {  // key.brk.execute dispatches on lut.extension_factor() (algorithm.rs:76-118); the scratch sits behind ours
        const size_t eb = align256(pz_blind_rotation_extended_tmp_bytes(M, &p->br, cbt_ext(p), batch));
        void* etmp = (char*)tmp + (rp ? pz_circuit_bootstrapping_to_exponent_tmp_bytes(M, p, rp->log_domain, batch)
                                      : pz_circuit_bootstrapping_tmp_bytes(M, p, batch)) - eb;
        PZ_TRY(blind_rotation_extended(M, acc, lwe_2n, lut, brk, &p->br, cbt_ext(p), etmp, eb, batch));
    } else {
        PZ_TRY(blind_rotation(M, acc, lwe_2n, lut, brk, &p->br, batch));
    }
    if (tsz > gsz) PZ_HIP(hipMemsetAsync(tr, 0, (size_t)B * rows * ct_t * 8, M->stream));  // glwe_copy zero-extends (glwe_trace.rs:114)
    for (int i = 0; i < rows; ++i) {
        PolyMap sm{gsz, cols, ct_g, (long long)cols * n, n, 0};
        PolyMap dm{gsz, cols, (long long)rows * ct_t, (long long)cols * n, n, (long long)i * ct_t};
        PZ_TRY(launch_rotate(M, B * gsz * cols, (const long long*)acc, sm, (long long*)tr, dm, 0, gsz * cols, nullptr, 0, 0,
                             -(long long)i * (long long)p->gap));
    }
    pz_glwe_op_params tp;
    tp.rank = p->br.rank; tp.dnum = p->atk_dnum; tp.dsize = 1; tp.key_size = p->atk_size; tp.key_base2k = p->br.base2k;
    tp.a_size = (uint64_t)tsz; tp.a_base2k = p->br.base2k; tp.res_size = (uint64_t)tsz; tp.res_base2k = p->br.base2k; tp.rank_out = p->br.rank;
    const int64_t* row_src = tr;
    if (!rp) {
        PZ_TRY(glwe_trace(M, tr, nsteps, gals, atk_pmats, &tp, (size_t)B * rows));
    } else {
        // post_process (circuit.rs:373-421) with log_gap_in != log_gap_out: partial trace, 2^log_domain shifted copies, glwe_pack
        size_t log_n = 0;
        while (((size_t)1 << log_n) < (size_t)M->n) ++log_n;
        PZ_REQUIRE(nsteps == log_n, "circuit_bootstrapping (exponent mode): gals / atk_pmats must cover all log2(n) trace steps");
        PZ_REQUIRE(rsz <= gsz, "circuit_bootstrapping (exponent mode): the GGSW must not have more limbs than the GLWE of the rotation");
        PZ_REQUIRE(rp->log_gap_in >= 1 && rp->log_gap_in <= log_n && rp->log_gap_out <= log_n && rp->log_domain <= 20 &&
                       (((size_t)1 << rp->log_domain) - 1) << rp->log_gap_out < (size_t)M->n,
                   "circuit_bootstrapping (exponent mode): gaps / domain out of range");
        const size_t skip = log_n - rp->log_gap_in + 1;
        PZ_TRY(glwe_trace(M, tr, log_n - skip, gals + skip, atk_pmats + skip, &tp, (size_t)B * rows));
        const size_t steps = (size_t)1 << rp->log_domain;
        const size_t rows_ct = align256((size_t)B * rows * ct_t * 8);
        char* base = (char*)tr + rows_ct;
        std::vector<int64_t*> cts(steps);
        std::vector<uint64_t> idx(steps);
        const PolyMap pm{tsz, cols, ct_t, (long long)cols * n, n, 0};
        for (size_t sidx = 0; sidx < steps; ++sidx) {
            cts[sidx] = (int64_t*)(base + sidx * rows_ct);
            idx[sidx] = (uint64_t)(sidx << rp->log_gap_out);
            PZ_TRY(launch_rotate(M, B * rows * tsz * cols, (const long long*)tr, pm, (long long*)cts[sidx], pm, 0, tsz * cols, nullptr, 0, 0,
                                 -(long long)(sidx << rp->log_gap_in)));
        }
        int64_t* packed = (int64_t*)(base + steps * rows_ct);
        void* pack_tmp = (void*)(base + (steps + 1) * rows_ct);
        PZ_TRY(glwe_pack(M, packed, steps, idx.data(), cts.data(), rp->log_gap_out, gals, atk_pmats, &tp, pack_tmp, 3 * rows_ct, (size_t)B * rows));
        row_src = packed;
    }
    // glwe_copy(res.at(i, 0), tmp) (glwe_trace.rs:121): the first res_size limbs, into the strided (row, 0) entries
    PZ_TRY(launch_ew(M, EW_COPY, ggsw, (long long)cols * ct_r, n, row_src, ct_t, n, nullptr, 0, 0, cols * rsz, B * rows));
    pz_glwe_op_params ep = tp;
    ep.dnum = p->tsk_dnum; ep.key_size = p->tsk_size; ep.a_size = (uint64_t)rsz; ep.res_size = (uint64_t)rsz;
    return ggsw_expand_row(M, ggsw, p->res_dnum, tsk_pmats, &ep, batch);
}
int pz_circuit_bootstrapping_execute_to_constant_batched(pz_module* M, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut,
                                                         const double* brk, size_t nsteps, const int64_t* gals,
                                                         const double* const* atk_pmats, const double* const* tsk_pmats,
                                                         const pz_circuit_bootstrapping_params* p, void* tmp, size_t tmp_bytes,
                                                         size_t batch) {
    PZ_ENTER(M);
    KeyHash k;
    k.add((int)3); k.add(ggsw); k.add(lwe_2n); k.add(lut); k.add(brk); k.add(nsteps); k.add(tmp); k.add(tmp_bytes); k.add(batch);
    if (p) {
        k.add(*p);
        for (size_t s = 0; s < nsteps && gals && atk_pmats; ++s) { k.add(gals[s]); k.add(atk_pmats[s]); }
        for (size_t c = 0; c < p->br.rank && tsk_pmats; ++c) k.add(tsk_pmats[c]);
    }
    graph_key_module(M, k);
    return with_graph(M, k.h, [&]() {
        return circuit_bootstrapping(M, ggsw, lwe_2n, lut, brk, nsteps, gals, atk_pmats, tsk_pmats, p, tmp, tmp_bytes, batch);
    });
}

// circuit_bootstrapping_execute_to_exponent (circuit.rs:197-216): equal gaps = the partial trace of post_process (:418-420),
// otherwise the repacking branch (:392-417).  gals / atk_pmats cover all log2(n) trace steps.
int pz_circuit_bootstrapping_execute_to_exponent_batched(pz_module* M, int64_t* ggsw, const int64_t* lwe_2n, const int64_t* lut,
                                                         const double* brk, const int64_t* gals, const double* const* atk_pmats,
                                                         const double* const* tsk_pmats, const pz_circuit_bootstrapping_params* p,
                                                         size_t log_gap_in, size_t log_gap_out, size_t log_domain, void* tmp,
                                                         size_t tmp_bytes, size_t batch) {
    PZ_ENTER(M);
    PZ_REQUIRE(gals != nullptr && atk_pmats != nullptr, "circuit_bootstrapping: null argument");
    size_t log_n = 0;
    while (((size_t)1 << log_n) < (size_t)M->n) ++log_n;
    PZ_REQUIRE(log_gap_in >= 1 && log_gap_in <= log_n, "circuit_bootstrapping (exponent mode): log_gap_in out of range");
    if (log_gap_in == log_gap_out) {
        const size_t skip = log_n - log_gap_in + 1;
        return circuit_bootstrapping(M, ggsw, lwe_2n, lut, brk, log_n - skip, gals + skip, atk_pmats + skip, tsk_pmats, p, tmp, tmp_bytes, batch);
    }
    CbtRepack rp{log_gap_in, log_gap_out, log_domain};
    return circuit_bootstrapping(M, ggsw, lwe_2n, lut, brk, log_n, gals, atk_pmats, tsk_pmats, p, tmp, tmp_bytes, batch, &rp);
}

// ------------------------------------------------------------------------------
// public: GLWEPacking::glwe_pack (poulpy-core/src/glwe_packing.rs:122-176, pack_internal :15-87) on `batch` independent packing
// problems with the same occupancy pattern, one base2k / size for ciphertexts, keys and result.  cts[s] points to the `batch`
// contiguous GLWEs of index indices[s] (the reference's HashMap entry); they are clobbered, as the reference's `&mut` entries.
// The tree is walked on the host exactly as the reference does; every step is a batched launch of the i64 kernels
// (rotate, add / sub, rsh, normalize) or of the fused automorphism pipeline.
// ------------------------------------------------------------------------------
size_t pz_glwe_pack_tmp_bytes(const pz_module* M, const pz_glwe_op_params* p, size_t batch) {
    if (!M || !p) return 0;
    return 3 * align256(batch * (size_t)M->n * (p->rank + 1) * p->res_size * 8);
}
static int glwe_pack(pz_module* M, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out,
                     const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p, void* tmp, size_t tmp_bytes,
                     size_t batch) {
    PZ_REQUIRE(p != nullptr && indices != nullptr && cts != nullptr && gals != nullptr && key_pmats != nullptr, "glwe_pack: null argument");
    PZ_REQUIRE(p->a_size == p->res_size && p->a_base2k == p->res_base2k && p->res_base2k == p->key_base2k && p->rank_out == p->rank &&
                   p->dsize == 1,
               "glwe_pack: ciphertexts, keys and result share base2k and size (the other cases re-normalize around this call)");
    size_t log_n = 0;
    while (((size_t)1 << log_n) < (size_t)M->n) ++log_n;
    PZ_REQUIRE(log_gap_out <= log_n && nslots >= 1, "glwe_pack: bad shape");
    PZ_REQUIRE(is_device_ptr(res) && is_device_ptr(tmp), "batched entry points take device pointers");
    PZ_REQUIRE(tmp_bytes >= pz_glwe_pack_tmp_bytes(M, p, batch), "glwe_pack: tmp is smaller than pz_glwe_pack_tmp_bytes");
    if (batch == 0) return PZ_OK;
    const long long n = (long long)M->n;
    const int cols = (int)p->rank + 1, size = (int)p->res_size, k = (int)p->res_base2k, B = (int)batch;
    const long long ct = n * cols * size;
    const size_t ctb = align256((size_t)B * ct * 8);
    int64_t* tmp_b = (int64_t*)tmp;
    int64_t* t1 = (int64_t*)((char*)tmp + ctb);
    int64_t* t2 = (int64_t*)((char*)tmp + 2 * ctb);
    std::vector<int64_t*> slots((size_t)M->n, nullptr);
    for (size_t s = 0; s < nslots; ++s) {
        PZ_REQUIRE(indices[s] < (uint64_t)M->n, "glwe_pack: index out of range");   // glwe_packing.rs:138
        PZ_REQUIRE(cts[s] != nullptr && is_device_ptr(cts[s]) && slots[indices[s]] == nullptr, "glwe_pack: bad or duplicate entry");
        slots[indices[s]] = cts[s];
    }
    const PolyMap pm{size, cols, ct, (long long)cols * n, n, 0};
    const int npolys = B * size * cols;
    auto rotate_to = [&](long long kk, int64_t* dst, const int64_t* src) {
        return launch_rotate(M, npolys, (const long long*)src, pm, (long long*)dst, pm, 0, size * cols, nullptr, 0, 0, kk);
    };
    auto rotate_assign = [&](long long kk, int64_t* x) {
        PZ_TRY(rotate_to(kk, t1, x));
        return launch_ew(M, EW_COPY, x, ct, n, t1, ct, n, nullptr, 0, 0, cols * size, B);
    };
    auto ew3 = [&](int op, int64_t* r, const int64_t* x, const int64_t* y) {   // limb-wise over whole ciphertexts (equal sizes)
        return launch_ew(M, op, r, ct, n, x, ct, n, y, ct, n, cols * size, B);
    };
    auto rsh1 = [&](int64_t* x) { return launch_rsh(M, B, (long long*)x, ct, cols, size, 0, cols, k, 1); };
    auto normalize_assign = [&](int64_t* x) {
        PZ_TRY(launch_ew(M, EW_COPY, t2, ct, n, x, ct, n, nullptr, 0, 0, cols * size, B));
        DV xv{x, ct, cols, size}, tv{t2, ct, cols, size};
        for (int c = 0; c < cols; ++c) PZ_TRY(dev_normalize(M, B, xv, k, 0, c, tv, k, c));
        return (int)PZ_OK;
    };
    for (size_t i = 0; i + log_gap_out < log_n; ++i) {
        const size_t tt = (size_t)1 << (log_n - 1 - i);
        PZ_REQUIRE((gals[i] & 1) != 0 && key_pmats[i] != nullptr, "glwe_pack: bad automorphism key");
        for (size_t j = 0; j < tt; ++j) {
            int64_t* a = slots[j];
            int64_t* b = slots[j + tt];
            slots[j] = nullptr;
            slots[j + tt] = nullptr;
            if (a && b) {                                                       // :41-70
                PZ_TRY(rotate_assign(-(long long)tt, a));
                PZ_TRY(ew3(EW_SUB_I64, tmp_b, a, b));
                PZ_TRY(rsh1(tmp_b));
                PZ_TRY(ew3(EW_ADD_I64, a, a, b));
                PZ_TRY(rsh1(a));
                PZ_TRY(normalize_assign(tmp_b));
                AutoSpec au{(long long)gals[i], 0};
                PZ_TRY(glwe_op(M, true, tmp_b, tmp_b, key_pmats[i], p, batch, &au));
                PZ_TRY(ew3(EW_SUB_I64, a, a, tmp_b));
                PZ_TRY(normalize_assign(a));
                PZ_TRY(rotate_assign((long long)tt, a));
                slots[j] = a;
            } else if (a) {                                                     // :71-75
                PZ_TRY(rsh1(a));
                AutoSpec au{(long long)gals[i], 1};
                PZ_TRY(glwe_op(M, true, a, a, key_pmats[i], p, batch, &au));
                slots[j] = a;
            } else if (b) {                                                     // :76-86
                PZ_TRY(rotate_to((long long)tt, tmp_b, b));
                PZ_TRY(rsh1(tmp_b));
                AutoSpec au{(long long)gals[i], 3};
                PZ_TRY(glwe_op(M, true, b, tmp_b, key_pmats[i], p, batch, &au));
                slots[j] = b;
            }
        }
    }
    PZ_REQUIRE(slots[0] != nullptr, "glwe_pack: no ciphertext ends at index 0");   // :175 a.get(&0).unwrap()
    PZ_TRY(launch_ew(M, EW_COPY, res, ct, n, slots[0], ct, n, nullptr, 0, 0, cols * size, B));
    const size_t skip = log_n - log_gap_out;
    return glwe_trace(M, res, log_n - skip, gals + skip, key_pmats + skip, p, batch);
}
int pz_glwe_pack_batched(pz_module* M, int64_t* res, size_t nslots, const uint64_t* indices, int64_t* const* cts, size_t log_gap_out,
                         const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p, void* tmp, size_t tmp_bytes,
                         size_t batch) {
    PZ_ENTER(M);
    return glwe_pack(M, res, nslots, indices, cts, log_gap_out, gals, key_pmats, p, tmp, tmp_bytes, batch);
}

}  // extern "C"

// api_lwe.hip composes the LWE <-> GLWE conversions around the batched key switch while holding the module lock
namespace pz {
int glwe_keyswitch_nolock(pz_module* M, int64_t* res, const int64_t* a, const double* key_pmat, const pz_glwe_op_params* p, size_t batch) {
    return glwe_entry(M, true, false, res, a, key_pmat, p, batch, nullptr);
}
}  // namespace pz

