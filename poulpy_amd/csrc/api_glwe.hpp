// api_glwe.hpp — what the C-ABI translation units share above the launch layer: the batched GLWE product (api.hip), the composite
// calls built on it (api_br.hip: blind rotation, circuit bootstrapping, packing) and the HIP-graph cache for launch-bound chains.
#pragma once
#include "api_common.hpp"

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
    static const int env_on = rt_knob("POULPY_DBG_GRAPHS", 1);
    if (!env_on || !M->graphs_on || M->timing || canary_mode()) return body();   // (a replayed graph would not re-arm the workspace guards)
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


// Automorphism family on top of the key switch / strided ciphertexts: see glwe_op in api.hip
struct AutoSpec {
    long long p;
    int mode;
};
// ciphertexts that are not tightly packed (the entries of one column of a GGSW) and a body that lands in another column
struct OpLayout {
    long long a_stride, res_stride;  // in i64 elements between consecutive ciphertexts
    int body_col;
};

// api.hip: drops the device mirror of a host-resident prepared key (a rewritten buffer's mirror would be stale)
int forget_host_key(pz_module* M, const void* host);
void host_key_invalidate(const void* p, size_t bytes);   // api.hip: process-wide (every module's mirrors of that host range)
// api.hip: device pointer of a prepared key - itself when it is one, else its validated (possibly refreshed) device mirror
int resolve_key(pz_module* M, const double* pmat, size_t bytes, const double** out);

// Defined inside api.hip's extern "C" block (C linkage, internal use; the caller holds the module lock):
extern "C" {
// the batched GLWE product on device-resident data: external product (ks = false), key switch, automorphism family (au), strided
// ciphertexts with the body landing in another column (lay), tensor relinearization
int glwe_op(pz_module* M, bool ks, int64_t* res, const int64_t* a, const double* pmat, const pz_glwe_op_params* p, size_t batch,
            const AutoSpec* au = nullptr, const OpLayout* lay = nullptr, bool tensor = false, bool* post_rsh = nullptr);
// post_rsh (glwe_trace): in: the caller would like the result shifted right by one bit (vec_znx_rsh_assign on every column) as it is
// stored; out: whether the path taken did that (spectral automorphism forms on the 256 x 128 plan) - otherwise the caller shifts itself
// glwe_trace_assign on a batch (poulpy-core glwe_trace.rs:129-176): one prepared key per step
int glwe_trace(pz_module* M, int64_t* res, size_t nsteps, const int64_t* gals, const double* const* key_pmats, const pz_glwe_op_params* p,
               size_t batch);
// tensor relinearization of GLWETensors held as 16-bit digits in the fused tail's tile order (api_cnv.hip's fused multiply + relinearize)
bool glwe_relin_t16_supported(const pz_module* M, const pz_glwe_op_params* p);
size_t glwe_relin_chunk(const pz_module* M, const pz_glwe_op_params* p, size_t batch);
int glwe_relin_t16(pz_module* M, int64_t* res, const short* a16, long long a16_cs, const double* pmat, const pz_glwe_op_params* p, size_t batch);
// ggsw_expand_row on `count` GGSWs (conversion/gglwe_to_ggsw.rs:116-268)
int ggsw_expand_row(pz_module* M, int64_t* ggsw, size_t dnum, const double* const* tsk_pmat, const pz_glwe_op_params* p, size_t count);
}
