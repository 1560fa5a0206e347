// module.hpp — pz_module: per-(N, device) state of the backend.
// Mirrors FFT64RefHandle { table_fft, table_ifft } (poulpy-cpu-ref/src/fft64/module.rs:34-69):
// immutable twiddle tables built once per Module, here resident in HBM, plus a
// stream and a grow-only device workspace for intermediates.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/poulpy_hip.h"
#include "device_fft.hpp"

namespace pz {

// Run-time switches.  rt_knob: the few a product build reads - alternative paths kept for cross-checks, each exercised by a -m gpu test
// (POULPY_DBG_CANARY, _GRAPHS, _SPLIT, _MID_R, _TENSOR_FUSED, _TENSOR_COMBINE, _TENSOR_ALLTERMS).  exp_knob: the switches of
// measured experiments (A/B records under profiles/): the default, as a constant, unless the library is built with -DPZ_EXPERIMENT
// (POULPY_BUILD_DEFS=-DPZ_EXPERIMENT POULPY_BUILD_TAG=exp ..., like -DPZ_ABLATE for the result-invalidating timing ablations).
inline int rt_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
inline int exp_knob(const char* name, int dflt) {
#ifdef PZ_EXPERIMENT
    return rt_knob(name, dflt);
#else
    (void)name;
    return dflt;
#endif
}

inline std::string& last_error_ref() {
    thread_local std::string e;
    return e;
}
inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define PZ_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return pz::fail(PZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define PZ_TRY(expr)          \
    do {                      \
        int r_ = (expr);      \
        if (r_ != PZ_OK) return r_; \
    } while (0)
#define PZ_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return pz::fail(PZ_ERR_INVALID, __VA_ARGS__); \
    } while (0)

struct FftPlan {
    int m1, m2;        // m = m1*m2, m2 >= m1
    int r1a, r1b;      // radices of the length-m1 pass (inverse direction / tail)
    int f1a, f1b;      // radices of the FORWARD length-m1 pass: smaller radix first, so that every thread of the
                       // workgroup issues global loads in the load stage
    int r2a, r2b;      // radices of the length-m2 pass
    int cb, qb;        // column block of pass 1 / q1 block of pass 2
};

inline bool radices_for(int L, int& ra, int& rb) {
    switch (L) {
        case 2: ra = 2; rb = 1; return true;   // N = 8, 16 (m = 2 x 2, 2 x 4)
        case 4: ra = 4; rb = 1; return true;
        case 8: ra = 8; rb = 1; return true;
        case 16: ra = 16; rb = 1; return true;
        case 32: ra = 8; rb = 4; return true;
        case 64: ra = 8; rb = 8; return true;
        case 128: ra = 16; rb = 8; return true;
        case 256: ra = 16; rb = 16; return true;
        default: return false;
    }
}

// n >= 8: the reference's smallest ring (poulpy-cpu-ref/src/reference/fft64/vmp.rs:67 asserts n >= 8; its backend tests run the
// convolution suite at Module::new(8), poulpy-cpu-ref/src/tests.rs:11-23; reim/fft_ref.rs:29-37 special-cases m <= 16).  N = 8 / 16 are
// m = 2 x 2 / 2 x 4 on the same four passes (radix-2 butterflies, 2- and 4-wide blocks): a handful of threads per polynomial - these
// rings exist for the reference's tests, not for throughput.
inline bool make_plan(uint64_t n, FftPlan& pl) {
    if (n < 8 || (n & (n - 1))) return false;
    uint64_t m = n >> 1;
    int k = 0;
    while ((1ull << k) < m) ++k;
    int k1 = k / 2;
    // N = 4096 / 8192: rows of 128 points (m1 = 16 / 32) instead of the balanced 32 x 64 / 64 x 64, so that the batched GLWE
    // pipeline can use the fused middle kernel (BASELINE configs[1] is N = 4096)
    if (k == 11) k1 = 4;
    if (k == 12) k1 = 5;
    // N = 65536: 256 x 128 ("wide"): rows of 128 points let the middle kernel hold four ciphertexts per tile, i.e. every key
    // value fetched from L2 serves four ciphertexts instead of two (middle kernel -16 %, external product +8 %); the
    // radix-16 x 16 column passes cost the same as the 8 x 16 ones of the 128 x 256 split once the tail's last butterfly is
    // shared by two threads (device_fft.hpp, SPLIT)
    if (k == 15) k1 = 8;
    if (const char* e = getenv("POULPY_DBG_SPLIT")) {  // diagnostic: force m1 > m2 ('w') or the balanced / m1 < m2 split ('t')
        if (e[0] == 'w' && (k & 1)) k1 = (k + 1) / 2;
        if (e[0] == 't') k1 = k / 2;
    }
    pl.m1 = 1 << k1;
    pl.m2 = 1 << (k - k1);
    if (!radices_for(pl.m1, pl.r1a, pl.r1b)) return false;
    if (!radices_for(pl.m2, pl.r2a, pl.r2b)) return false;
    pl.f1a = pl.r1a; pl.f1b = pl.r1b;
    if (pl.m1 == 16) { pl.f1a = 4; pl.f1b = 4; }
    if (pl.m1 == 32) { pl.f1a = 4; pl.f1b = 8; }
    if (pl.m1 == 128) { pl.f1a = 8; pl.f1b = 16; }
    pl.cb = pl.m2 >= 16 ? 16 : std::min(4, pl.m2);
    pl.qb = pl.m1 >= 16 ? 16 : std::min(4, pl.m1);
#ifdef PZ_EXPERIMENT
    if (const char* e = getenv("POULPY_DBG_CB")) pl.cb = atoi(e);  // experiment builds: column-block width of pass 1 / tail
#endif
    return true;
}

}  // namespace pz

struct pz_module {
    uint64_t n = 0, m = 0;
    int device = 0;
    pz::FftPlan plan{};
    hipStream_t stream = nullptr;
    // side stream of two-stream sections (SideStream below): a second chain of launches that runs beside the module stream and joins it again
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream_out = nullptr;   // device -> host copies of the duplex host path (api_glwe.hip: glwe_entry_duplex); stream2 carries its host -> device copies
    int cu_count = 0;   // CUs the module stream may use (0 = all; diagnostic POULPY_DBG_CU_MASK): grid of the persistent kernels
    // device tables (cplx): tw1[m1], tw1inv[m1], wL1[m1], wL2[m2], tw12[m] ([j2][q1])
    pz::cplx *tw1 = nullptr, *tw1inv = nullptr, *wL1 = nullptr, *wL2 = nullptr, *tw12 = nullptr;
    pz::cplx* tw12t = nullptr;  // the same table as [q1][j2] (row-major pipeline)
    pz::cplx* w2n = nullptr;    // exp(2 pi i t / 2n), t < 2n: DFT of the monomials X^a (blind rotation), built on first use
    // the seven tables above are immutable and shared by the siblings of pz_module_clone: freed by the last one (atomic count on the heap)
    std::atomic<int>* tables_ref = nullptr;
    // tables of the small-ring pipeline (m = M1 x 128; launch_small.hip): the module's own at N = 4096, built on first use otherwise
    pz::cplx *s_tw1 = nullptr, *s_tw1inv = nullptr, *s_tw12t = nullptr, *s_wL2 = nullptr;
    bool s_owned = false;
    // grow-only workspace
    void* ws = nullptr;
    size_t ws_bytes = 0;
    void* ws2 = nullptr;   // second grow-only buffer for callers that nest an operation which owns `ws`
    size_t ws2_bytes = 0;
    // staging arena for host-pointer calls: chunks are kept for the module's lifetime and the bump
    // pointer is reset at the start of every call (calls are serialised by `mu` and end with a sync)
    struct Chunk { void* p; size_t bytes; };
    std::vector<Chunk> arena;
    size_t arena_chunk = 0, arena_off = 0;
    struct PendingOut { void* host; const void* dev; size_t bytes; };
    std::vector<PendingOut> pending_out;
    std::mutex mu;
    unsigned long long* margin = nullptr;  // device word, bits of max |x-round(x)|; the word behind it: the "wide digits" flag of the 16-bit body
                                           // pre-pass of the spectral automorphism forms (wide16 below)
    unsigned* wide16() const { return reinterpret_cast<unsigned*>(margin + 1); }
    bool probe = false;
    size_t chunk = 0;
    size_t ws_shift = 0;   // diagnostic: extra bytes of padding in front of T2' in the fused workspace (placement experiments)
    int dbg_stages = 7;  // diagnostic: bit 0 pass 1, bit 1 middle, bit 2 tail of the fused pipeline (results invalid unless 7)
    // prepared keys the caller declared immutable (pz_module_pin_key): their row-sliced copies for the fused pipeline
    struct PinnedKey { const void* key; pz::cplx* sliced; size_t bytes; };
    std::vector<PinnedKey> pinned;
    // device mirrors of HOST-resident prepared keys handed to the batched GLWE entry points (the Rust shim's prepared layouts
    // live in pinned host memory, poulpy-hal requires host-addressable buffers): uploaded on first use, re-validated on every
    // call by a sampled fingerprint of the host bytes, dropped by pz_vmp_prepare / pz_vmp_zero / pz_module_forget_host_key
    // (round 3: + the process-wide invalidation epoch it was validated at: pz_vmp_prepare / pz_vmp_zero / pz_module_forget_host_key /
    //  pz_free_bytes on ANY module or thread publish the host range they touch, api.hip host_key_invalidate)
    struct KeyMirror { const void* host; size_t bytes; void* dev; uint64_t fp; uint64_t stamp; uint64_t epoch; };
    std::vector<KeyMirror> mirrors;
    uint64_t mirror_clock = 0;
    uint64_t mirror_seen_epoch = 0;   // process-wide invalidation epoch at this module's last sweep of its mirrors (api.hip)
    // RCCL communicator for pz_bcast_key (api_dist.hip); owned by the module
    void* comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    bool fuse_tail = true, fuse_mid = true;  // kernel-fusion knobs of the batched GLWE ops (tests run both settings)
    bool small_path = true;                  // N = 4096: the two-kernel pipeline of device_small.hpp where it applies
    // per-kernel-class HIP-event timing (bench.py's roofline leg); off by default
    bool timing = false;
    struct Timed { int cls; hipEvent_t e0, e1; };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> event_pool;
    double cls_ms[PZ_KCLASS_COUNT] = {0};
    unsigned long long cls_count[PZ_KCLASS_COUNT] = {0};
    // HIP graphs of the launch-bound composite calls (blind rotation, trace, circuit bootstrapping): a call seen twice with the
    // same arguments is captured once and replayed afterwards (api.hip, with_graph)
    struct GraphEntry { uint64_t key; hipGraph_t graph; hipGraphExec_t exec; uint64_t stamp; bool failed; };
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0, graph_epoch = 0;  // epoch: bumped by anything that changes what a captured call would launch
    bool graphs_on = true;
    unsigned long long graph_launches = 0;
    // POULPY_DBG_CANARY=1 (debug; tests/conftest.py runs the GPU suite with it once): a 256-byte guard behind every segment carved out
    // of the workspaces and behind the end of every reservation, armed on the module stream when it is carved and verified when the
    // API call returns (canary_verify, from PZ_ENTER's scope object).  An overrun into the slack between segments stays bit-exact
    // and therefore invisible to the parity tests (the spectral automorphism's body operand).
    std::vector<void*> guards;
    // the distinct kernel instantiations the hot dispatch sites chose since the last pz_module_dispatch_notes(reset) (bench tools print
    // them next to their numbers: "which variant ran" is part of a measurement)
    std::vector<std::string> notes;
};

namespace pz {

// RAII bracket: records an event pair around one kernel launch when timing is on
struct KTimer {
    pz_module* M;
    int cls;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    static hipEvent_t get(pz_module* M) {
        if (!M->event_pool.empty()) { hipEvent_t e = M->event_pool.back(); M->event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    KTimer(pz_module* M_, int cls_) : M(M_), cls(cls_) {
        if (!M->timing || M->timed.size() >= (1u << 16)) return;
        e0 = get(M); e1 = get(M);
        if (e0 && e1) (void)hipEventRecord(e0, M->stream);
    }
    ~KTimer() {
        if (e0 && e1) {
            (void)hipEventRecord(e1, M->stream);
            M->timed.push_back({cls, e0, e1});
        }
    }
};

inline void dispatch_note(pz_module* M, const char* fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    for (auto& s : M->notes) if (s == buf) return;
    if (M->notes.size() < 32) M->notes.emplace_back(buf);
#ifdef PZ_EXPERIMENT
    // experiment builds (-DPZ_EXPERIMENT): every new note of a module also goes to the file POULPY_DBG_DISPATCH_LOG names - the census of
    // instantiations a test / bench run dispatches (what tools/dbg/dispatch_census.sh collects before instantiations are pruned)
    if (const char* path = getenv("POULPY_DBG_DISPATCH_LOG")) {
        if (FILE* f = fopen(path, "a")) { fprintf(f, "%s\n", buf); fclose(f); }
    }
#endif
}

// Two-stream section of one API call: independent halves of a batch whose kernels are bound by different units (a compute-bound product beside a
// memory-bound transform) run on the module stream and on a side stream and overlap on the device.  fork(): the side stream waits for everything
// issued so far; on(side): the launchers' M->stream; the destructor (or join()) makes the module stream wait for the side stream and restores
// M->stream.  Legal under stream capture (fork / join by events: the graph gets two parallel branches).
struct SideStream {
    pz_module* M;
    hipStream_t main;
    bool forked = false;
    explicit SideStream(pz_module* M_) : M(M_), main(M_->stream) {}
    int fork() {
        if (!M->stream2) {
            if (hipStreamCreateWithFlags(&M->stream2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); M->stream2 = nullptr; return PZ_ERR_HIP; }
            if (hipEventCreateWithFlags(&M->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&M->ev_join, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                return PZ_ERR_HIP;
            }
        }
        if (hipEventRecord(M->ev_fork, main) != hipSuccess || hipStreamWaitEvent(M->stream2, M->ev_fork, 0) != hipSuccess) { (void)hipGetLastError(); return PZ_ERR_HIP; }
        forked = true;
        return PZ_OK;
    }
    void on(bool side) { M->stream = (side && forked) ? M->stream2 : main; }
    int join() {
        M->stream = main;
        if (!forked) return PZ_OK;
        forked = false;
        if (hipEventRecord(M->ev_join, M->stream2) != hipSuccess || hipStreamWaitEvent(main, M->ev_join, 0) != hipSuccess) { (void)hipGetLastError(); return PZ_ERR_HIP; }
        return PZ_OK;
    }
    ~SideStream() { (void)join(); }
};

constexpr size_t kGuardBytes = 256;
constexpr size_t kGuardSlack = 64 * kGuardBytes;   // room for the guards of a call's segments: part of every reservation
constexpr unsigned char kGuardByte = 0xC5;
inline bool canary_mode() {
    static const bool on = (rt_knob("POULPY_DBG_CANARY", 0) != 0);
    return on;
}
// arms a guard at p (canary mode only)
inline int guard_arm(pz_module* M, void* p) {
    if (!canary_mode()) return PZ_OK;
    PZ_HIP(hipMemsetAsync(p, kGuardByte, kGuardBytes, M->stream));
    M->guards.push_back(p);
    return PZ_OK;
}
// One segment of `bytes` at `base`; in canary mode a guard sits right behind it (and the next segment starts behind the guard).
// Call sites reserve kGuardSlack on top of the sum of their segments (ws_reserve does it for them).
template <typename T>
inline int ws_take(pz_module* M, char*& base, size_t bytes, T** out) {
    *out = reinterpret_cast<T*>(base);
    base += bytes;
    if (canary_mode() && bytes) {
        PZ_TRY(guard_arm(M, base));
        base += kGuardBytes;
    }
    return PZ_OK;
}
// verifies and clears the guards armed since the call started; aborts loudly on an overrun (debug mode only)
inline void canary_verify(pz_module* M, const char* where, const void* lo = nullptr, const void* hi = nullptr) {
    if (M->guards.empty()) return;
    std::vector<void*> gs, keep;
    for (void* g : M->guards) ((lo == nullptr || ((const char*)g >= (const char*)lo && (const char*)g < (const char*)hi)) ? gs : keep).push_back(g);
    M->guards.swap(keep);
    if (gs.empty()) return;
    if (hipStreamSynchronize(M->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    unsigned char host[kGuardBytes];
    for (void* g : gs) {
        if (hipMemcpy(host, g, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); continue; }
        for (size_t i = 0; i < kGuardBytes; ++i)
            if (host[i] != kGuardByte) {
                fprintf(stderr, "[poulpy_hip] WORKSPACE OVERRUN in %s: guard at %p (ws %p + %zu, ws2 %p) byte %zu = %#x\n", where, g, M->ws,
                        (size_t)((char*)g - (char*)M->ws), M->ws2, i, (unsigned)host[i]);
                fflush(stderr);
                abort();
            }
    }
}

// hipMalloc that first gives back the device mirrors of dead host keys - this module's and every idle module's - when memory runs
// out (api.hip; the caller holds M->mu)
int device_malloc_retry(pz_module* M, void** out, size_t bytes);

inline int ws_reserve(pz_module* M, size_t bytes) {
    // a new carve: the guards of the previous one (a composite call carves the workspace several times, with different layouts) are
    // verified now, before their bytes become someone else's segment
    if (canary_mode() && M->ws) canary_verify(M, "a workspace segment carved earlier in this call", M->ws, (char*)M->ws + M->ws_bytes);
    const size_t need = bytes + kGuardSlack;
    if (need > M->ws_bytes) {
        PZ_HIP(hipStreamSynchronize(M->stream));
        if (M->ws) PZ_HIP(hipFree(M->ws));
        M->ws = nullptr;
        M->ws_bytes = 0;
        size_t want = need + (need >> 3);
        PZ_TRY(device_malloc_retry(M, &M->ws, want));
        M->ws_bytes = want;
    }
    // the end of what this call asked for (+ the room of its inner guards): nothing may be written from here on
    return guard_arm(M, (char*)M->ws + need - kGuardBytes);
}

inline int ws2_reserve(pz_module* M, size_t bytes) {
    if (canary_mode() && M->ws2) canary_verify(M, "a workspace segment (ws2) carved earlier in this call", M->ws2, (char*)M->ws2 + M->ws2_bytes);
    const size_t need = bytes + kGuardSlack;
    if (need > M->ws2_bytes) {
        PZ_HIP(hipStreamSynchronize(M->stream));
        if (M->ws2) PZ_HIP(hipFree(M->ws2));
        M->ws2 = nullptr;
        M->ws2_bytes = 0;
        PZ_TRY(device_malloc_retry(M, &M->ws2, need));
        M->ws2_bytes = need;
    }
    return guard_arm(M, (char*)M->ws2 + need - kGuardBytes);
}

inline void arena_reset(pz_module* M) {
    M->arena_chunk = 0;
    M->arena_off = 0;
    M->pending_out.clear();
}
inline int arena_alloc(pz_module* M, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    while (M->arena_chunk < M->arena.size()) {
        auto& c = M->arena[M->arena_chunk];
        if (M->arena_off + bytes <= c.bytes) {
            *out = (char*)c.p + M->arena_off;
            M->arena_off += bytes;
            return PZ_OK;
        }
        M->arena_chunk++;
        M->arena_off = 0;
    }
    size_t want = std::max<size_t>(bytes, (size_t)8 << 20);
    if (!M->arena.empty()) want = std::max(want, 2 * M->arena.back().bytes);
    void* p = nullptr;
    PZ_TRY(device_malloc_retry(M, &p, want));
    M->arena.push_back({p, want});
    M->arena_chunk = M->arena.size() - 1;
    M->arena_off = bytes;
    *out = p;
    return PZ_OK;
}

inline void root_of_unity(long long num, long long den, double& c, double& s) {
    // exp(2*pi*i*num/den) evaluated in long double, rounded once to double
    num %= den;
    if (num < 0) num += den;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    long double ang = two_pi * (long double)num / (long double)den;
    c = (double)cosl(ang);
    s = (double)sinl(ang);
    // exact values on the axes
    if (num == 0) { c = 1; s = 0; }
    else if (4 * num == den) { c = 0; s = 1; }
    else if (2 * num == den) { c = -1; s = 0; }
    else if (4 * num == 3 * den) { c = 0; s = -1; }
}

inline int upload_table(cplx** dst, const std::vector<cplx>& h) {
    PZ_HIP(hipMalloc(dst, h.size() * sizeof(cplx)));
    PZ_HIP(hipMemcpy(*dst, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice));
    return PZ_OK;
}

inline int build_tables(pz_module* M) {
    const long long m = (long long)M->m;
    const int m1 = M->plan.m1, m2 = M->plan.m2;
    std::vector<cplx> h;
    double c, s;
    h.resize(m1);
    for (int j1 = 0; j1 < m1; ++j1) { root_of_unity(j1, 4ll * m1, c, s); h[j1] = make_double2(c, s); }
    PZ_TRY(upload_table(&M->tw1, h));
    const double inv_m = 1.0 / (double)m;  // exact power of two
    for (int j1 = 0; j1 < m1; ++j1) { h[j1].x = h[j1].x * inv_m; h[j1].y = -h[j1].y * inv_m; }
    PZ_TRY(upload_table(&M->tw1inv, h));
    for (int t = 0; t < m1; ++t) { root_of_unity(t, m1, c, s); h[t] = make_double2(c, s); }
    PZ_TRY(upload_table(&M->wL1, h));
    h.resize(m2);
    for (int t = 0; t < m2; ++t) { root_of_unity(t, m2, c, s); h[t] = make_double2(c, s); }
    PZ_TRY(upload_table(&M->wL2, h));
    h.resize((size_t)m);
    for (long long j2 = 0; j2 < m2; ++j2)
        for (long long q1 = 0; q1 < m1; ++q1) {
            root_of_unity(j2 * (4 * q1 + 1), 4 * m, c, s);
            h[(size_t)(j2 * m1 + q1)] = make_double2(c, s);
        }
    PZ_TRY(upload_table(&M->tw12, h));
    {
        std::vector<cplx> ht((size_t)m);
        for (long long j2 = 0; j2 < m2; ++j2)
            for (long long q1 = 0; q1 < m1; ++q1) ht[(size_t)(q1 * m2 + j2)] = h[(size_t)(j2 * m1 + q1)];
        PZ_TRY(upload_table(&M->tw12t, ht));
    }
    return PZ_OK;
}

}  // namespace pz
