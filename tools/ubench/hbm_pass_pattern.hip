// What do the HBM access SHAPES of pass 1 / the tail allow?  Copy kernels with exactly those shapes and no arithmetic:
// one workgroup per (polynomial, block of CB columns): reads 2 x 256 rows of CB x 8 B (the i64 halves of a limb, row stride 1 KiB),
// writes 256 rows of CB x 16 B (row stride 2 KiB) — pass 1; the tail is the mirror image.  CB = 16 is what the kernels do
// (128 B read runs, 256 B write runs); 32 / 64 / 128 would need wider column blocks.
// build: hipcc -O3 --offload-arch=gfx950 -o hbm_pass_pattern hbm_pass_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2 __attribute__((ext_vector_type(2)));

template <int CB, bool NT>
__global__ void __launch_bounds__(256) k_pass1_shape(const long long* __restrict__ a, v2* __restrict__ t, int npolys) {
    constexpr int M1 = 256, M2 = 128;
    const int ncb = M2 / CB;
    const int p = blockIdx.x / ncb, c0 = (blockIdx.x % ncb) * CB;
    const long long* src = a + (long long)p * 2 * M1 * M2;
    v2* dst = t + (long long)p * M1 * M2;
    const int c = threadIdx.x % CB, r0 = threadIdx.x / CB;
    constexpr int RPI = 256 / CB;                 // rows per iteration
#pragma unroll 4
    for (int r = r0; r < M1; r += RPI) {
        const long long idx = (long long)r * M2 + c0 + c;
        const long long re = NT ? __builtin_nontemporal_load(src + idx) : src[idx];
        const long long im = NT ? __builtin_nontemporal_load(src + idx + M1 * M2) : src[idx + M1 * M2];
        v2 v = {(double)re, (double)im};
        if (NT) __builtin_nontemporal_store(v, dst + idx); else dst[idx] = v;
    }
}
// variant: XCD-aware block order — the 8 column blocks of one polynomial run back to back on ONE XCD (workgroup ids go round-robin
// over the 8 XCDs), so that the eight 128 B pieces of every 1 KiB row are requested through one L2 close in time
template <int CB, bool NT>
__global__ void __launch_bounds__(256) k_pass1_xcd(const long long* __restrict__ a, v2* __restrict__ t, int npolys) {
    constexpr int M1 = 256, M2 = 128;
    const int ncb = M2 / CB;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p = (slot / ncb) * 8 + xcd, c0 = (slot % ncb) * CB;
    if (p >= npolys) return;
    const long long* src = a + (long long)p * 2 * M1 * M2;
    v2* dst = t + (long long)p * M1 * M2;
    const int c = threadIdx.x % CB, r0 = threadIdx.x / CB;
    constexpr int RPI = 256 / CB;
#pragma unroll 4
    for (int r = r0; r < M1; r += RPI) {
        const long long idx = (long long)r * M2 + c0 + c;
        const long long re = NT ? __builtin_nontemporal_load(src + idx) : src[idx];
        const long long im = NT ? __builtin_nontemporal_load(src + idx + M1 * M2) : src[idx + M1 * M2];
        v2 v = {(double)re, (double)im};
        if (NT) __builtin_nontemporal_store(v, dst + idx); else dst[idx] = v;
    }
}
// variant: T' stored BLOCKED as [polynomial][column block][row][CB columns], so that the 256 x (CB x 16 B) pieces one workgroup
// writes are ONE contiguous 64 KiB run (reads unchanged)
template <int CB, bool NT>
__global__ void __launch_bounds__(256) k_pass1_blocked(const long long* __restrict__ a, v2* __restrict__ t, int npolys) {
    constexpr int M1 = 256, M2 = 128;
    const int ncb = M2 / CB;
    const int p = blockIdx.x / ncb, cb = blockIdx.x % ncb, c0 = cb * CB;
    const long long* src = a + (long long)p * 2 * M1 * M2;
    v2* dst = t + (long long)p * M1 * M2 + (long long)cb * M1 * CB;
    const int c = threadIdx.x % CB, r0 = threadIdx.x / CB;
    constexpr int RPI = 256 / CB;
#pragma unroll 4
    for (int r = r0; r < M1; r += RPI) {
        const long long idx = (long long)r * M2 + c0 + c;
        const long long re = NT ? __builtin_nontemporal_load(src + idx) : src[idx];
        const long long im = NT ? __builtin_nontemporal_load(src + idx + M1 * M2) : src[idx + M1 * M2];
        v2 v = {(double)re, (double)im};
        if (NT) __builtin_nontemporal_store(v, dst + r * CB + c); else dst[r * CB + c] = v;
    }
}
// the tail's mirror image: reads T2' (row-major 256 B pieces, or blocked = one 64 KiB run), writes the two i64 halves (128 B runs)
template <int CB, bool NT, bool BLOCKED>
__global__ void __launch_bounds__(256) k_tail_shape(const v2* __restrict__ t, long long* __restrict__ a, int npolys) {
    constexpr int M1 = 256, M2 = 128;
    const int ncb = M2 / CB;
    const int p = blockIdx.x / ncb, cb = blockIdx.x % ncb, c0 = cb * CB;
    long long* dst = a + (long long)p * 2 * M1 * M2;
    const v2* src = t + (long long)p * M1 * M2 + (BLOCKED ? (long long)cb * M1 * CB : 0);
    const int c = threadIdx.x % CB, r0 = threadIdx.x / CB;
    constexpr int RPI = 256 / CB;
#pragma unroll 4
    for (int r = r0; r < M1; r += RPI) {
        const long long idx = (long long)r * M2 + c0 + c;
        const long long sidx = BLOCKED ? (long long)(r * CB + c) : idx;
        const v2 v = NT ? __builtin_nontemporal_load(src + sidx) : src[sidx];
        if (NT) { __builtin_nontemporal_store((long long)v.x, dst + idx); __builtin_nontemporal_store((long long)v.y, dst + idx + M1 * M2); }
        else { dst[idx] = (long long)v.x; dst[idx + M1 * M2] = (long long)v.y; }
    }
}
// the tail with the XCD-aware block order
template <int CB, bool NT>
__global__ void __launch_bounds__(256) k_tail_xcd(const v2* __restrict__ t, long long* __restrict__ a, int npolys) {
    constexpr int M1 = 256, M2 = 128;
    const int ncb = M2 / CB;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p = (slot / ncb) * 8 + xcd, c0 = (slot % ncb) * CB;
    if (p >= npolys) return;
    long long* dst = a + (long long)p * 2 * M1 * M2;
    const v2* src = t + (long long)p * M1 * M2;
    const int c = threadIdx.x % CB, r0 = threadIdx.x / CB;
    constexpr int RPI = 256 / CB;
#pragma unroll 4
    for (int r = r0; r < M1; r += RPI) {
        const long long idx = (long long)r * M2 + c0 + c;
        const v2 v = NT ? __builtin_nontemporal_load(src + idx) : src[idx];
        if (NT) { __builtin_nontemporal_store((long long)v.x, dst + idx); __builtin_nontemporal_store((long long)v.y, dst + idx + M1 * M2); }
        else { dst[idx] = (long long)v.x; dst[idx + M1 * M2] = (long long)v.y; }
    }
}
// the middle kernel's shape: 512 threads = 64 rows x 8 lanes; a tile = 64 rows (one frequency row q1 of 64 polynomials), every lane
// 16 x 16 B at stride 128 B: row-major rows are 2 KiB contiguous; blocked rows are 8 pieces of 256 B, 64 KiB apart.  Copy T' -> T2'.
template <bool NT, bool BLOCKED>
__global__ void __launch_bounds__(512) k_mid_shape(const v2* __restrict__ t, v2* __restrict__ t2, int npolys) {
    constexpr int M1 = 256, M2 = 128, CB = 16;
    const int row = threadIdx.x >> 3, o = threadIdx.x & 7;
    const int ntile_p = npolys / 64;
    for (int tile = blockIdx.x; tile < ntile_p * M1; tile += gridDim.x) {
        const int q1 = tile / ntile_p, p = (tile % ntile_p) * 64 + row;
        const long long base = (long long)p * M1 * M2;
        v2 x[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int j2 = o + 8 * n1;
            const long long idx = BLOCKED ? base + (long long)(j2 / CB) * M1 * CB + (long long)q1 * CB + (j2 % CB) : base + (long long)q1 * M2 + j2;
            x[n1] = NT ? __builtin_nontemporal_load(t + idx) : t[idx];
        }
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int j2 = o + 8 * n1;
            const long long idx = BLOCKED ? base + (long long)(j2 / CB) * M1 * CB + (long long)q1 * CB + (j2 % CB) : base + (long long)q1 * M2 + j2;
            if (NT) __builtin_nontemporal_store(x[n1], t2 + idx); else t2[idx] = x[n1];
        }
    }
}
int main() {
    const int npolys = 16384;                      // 1024 ciphertexts x 16 polynomials: 8 GiB in, 8 GiB out
    long long* a; v2* t;
    hipMalloc(&a, (size_t)npolys * 65536 * 8); hipMalloc(&t, (size_t)npolys * 32768 * 16);
    hipMemset(a, 1, (size_t)npolys * 65536 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto f, const char* name) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.2f TB/s   (%.2f ms per 1024 ciphertexts)\n", name, 5 * 2.0 * npolys * 65536 * 8 / ms / 1e9, ms / 5);
    };
    time([&] { k_pass1_shape<16, false><<<npolys * 8, 256>>>(a, t, npolys); }, "CB 16 (128 B reads, 256 B writes)");
    time([&] { k_pass1_shape<16, true><<<npolys * 8, 256>>>(a, t, npolys); }, "CB 16, non-temporal");
    time([&] { k_pass1_shape<32, false><<<npolys * 4, 256>>>(a, t, npolys); }, "CB 32 (256 B reads, 512 B writes)");
    time([&] { k_pass1_shape<32, true><<<npolys * 4, 256>>>(a, t, npolys); }, "CB 32, non-temporal");
    time([&] { k_pass1_shape<64, true><<<npolys * 2, 256>>>(a, t, npolys); }, "CB 64, non-temporal");
    time([&] { k_pass1_shape<128, true><<<npolys * 1, 256>>>(a, t, npolys); }, "CB 128 (whole rows), non-temporal");
    time([&] { k_pass1_xcd<16, false><<<npolys * 8, 256>>>(a, t, npolys); }, "pass 1, CB 16, column blocks of a polynomial on one XCD");
    time([&] { k_pass1_xcd<16, true><<<npolys * 8, 256>>>(a, t, npolys); }, "pass 1, CB 16, one XCD per polynomial, non-temporal");
    time([&] { k_pass1_blocked<16, false><<<npolys * 8, 256>>>(a, t, npolys); }, "pass 1, CB 16, T' blocked (64 KiB write runs)");
    time([&] { k_pass1_blocked<16, true><<<npolys * 8, 256>>>(a, t, npolys); }, "pass 1, CB 16, T' blocked, non-temporal");
    time([&] { k_tail_shape<16, true, false><<<npolys * 8, 256>>>(t, a, npolys); }, "tail, CB 16, row-major T2', non-temporal");
    time([&] { k_tail_shape<16, true, true><<<npolys * 8, 256>>>(t, a, npolys); }, "tail, CB 16, T2' blocked, non-temporal");
    time([&] { k_tail_xcd<16, true><<<npolys * 8, 256>>>(t, a, npolys); }, "tail, CB 16, one XCD per polynomial, non-temporal");
    time([&] { k_tail_xcd<16, false><<<npolys * 8, 256>>>(t, a, npolys); }, "tail, CB 16, one XCD per polynomial, plain");
    time([&] { k_tail_xcd<32, true><<<npolys * 4, 256>>>(t, a, npolys); }, "tail, CB 32, one XCD per polynomial, non-temporal");
    time([&] { k_tail_shape<32, true, false><<<npolys * 4, 256>>>(t, a, npolys); }, "tail, CB 32, row-major, non-temporal");
    v2* t2; hipMalloc(&t2, (size_t)npolys * 32768 * 16);
    time([&] { k_mid_shape<true, false><<<256, 512>>>(t, t2, npolys); }, "mid shape (persistent 256 WGs), row-major, nt");
    time([&] { k_mid_shape<true, true><<<256, 512>>>(t, t2, npolys); }, "mid shape, blocked, nt");
    time([&] { k_mid_shape<true, false><<<512, 512>>>(t, t2, npolys); }, "mid shape (512 WGs), row-major, nt");
    time([&] { k_mid_shape<true, true><<<512, 512>>>(t, t2, npolys); }, "mid shape (512 WGs), blocked, nt");
    return 0;
}
