// Tuned HBM copy micro-benchmark: can a plain copy reach the 6.29 TB/s the hardware guide records for a
// float4 copy, and with which issue pattern?  Sweeps grid size (k x 256 CUs), loads in flight per thread (U), 16 B per lane,
// plain vs non-temporal loads / stores, block-contiguous vs grid-strided indexing, and buffer size (0.5 - 4 GiB per buffer).
// build: hipcc -O3 --offload-arch=gfx950 -o hbm_copy_tuned hbm_copy_tuned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_copy_strided(const v4* __restrict__ a, v4* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n) v[u] = NTL ? __builtin_nontemporal_load(&a[i + u * stride]) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n) {
                if (NTS) __builtin_nontemporal_store(v[u], &b[i + u * stride]);
                else b[i + u * stride] = v[u];
            }
    }
}
// each workgroup owns a contiguous chunk (what a "one workgroup = one block of a polynomial" kernel does)
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_copy_chunked(const v4* __restrict__ a, v4* __restrict__ b, long long n) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = (lo + per < n) ? lo + per : n;
    for (long long i = lo + threadIdx.x; i < hi; i += 256 * U) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * 256 < hi) v[u] = NTL ? __builtin_nontemporal_load(&a[i + u * 256]) : a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * 256 < hi) {
                if (NTS) __builtin_nontemporal_store(v[u], &b[i + u * 256]);
                else b[i + u * 256] = v[u];
            }
    }
}
template <int U>
__global__ void __launch_bounds__(256) k_read(const v4* __restrict__ a, float* out, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    float s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) v[u] = a[i + u * stride]; else v[u] = (v4){0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 1.234e-30f) out[0] = s;
}
template <bool NTS>
__global__ void __launch_bounds__(256) k_write(v4* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    const v4 v = (v4){1, 2, 3, 4};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (NTS) __builtin_nontemporal_store(v, &b[i]); else b[i] = v;
    }
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const long long n = (long long)(gib * (1ll << 30)) / 16;
    v4 *a, *b; float* out;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&out, 64);
    hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto f, const char* name, int g, double bytes) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s grid %6d  %.2f TB/s\n", name, g, 5 * bytes / ms / 1e9);
    };
    printf("# buffers: %.2f GiB each\n", gib);
    for (int k : {2, 4, 8, 16, 32}) {
        const int G = 256 * k;
        time([&] { k_copy_strided<1, false, false><<<G, 256>>>(a, b, n); }, "copy strided U=1", G, 2.0 * n * 16);
        time([&] { k_copy_strided<4, false, false><<<G, 256>>>(a, b, n); }, "copy strided U=4", G, 2.0 * n * 16);
        time([&] { k_copy_strided<8, false, false><<<G, 256>>>(a, b, n); }, "copy strided U=8", G, 2.0 * n * 16);
        time([&] { k_copy_strided<4, false, true><<<G, 256>>>(a, b, n); }, "copy strided U=4 nt-store", G, 2.0 * n * 16);
        time([&] { k_copy_strided<4, true, true><<<G, 256>>>(a, b, n); }, "copy strided U=4 nt-load nt-store", G, 2.0 * n * 16);
        time([&] { k_copy_chunked<4, false, false><<<G, 256>>>(a, b, n); }, "copy chunked U=4", G, 2.0 * n * 16);
        time([&] { k_copy_chunked<4, true, true><<<G, 256>>>(a, b, n); }, "copy chunked U=4 nt-load nt-store", G, 2.0 * n * 16);
    }
    {   // one thread per element (no loop): n/256 workgroups
        const long long G = (n + 255) / 256;
        if (G < (1ll << 31)) {
            time([&] { k_copy_strided<1, false, false><<<(unsigned)G, 256>>>(a, b, n); }, "copy, one 16 B element per thread", (int)G, 2.0 * n * 16);
            time([&] { k_copy_strided<1, true, true><<<(unsigned)G, 256>>>(a, b, n); }, "copy, one element per thread, nt", (int)G, 2.0 * n * 16);
        }
    }
    for (int k : {4, 16}) {
        const int G = 256 * k;
        time([&] { k_read<4><<<G, 256>>>(a, out, n); }, "read only U=4", G, 1.0 * n * 16);
        time([&] { k_write<false><<<G, 256>>>(b, n); }, "write only", G, 1.0 * n * 16);
        time([&] { k_write<true><<<G, 256>>>(b, n); }, "write only nt-store", G, 1.0 * n * 16);
    }
    time([&] { hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy D2D", 0, 2.0 * n * 16);
    return 0;
}
