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
    return 0;
}
