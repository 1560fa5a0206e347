// HBM streaming rates for the access shapes of the pipeline kernels: linear copy, read-only, write-only, and
// copy where each workgroup moves SEG-byte segments that are STRIDE bytes apart (pass 1 / tail / middle kernel shape).
// build: hipcc -O3 --offload-arch=gfx950 -o hbm_patterns hbm_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double2 cplx;

__global__ void __launch_bounds__(256) k_copy(const cplx* __restrict__ a, cplx* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_copy_nt(const cplx* __restrict__ a, cplx* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        cplx v = a[i];
        __builtin_nontemporal_store(v.x, &b[i].x);
        __builtin_nontemporal_store(v.y, &b[i].y);
    }
}
__global__ void __launch_bounds__(256) k_read(const cplx* __restrict__ a, double* out, long long n) {
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { cplx v = a[i]; s += v.x + v.y; }
    if (s == 1.234e-300) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write_nt(cplx* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        __builtin_nontemporal_store(1.0, &b[i].x);
        __builtin_nontemporal_store(2.0, &b[i].y);
    }
}
__global__ void __launch_bounds__(256) k_write(cplx* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = make_double2(1.0, 2.0);
}
// each workgroup owns column block cb of a [rows][rowlen] matrix of points and copies rows x SEGP points
template <int SEGP>
__global__ void __launch_bounds__(256) k_seg(const cplx* __restrict__ a, cplx* __restrict__ b, int rows, int rowlen, long long nmat) {
    const int lanes = SEGP, rpi = 256 / lanes;   // rows per iteration
    const int cbs = rowlen / SEGP;
    for (long long t = blockIdx.x; t < nmat * cbs; t += gridDim.x) {
        const long long mat = t / cbs; const int cb = t % cbs;
        const cplx* src = a + mat * rows * rowlen + cb * SEGP + (threadIdx.x % lanes);
        cplx* dst = b + mat * rows * rowlen + cb * SEGP + (threadIdx.x % lanes);
        for (int r = threadIdx.x / lanes; r < rows; r += rpi) dst[(long long)r * rowlen] = src[(long long)r * rowlen];
    }
}
int main() {
    const long long n = 1ll << 27;   // 2 GiB per buffer
    cplx *a, *b; double* out;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&out, 64);
    hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto f, const char* name, double bytes) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 3; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.2f TB/s\n", name, 3 * bytes / ms / 1e9);
    };
    const int G = 256 * 16;
    time([&] { k_copy<<<G, 256>>>(a, b, n); }, "linear copy (R+W bytes)", 2.0 * n * 16);
    time([&] { k_copy_nt<<<G, 256>>>(a, b, n); }, "linear copy, nontemporal stores", 2.0 * n * 16);
    time([&] { k_read<<<G, 256>>>(a, out, n); }, "read only", 1.0 * n * 16);
    time([&] { k_write<<<G, 256>>>(b, n); }, "write only", 1.0 * n * 16);
    time([&] { k_write_nt<<<G, 256>>>(b, n); }, "write only, nontemporal", 1.0 * n * 16);
    time([&] { hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy D2D (R+W bytes)", 2.0 * n * 16);
    // matrices of 128 rows x 256 points (one polynomial of the tall plan): segment = 16 points (256 B), stride 4 KiB
    time([&] { k_seg<16><<<G, 256>>>(a, b, 128, 256, n / (128 * 256)); }, "256 B segments, stride 4 KiB (pass1/tail)", 2.0 * n * 16);
    time([&] { k_seg<8><<<G, 256>>>(a, b, 128, 256, n / (128 * 256)); }, "128 B segments, stride 4 KiB", 2.0 * n * 16);
    time([&] { k_seg<64><<<G, 256>>>(a, b, 128, 256, n / (128 * 256)); }, "1 KiB segments, stride 4 KiB", 2.0 * n * 16);
    time([&] { k_seg<256><<<G, 256>>>(a, b, 128, 256, n / (128 * 256)); }, "4 KiB rows (contiguous)", 2.0 * n * 16);
    return 0;
}
