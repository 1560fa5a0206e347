// L2 -> CU streaming bandwidth with the access pattern of the fused middle kernel's product phase:
// every workgroup (one per CU, 512 threads) sweeps a 1 MiB key slice that the other workgroups of its XCD
// sweep too (L2 resident), 16 B per lane, NC independent loads in flight per lane and step.
// build: hipcc -O3 --offload-arch=gfx950 -o l2_stream l2_stream.hip ; run: ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double2 cplx;

template <int NC, bool ROT>
__global__ void __launch_bounds__(512) k_stream(const cplx* __restrict__ P, double* out, int rows_per_xcd, int reps) {
    const int tid = threadIdx.x, q2 = tid & 255, cg = tid >> 8;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    double s = 0.0;
    for (int rep = 0; rep < reps; ++rep) {
        const int k = rep % rows_per_xcd;
        const cplx* base = P + ((long long)(k * 8 + xcd) * 256) * 256 + q2;   // slice: 16 x 16 x 256 points
        const int rot = ROT ? (w & 15) : 0;
        for (int it = 0; it < 16; ++it) {
            const int r = (it + rot) & 15;
            cplx v[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) v[j] = base[(long long)(r * 16 + cg * NC + j) * 256];
#pragma unroll
            for (int j = 0; j < NC; ++j) s += v[j].x + v[j].y;
        }
    }
    if (s == 1.2345e-300) out[blockIdx.x] = s;
}

int main() {
    const int m1 = 128;
    const size_t bytes = (size_t)m1 * 256 * 256 * sizeof(cplx);   // 128 MiB: the metric key
    cplx* P; double* out;
    hipMalloc(&P, bytes); hipMalloc(&out, 4096);
    hipMemset(P, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int grid, int nc) {
        const int reps = 64;
        kern<<<grid, 512>>>(P, out, m1 / 8, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<grid, 512>>>(P, out, m1 / 8, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double b = (double)grid * reps * 16.0 * (2 * nc) * 256 * sizeof(cplx);
        printf("%-28s grid=%d  %.2f TB/s (L2->CU)\n", name, grid, b / ms / 1e9);
    };
    run(k_stream<8, true>, "NC=8 rotated rows", 256, 8);
    run(k_stream<8, false>, "NC=8 same row order", 256, 8);
    run(k_stream<4, true>, "NC=4 (half the columns)", 256, 4);
    run(k_stream<8, true>, "NC=8 rotated, 2 WG/CU", 512, 8);
    return 0;
}
