// The 8-lane row transposes of k_mid128r's 128-point row DFT (16 complex values per lane, 8 lanes per row): through LDS (what the
// kernel does: 16 ds_write_b128 at stride 9, 16 ds_read_b128) against DPP lane exchanges (three butterfly stages xor 1 / 2 / 4 over
// the 8 lanes: quad_perm for 1 and 2, row_shl:4 / row_shr:4 with bank masks for 4).  Same launch shape as the kernel: 512 threads,
// one workgroup per CU, 2 waves per SIMD.  VERDICT r03 item 2c asked for the measurement; the count of instructions said
// 16 + 16 LDS instructions (~270 issue cycles per wave) against ~350 full-rate VALU instructions (~1400) on the unit that is the
// busiest of that kernel.
// build: hipcc -O3 --offload-arch=gfx950 -o row_transpose row_transpose.hip ; run: ./row_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double2 cplx;

__device__ __forceinline__ void row_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// new x_o[8h + oo] = old x_oo[8h + o]  (o = lane within the row)
__device__ __forceinline__ void transpose_lds(cplx (&x)[16], cplx* rowbuf, int o) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int oo = 0; oo < 8; ++oo) rowbuf[(8 * h + oo) * 9 + o] = x[8 * h + oo];
    row_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int oo = 0; oo < 8; ++oo) x[8 * h + oo] = rowbuf[(8 * h + o) * 9 + oo];
    row_sync();
}

// the same exchange with the real and imaginary parts in two planes: 8-byte LDS accesses, contiguous over the 8 lanes of a row
__device__ __forceinline__ void transpose_lds_planar(cplx (&x)[16], double* re, double* im, int o) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int oo = 0; oo < 8; ++oo) { re[(8 * h + oo) * 9 + o] = x[8 * h + oo].x; im[(8 * h + oo) * 9 + o] = x[8 * h + oo].y; }
    row_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int oo = 0; oo < 8; ++oo) { x[8 * h + oo].x = re[(8 * h + o) * 9 + oo]; x[8 * h + oo].y = im[(8 * h + o) * 9 + oo]; }
    row_sync();
}

template <int CTRL, int BANK>
__device__ __forceinline__ int dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, BANK, false); }

// one butterfly stage of the transpose on one dword of the cells (j, j | B) for all j with bit B clear
template <int B>
__device__ __forceinline__ void stage(int (&c)[8], bool hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j & B) continue;
        int& lo_cell = c[j];
        int& hi_cell = c[j | B];
        if constexpr (B == 4) {
            // lanes 0-3 of every 8 (banks 0, 2): hi_cell <- lane + 4's lo_cell; lanes 4-7 (banks 1, 3): lo_cell <- lane - 4's hi_cell
            const int t = dpp<0x114, 0xa>(0, hi_cell);          // row_shr:4 into banks 1, 3
            hi_cell = dpp<0x104, 0x5>(hi_cell, lo_cell);         // row_shl:4 into banks 0, 2
            lo_cell = dpp<0xe4, 0xa>(lo_cell, t);                // identity quad_perm, banks 1, 3 only
        } else {
            constexpr int ctrl = B == 1 ? 0xb1 : 0x4e;           // quad_perm [1,0,3,2] / [2,3,0,1]
            const int send = hi ? lo_cell : hi_cell;
            const int r = dpp<ctrl, 0xf>(0, send);
            lo_cell = hi ? r : lo_cell;
            hi_cell = hi ? hi_cell : r;
        }
    }
}

__device__ __forceinline__ void transpose_dpp(cplx (&x)[16], int o) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            int c[8];
#pragma unroll
            for (int oo = 0; oo < 8; ++oo) {
                const double v = (d < 2) ? x[8 * h + oo].x : x[8 * h + oo].y;
                c[oo] = (d & 1) ? __double2hiint(v) : __double2loint(v);
            }
            stage<1>(c, (o & 1) != 0);
            stage<2>(c, (o & 2) != 0);
            stage<4>(c, (o & 4) != 0);
#pragma unroll
            for (int oo = 0; oo < 8; ++oo) {
                double& v = (d < 2) ? x[8 * h + oo].x : x[8 * h + oo].y;
                v = (d & 1) ? __hiloint2double(c[oo], __double2loint(v)) : __hiloint2double(__double2hiint(v), c[oo]);
            }
        }
}

template <int MODE>   // 0: LDS, 1: DPP, 2: neither (loop overhead), 3: LDS, planar (two 8-byte accesses per value)
__global__ void __launch_bounds__(512) k_transpose(cplx* out, int iters) {
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x, row = tid >> 3, o = tid & 7;
    cplx* rowbuf = lds + row * 144;
    cplx x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = make_double2((double)(tid * 16 + i), (double)(blockIdx.x * 8192 + tid * 16 + i) + 0.5);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) transpose_lds(x, rowbuf, o);
        if (MODE == 1) transpose_dpp(x, o);
        if (MODE == 3) transpose_lds_planar(x, (double*)rowbuf, (double*)rowbuf + 144, o);
#pragma unroll
        for (int i = 0; i < 16; ++i) { x[i].x += 1.0; asm volatile("" : "+v"(x[i].y)); }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[((long long)blockIdx.x * 512 + tid) * 16 + i] = x[i];
}

int main() {
    const int grid = 256, iters = 2000;
    cplx* out[4];
    for (auto& p : out) hipMalloc(&p, (size_t)grid * 512 * 16 * sizeof(cplx));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 64 * 144 * sizeof(cplx);
    // the same LDS reservation for all three, so that each runs one workgroup per CU like the kernel
    (void)hipFuncSetAttribute((const void*)k_transpose<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_transpose<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_transpose<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_transpose<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms[4];
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_transpose<0><<<grid, 512, lds>>>(out[0], iters);
            if (mode == 1) k_transpose<1><<<grid, 512, lds>>>(out[1], iters);
            if (mode == 2) k_transpose<2><<<grid, 512, lds>>>(out[2], iters);
            if (mode == 3) k_transpose<3><<<grid, 512, lds>>>(out[3], iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
    }
    // one transpose: results of both exchanges must agree
    k_transpose<0><<<grid, 512, lds>>>(out[0], 1);
    k_transpose<1><<<grid, 512, lds>>>(out[1], 1);
    hipDeviceSynchronize();
    std::vector<cplx> a((size_t)grid * 512 * 16), b(a.size());
    hipMemcpy(a.data(), out[0], a.size() * sizeof(cplx), hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), out[1], b.size() * sizeof(cplx), hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += (a[i].x != b[i].x || a[i].y != b[i].y);
    printf("DPP transpose == LDS transpose on %zu values: %s (%zu differ)\n", a.size(), bad ? "NO" : "yes", bad);
    const double clk = 2.4e6;   // cycles per ms at 2.4 GHz (nominal; the ratio is what matters)
    for (int mode = 0; mode < 3; ++mode)
        printf("%-28s %8.3f ms for %d transposes of a 64-row tile per CU  -> %7.0f cycles per transpose (2 waves per SIMD in lockstep)\n",
               mode == 0 ? "LDS (stride 9, b128)" : mode == 1 ? "DPP (xor 1, 2, 4 stages)" : "neither (loop + increments)", ms[mode], iters,
               ms[mode] * clk / iters);
    k_transpose<3><<<grid, 512, lds>>>(out[3], 1);
    hipDeviceSynchronize();
    hipMemcpy(b.data(), out[3], b.size() * sizeof(cplx), hipMemcpyDeviceToHost);
    bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += (a[i].x != b[i].x || a[i].y != b[i].y);
    printf("planar LDS transpose == LDS transpose: %s (%zu differ); %8.3f ms -> %7.0f cycles per transpose, net %.0f\n", bad ? "NO" : "yes", bad, ms[3],
           ms[3] * clk / iters, (ms[3] - ms[2]) * clk / iters);
    printf("net: LDS %.0f cycles, DPP %.0f cycles per transpose; k_mid128r does 4 such transposes per tile of ~29 700 cycles\n",
           (ms[0] - ms[2]) * clk / iters, (ms[1] - ms[2]) * clk / iters);
    return 0;
}
