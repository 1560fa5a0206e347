// Micro-benchmarks that calibrate the rooflines DESIGN.md prices against (gfx950 only).
//   1. v_fma_f64 issue rate (VALU FP64 peak)
//   2. v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 rates
//   3. VALU-FP64 + MFMA-FP64 co-issue (same wave interleaved / different waves)
//   4. HBM streaming copy with 16 B per lane
// Build: hipcc -O3 --offload-arch=gfx950 fp64_peaks.hip -o fp64_peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double s) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double b = s, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
      a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void __launch_bounds__(256) k_mfma16(double* out, int iters, double s) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * s, b = s;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  }
  d4 r = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}

__global__ void __launch_bounds__(256) k_mfma4(double* out, int iters, double s) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  double a = threadIdx.x * s, b = s;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

// same wave: 1 MFMA16 + NV independent FMAs per step
template <int NV>
__global__ void __launch_bounds__(256) k_mix_same(double* out, int iters, double s) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0;
  double a = threadIdx.x * s, b = s, c = 1e-9;
  double v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a + j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fma(v[j & 7], b, c);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fma(v[j & 7], b, c);
    }
  }
  d4 r = c0 + c1;
  double t = r[0] + r[1] + r[2] + r[3];
#pragma unroll
  for (int j = 0; j < 8; ++j) t += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// different waves: even waves MFMA only, odd waves FMA only (equal "peak time" each)
__global__ void __launch_bounds__(512) k_mix_waves(double* out, int iters, double s) {
  int wave = threadIdx.x >> 6;
  double t = 0;
  if (wave & 1) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = s, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
        a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
      }
    }
    t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  } else {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * s, b = s;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    d4 r = c0 + c1 + c2 + c3;
    t = r[0] + r[1] + r[2] + r[3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

__global__ void __launch_bounds__(256) k_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

// LDS exchange rate: every lane writes 16 B then reads 16 B from a transposed slot
__global__ void __launch_bounds__(256) k_lds(double* out, int iters) {
  __shared__ double2 buf[256 * 17];
  double2 v = {(double)threadIdx.x, 1.0};
  int t = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) buf[(u * 256 + t) + ((u * 256 + t) >> 4)] = v;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) { double2 w = buf[(t * 16 + u) + ((t * 16 + u) >> 4)]; v.x += w.x; v.y += w.y; }
    __syncthreads();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = v.x + v.y;
}

template <typename F>
static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz L2 %d B lds/blk %zu\n", p.name, p.multiProcessorCount, p.clockRate, p.l2CacheSize, p.sharedMemPerBlock);
  int cus = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, sizeof(double) * 512 * 8192));
  const int iters = 2000;
  for (int wpc : {4, 8, 16, 32}) {  // waves per CU
    int blocks = cus * wpc / 4;
    float ms = time_ms([&] { k_fma<<<blocks, 256>>>(out, iters, 0.999); }, 5);
    double flops = 2.0 * 64 * iters * (double)blocks * 256;  // 8 chains x 8 unrolled FMAs per iteration
    printf("fma_f64      waves/CU %2d: %8.3f ms  %7.2f TFLOP/s\n", wpc, ms, flops / ms * 1e-9);
  }
  for (int wpc : {4, 8, 16}) {
    int blocks = cus * wpc / 4;
    float ms = time_ms([&] { k_mfma16<<<blocks, 256>>>(out, iters, 0.5); }, 5);
    double flops = 2.0 * 16 * 16 * 4 * 16.0 * iters * (double)blocks * 4;
    printf("mfma16x16x4  waves/CU %2d: %8.3f ms  %7.2f TFLOP/s\n", wpc, ms, flops / ms * 1e-9);
    ms = time_ms([&] { k_mfma4<<<blocks, 256>>>(out, iters, 0.5); }, 5);
    flops = 2.0 * 4 * 4 * 4 * 4 * 16.0 * iters * (double)blocks * 4;
    printf("mfma4x4x4_4b waves/CU %2d: %8.3f ms  %7.2f TFLOP/s\n", wpc, ms, flops / ms * 1e-9);
  }
  {
    int blocks = cus * 2;
    auto rep = [&](const char* name, float ms, int nv) {
      double mf = 2.0 * 1024 * 8.0 * iters * (double)blocks * 4;
      double vf = 2.0 * 64 * 8.0 * nv * iters * (double)blocks * 4;
      printf("%s: %8.3f ms  mfma %7.2f + valu %7.2f = %7.2f TFLOP/s\n", name, ms, mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9);
    };
    rep("mix same wave NV=0 ", time_ms([&] { k_mix_same<0><<<blocks, 256>>>(out, iters, 0.5); }, 5), 0);
    rep("mix same wave NV=4 ", time_ms([&] { k_mix_same<4><<<blocks, 256>>>(out, iters, 0.5); }, 5), 4);
    rep("mix same wave NV=8 ", time_ms([&] { k_mix_same<8><<<blocks, 256>>>(out, iters, 0.5); }, 5), 8);
    rep("mix same wave NV=16", time_ms([&] { k_mix_same<16><<<blocks, 256>>>(out, iters, 0.5); }, 5), 16);
    int b2 = cus * 2;
    float ms = time_ms([&] { k_mix_waves<<<b2, 512>>>(out, iters, 0.5); }, 5);
    double mf = 2.0 * 1024 * 4.0 * iters * (double)b2 * 4;
    double vf = 2.0 * 64 * 64.0 * iters * (double)b2 * 4;  // 64 FMAs x 64 lanes per iteration, 4 VALU waves per block
    printf("mix diff waves     : %8.3f ms  mfma %7.2f + valu %7.2f = %7.2f TFLOP/s\n", ms, mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9);
  }
  {
    size_t n = (size_t)1 << 27;  // 2 GiB each way
    double2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16));
    for (int bpc : {8, 16, 32}) {
      float ms = time_ms([&] { k_copy<<<cus * bpc, 256>>>(a, b, n); }, 5);
      printf("copy 2x2GiB blocks/CU %2d: %8.3f ms  %7.2f GB/s (read+write)\n", bpc, ms, 2.0 * n * 16 / ms * 1e-6);
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  {
    int blocks = cus * 2;
    float ms = time_ms([&] { k_lds<<<blocks, 256>>>(out, 500); }, 5);
    double bytes = 2.0 * 16 * 16 * 256 * 500.0 * blocks;
    printf("lds exchange (w+r 16B, padded): %8.3f ms  %7.2f TB/s  (%.1f B/clk/CU at 2.4GHz)\n", ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / cus / 2.4);
  }
  return 0;
}
