// What matrix rate does this MI355X SUSTAIN?  Bare f16 MFMA loops (operands in registers, no memory traffic inside the loop)
// on random and on all-zero data, both shapes, one and two waves per SIMD, ~0.1-0.2 s each; in-kernel clock from
// s_memtime / s_memrealtime (MI355X_MICROARCH.md, DVFS give-back item 6).  The nominal peak (2.5 PFLOP/s) assumes 2.4 GHz;
// a loop that the chip clocks at 1.6 GHz cannot exceed 1.67.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256) void mfma_loop(const f16x8* __restrict__ src, float* __restrict__ out, uint64_t* __restrict__ clk, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(tid * 8 + i) & 0xFFFF];
    b[i] = src[(tid * 8 + 4 + i) & 0xFFFF];
  }
  uint64_t t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  float res = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) res += acc[i][r];
  } else {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[i & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) res += acc[i][r];
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : "v"(res) : "memory");
  out[tid] = res;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  const int NSRC = 1 << 16;
  std::vector<_Float16> h(NSRC * 8);
  srand(1);
  f16x8* src;
  float* out;
  uint64_t* clk;
  hipMalloc(&src, NSRC * 16);
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMalloc(&clk, 256 * 8 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-10s %-8s %-12s %10s %10s %12s\n", "shape", "data", "waves/SIMD", "ms", "TFLOP/s", "clock MHz");
  for (int data = 0; data < 2; ++data) {
    for (size_t i = 0; i < h.size(); ++i) h[i] = data ? (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f) : (_Float16)0.f;
    hipMemcpy(src, h.data(), NSRC * 16, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape)
      for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;   // 256-thread blocks: one wave per SIMD each
        const int iters = 40000 / wps;
        const int per_iter = shape == 0 ? 16 : 32;
        const double flop = (double)blocks * 4 * iters * per_iter * (shape == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, src, out, clk, iters);
          else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, src, out, clk, iters);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          std::vector<uint64_t> hc(blocks * 2);
          hipMemcpy(hc.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
          double mhz = 0;
          for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / (double)hc[2 * i + 1] * 100.0;
          if (rep == 2)
            printf("%-10s %-8s %-12d %10.2f %10.0f %12.0f\n", shape == 0 ? "32x32x16" : "16x16x32", data ? "random" : "zeros", wps, ms,
                   flop / ms / 1e9, mhz / blocks);
        }
      }
  }
  return 0;
}
