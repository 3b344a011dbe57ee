// Lab (standalone, not part of the library): s_memtime stamps of the persistent 8-phase GEMM's barrier intervals.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iadaptersis_amd/csrc scripts/p8_stamps.hip -o gpurun_out/p8_stamps
//   gpurun_out/p8_stamps            (on the GPU box)  -> per wave and phase: load part, wait at the opening barrier, burst + closing barrier
// gemm_p8.h LAB & 8 records, for workgroup 0's first tile and K tiles 4..11, per phase: load-part start, load-part end, burst start
// (stamps issued without a wait, stored behind the burst: the stamped kernel runs within a few percent of the plain one).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "asis_common.h"
namespace { __device__ __attribute__((aligned(16))) uint4 g_zero_page[1]; }
#include "gemm_p8.h"

__global__ void fill_kernel(_Float16* p, int64_t n, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (_Float16)(((int)(x & 0xffff) - 32768) / 32768.0f);
  }
}
__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, int64_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(cnt, c);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int M = 42348, N = 4096, K = 1024;
  _Float16 *A, *B, *C;
  float* bias;
  uint32_t* st;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&st, 8 * 128 * 4));
  CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(st, 0, 8 * 128 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, A, (int64_t)M * K, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, B, (int64_t)N * K, 7u);
  asis_gemm_desc d = {};
  d.A = A; d.B = B; d.C = C; d.lda = K; d.ldb = K; d.ldc = N; d.batch = 1; d.M = M; d.N = N; d.K = K;
  d.bias_n = bias; d.act = ASIS_ACT_GELU; d.dtype = ASIS_F16;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 300; ++it) hipLaunchKernelGGL((gemm_p8_kernel<f16, 0, 0>), dim3(256), dim3(512), 0, 0, d, 4);
  CK(hipEventRecord(e0));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((gemm_p8_kernel<f16, 0, 0>), dim3(256), dim3(512), 0, 0, d, 4);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("plain kernel: %.1f us per launch (%.0f TFLOP/s)\n", ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12);
  d.stats = reinterpret_cast<float*>(st);
  CK(hipEventRecord(e0));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((gemm_p8_kernel<f16, 8, 0>), dim3(256), dim3(512), 0, 0, d, 4);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("stamped kernel: %.1f us per launch\n", ms / 20 * 1e3);
  std::vector<uint32_t> h(8 * 128);
  CK(hipMemcpy(h.data(), st, 8 * 128 * 4, hipMemcpyDeviceToHost));
  // 8 K tiles x 4 phases x 3 stamps: load-part start (behind the previous closing barrier), load-part end (in front of the opening
  // barrier, behind the counted wait), burst start (behind the opening barrier)
  for (int w = 0; w < 8; ++w) {
    double load[4] = {0, 0, 0, 0}, bar1[4] = {0, 0, 0, 0}, rest[4] = {0, 0, 0, 0};
    int n = 0;
    for (int kt = 0; kt < 7; ++kt) {
      for (int p = 0; p < 4; ++p) {
        const uint32_t* s = &h[w * 128 + (kt * 4 + p) * 3];
        load[p] += (double)(uint32_t)(s[1] - s[0]);
        bar1[p] += (double)(uint32_t)(s[2] - s[1]);
        rest[p] += (double)(uint32_t)(s[3] - s[2]);     // burst + closing barrier (next phase's load-part start)
      }
      ++n;
    }
    printf("wave %d (row %d):", w, w >> 2);
    for (int p = 0; p < 4; ++p) printf("  p%d load %5.0f wait %5.0f burst+close %5.0f |", p, load[p] / n, bar1[p] / n, rest[p] / n);
    printf("  K tile %6.0f cycles\n", (double)(uint32_t)(h[w * 128 + 7 * 12] - h[w * 128]) / 7.0);
  }
  return 0;
}
