// Standalone GEMM laboratory (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I adaptersis_amd/csrc
//   scripts/gemm_lab.hip adaptersis_amd/csrc/core.hip -o /tmp/gemm_lab && /tmp/gemm_lab
// Times kernel variants (loads-only / math-only / no-epilogue) to locate the bottleneck.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_big.h"

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1);} } while (0)

template <typename K>
float time_kernel(K launch, int iters = 20) {
  hipEvent_t s, e;
  CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(s));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e));
  CK(hipEventSynchronize(e));
  float ms; CK(hipEventElapsedTime(&ms, s, e));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 21168, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 1024;
  std::vector<_Float16> ha((size_t)M * K), hb((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)(rnd() * 0.05f);
  _Float16 *A, *B, *C; float* bias;
  CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  asis_gemm_desc d = {};
  d.A = A; d.B = B; d.C = C; d.lda = K; d.ldb = K; d.ldc = N; d.batch = 1; d.M = M; d.N = N; d.K = K;
  d.bias_n = bias; d.act = 0; d.out_f32 = 0; d.dtype = 0;
  const double fl = 2.0 * M * N * K;
  printf("M=%d N=%d K=%d  (%.1f GFLOP)\n", M, N, K, fl / 1e9);
#define RUN(name, ...)                                                                        \
  {                                                                                           \
    const int bm = 256, bn = BNV;                                                             \
    dim3 grid(((M + bm - 1) / bm) * ((N + bn - 1) / bn)), block(512);                         \
    float ms = time_kernel([&]() { hipLaunchKernelGGL((gemm_big_kernel<__VA_ARGS__>), grid, block, 0, 0, d, 8); }); \
    CK(hipGetLastError());                                                                    \
    printf("%-34s %8.3f ms  %7.1f TFLOP/s\n", name, ms, fl / ms / 1e9);                     \
  }
#define BNV 256
  RUN("256x256 NS2 full", _Float16, 2, 4, 4, 2, 2, 0)
  RUN("256x256 NS2 no-epilogue", _Float16, 2, 4, 4, 2, 2, 4)
  RUN("256x256 NS2 math-only", _Float16, 2, 4, 4, 2, 2, 5)
  RUN("256x256 NS2 loads-only", _Float16, 2, 4, 4, 2, 2, 6)
#undef BNV
#define BNV 128
  RUN("256x128 NS3 full", _Float16, 4, 2, 2, 2, 3, 0)
  RUN("256x128 NS3 no-epilogue", _Float16, 4, 2, 2, 2, 3, 4)
  RUN("256x128 NS3 math-only", _Float16, 4, 2, 2, 2, 3, 5)
  RUN("256x128 NS3 loads-only", _Float16, 4, 2, 2, 2, 3, 6)
  RUN("256x128 BK32 NS3 occ4 full", _Float16, 4, 2, 2, 2, 3, 0, false, false, 32, 4)
  RUN("256x128 BK32 NS3 occ4 no-epi", _Float16, 4, 2, 2, 2, 3, 4, false, false, 32, 4)
  RUN("256x128 BK32 NS4 occ4 full", _Float16, 4, 2, 2, 2, 4, 0, false, false, 32, 4)
  return 0;
}
