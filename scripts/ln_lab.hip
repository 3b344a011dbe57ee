// Lab: LayerNorm forward variants against a plain convert-copy of the same traffic (fp32 rows in, 16-bit rows out).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ln_lab scripts/ln_lab.hip && gpurun_out/ln_lab [rows] [D]
#include "../adaptersis_amd/csrc/norm.hip"

namespace {

// ceiling: y = (T)x, grid-stride, 4 x 16-byte loads in flight per lane
__global__ __launch_bounds__(256) void copy_cvt_kernel(const float4* __restrict__ x, uint2* __restrict__ y, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x[i + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint2 p;
      p.x = pack2<f16>(v[k].x, v[k].y);
      p.y = pack2<f16>(v[k].z, v[k].w);
      y[i + k * stride] = p;
    }
  }
  for (; i < n4; i += stride) {
    const float4 v = x[i];
    uint2 p;
    p.x = pack2<f16>(v.x, v.y);
    p.y = pack2<f16>(v.z, v.w);
    y[i] = p;
  }
}

// pure write stream: 16 bytes per lane, grid-stride (what bn_relu_upsample's hi + lo outputs look like)
template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(uint4* __restrict__ y, int64_t n16, uint32_t v) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const uint4 w = make_uint4(v, v + 1, v + 2, v + 3);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 ww = {w.x, w.y, w.z, w.w};
    if (NT) __builtin_nontemporal_store(ww, reinterpret_cast<u4*>(y) + i);
    else y[i] = w;
  }
}

__device__ __forceinline__ float wave_sum_dpp(float v) {
  // row_shr / row_bcast reductions stay in the VALU (no LDS crossbar)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
  // lane 15 of each row holds the row sum; gather the four rows
  float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
  t += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  t += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 47));
  t += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return t;
}

// persistent: every wave walks rows gw, gw + nw, ... with the next row's loads issued before the reductions of this one
template <typename T, int NCH, bool DPP>
__global__ __launch_bounds__(256) void layernorm_stream_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                               const float* __restrict__ b, float eps, T* __restrict__ y,
                                                               int64_t ldy, int64_t rows) {
  constexpr int D = 256 * NCH;
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 ww[NCH], bb[NCH], v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) v[i] = reinterpret_cast<const float4*>(x + row * ldx)[lane + 64 * i];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    ww[i] = reinterpret_cast<const float4*>(w)[lane + 64 * i];
    bb[i] = reinterpret_cast<const float4*>(b)[lane + 64 * i];
  }
  for (; row < rows; row += nw) {
    float4 n[NCH];
    const int64_t nrow = row + nw < rows ? row + nw : row;
#pragma unroll
    for (int i = 0; i < NCH; ++i) n[i] = reinterpret_cast<const float4*>(x + nrow * ldx)[lane + 64 * i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = (DPP ? wave_sum_dpp(s) : wave_sum(s)) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = 1.0f / sqrtf((DPP ? wave_sum_dpp(q) : wave_sum(q)) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      uint2 p;
      p.x = pack2<T>((v[i].x - mean) * rstd * ww[i].x + bb[i].x, (v[i].y - mean) * rstd * ww[i].y + bb[i].y);
      p.y = pack2<T>((v[i].z - mean) * rstd * ww[i].z + bb[i].z, (v[i].w - mean) * rstd * ww[i].w + bb[i].w);
      reinterpret_cast<uint2*>(y + row * ldy)[lane + 64 * i] = p;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = n[i];
  }
}

// 8 consecutive elements per lane: 2 adjacent 16-byte loads, ONE 16-byte store; NCH2 = D / 512
template <typename T, int NCH2, int RPW>
__global__ __launch_bounds__(256) void layernorm_wide_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                             const float* __restrict__ b, float eps, T* __restrict__ y,
                                                             int64_t ldy, int64_t rows) {
  constexpr int D = 512 * NCH2;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  float4 v[RPW][NCH2][2];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
#pragma unroll
    for (int i = 0; i < NCH2; ++i) {
      v[r][i][0] = xr[2 * (lane + 64 * i)];
      v[r][i][1] = xr[2 * (lane + 64 * i) + 1];
    }
  }
  float4 ww[NCH2][2], bb[NCH2][2];
#pragma unroll
  for (int i = 0; i < NCH2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ww[i][h] = reinterpret_cast<const float4*>(w)[2 * (lane + 64 * i) + h];
      bb[i][h] = reinterpret_cast<const float4*>(b)[2 * (lane + 64 * i) + h];
    }
  float mean[RPW], rstd[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) s += (v[r][i][h].x + v[r][i][h].y) + (v[r][i][h].z + v[r][i][h].w);
    mean[r] = s;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) mean[r] = wave_sum(mean[r]) / (float)D;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float a0 = v[r][i][h].x - mean[r], a1 = v[r][i][h].y - mean[r], a2 = v[r][i][h].z - mean[r], a3 = v[r][i][h].w - mean[r];
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    rstd[r] = q;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(rstd[r]) / (float)D + eps);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r;
    if (row >= rows) break;
#pragma unroll
    for (int i = 0; i < NCH2; ++i) {
      uint4 p;
      uint32_t* pp = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 t = v[r][i][h], g = ww[i][h], o = bb[i][h];
        pp[2 * h] = pack2<T>((t.x - mean[r]) * rstd[r] * g.x + o.x, (t.y - mean[r]) * rstd[r] * g.y + o.y);
        pp[2 * h + 1] = pack2<T>((t.z - mean[r]) * rstd[r] * g.z + o.z, (t.w - mean[r]) * rstd[r] * g.w + o.w);
      }
      reinterpret_cast<uint4*>(y + row * ldy)[lane + 64 * i] = p;
    }
  }
}

}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 21180;
  const int D = argc > 2 ? atoi(argv[2]) : 1024;
  const int NB = 4;  // rotate buffers: 4 x (87 + 43) MB > 256 MB Infinity Cache
  float* x[NB];
  f16* y[NB];
  float *w, *b;
  for (int i = 0; i < NB; ++i) {
    CK(hipMalloc(&x[i], rows * D * 4));
    CK(hipMalloc(&y[i], rows * D * 2));
    CK(hipMemset(x[i], 0x3c, rows * D * 4));
  }
  CK(hipMalloc(&w, D * 4));
  CK(hipMalloc(&b, D * 4));
  CK(hipMemset(w, 0, D * 4));
  CK(hipMemset(b, 0, D * 4));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = (double)rows * D * 6;
  auto report = [&](const char* name, float ms, int n) {
    printf("%-46s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / n, bytes / (ms * 1e-3 / n) / 1e12);
  };
  const int IT = 40;
#define TIME(name, launch, rot)                                         \
  do {                                                                  \
    for (int it = 0; it < 4; ++it) { const int k = (rot) ? it % NB : 0; launch; } \
    CK(hipEventRecord(e0, s));                                          \
    for (int it = 0; it < IT; ++it) { const int k = (rot) ? it % NB : 0; launch; } \
    CK(hipEventRecord(e1, s));                                          \
    CK(hipEventSynchronize(e1));                                        \
    float ms;                                                           \
    CK(hipEventElapsedTime(&ms, e0, e1));                               \
    CK(hipGetLastError());                                              \
    report(name, ms, IT);                                               \
  } while (0)
  {
    // 1.4 GB of pure writes (two 16-bit outputs of the 294^2 x 256 decoder stage at B = 12)
    const int64_t nbytes = (int64_t)1400 << 20;
    uint4* big;
    CK(hipMalloc(&big, nbytes));
    for (int g : {2048, 8192, 32768}) {
      for (int nt = 0; nt < 2; ++nt) {
        for (int it = 0; it < 3; ++it) {
          if (nt) hipLaunchKernelGGL((fill_kernel<true>), dim3(g), dim3(256), 0, s, big, nbytes / 16, 7u);
          else hipLaunchKernelGGL((fill_kernel<false>), dim3(g), dim3(256), 0, s, big, nbytes / 16, 7u);
        }
        CK(hipEventRecord(e0, s));
        for (int it = 0; it < 10; ++it) {
          if (nt) hipLaunchKernelGGL((fill_kernel<true>), dim3(g), dim3(256), 0, s, big, nbytes / 16, 7u);
          else hipLaunchKernelGGL((fill_kernel<false>), dim3(g), dim3(256), 0, s, big, nbytes / 16, 7u);
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pure write 1.4 GB grid %5d %s: %8.1f us  %5.2f TB/s\n", g, nt ? "nontemporal" : "plain      ", ms * 1e3 / 10,
               (double)nbytes / (ms * 1e-3 / 10) / 1e12);
      }
    }
    CK(hipMemsetAsync(big, 0, nbytes, s));
    CK(hipEventRecord(e0, s));
    for (int it = 0; it < 10; ++it) CK(hipMemsetAsync(big, 0, nbytes, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemsetAsync 1.4 GB: %8.1f us  %5.2f TB/s\n", ms * 1e3 / 10, (double)nbytes / (ms * 1e-3 / 10) / 1e12);
    CK(hipFree(big));
  }
  for (int rot = 1; rot >= 0; --rot) {
    printf("---- rows %ld D %d, %s\n", (long)rows, D, rot ? "rotating 4 buffer sets (HBM)" : "one buffer set (cache-warm)");
    const int64_t n4 = rows * D / 4;
    for (int g : {1024, 2048, 4096, 8192})
      TIME((g == 1024 ? "copy_cvt grid 1024" : g == 2048 ? "copy_cvt grid 2048" : g == 4096 ? "copy_cvt grid 4096" : "copy_cvt grid 8192"),
           hipLaunchKernelGGL(copy_cvt_kernel, dim3(g), dim3(256), 0, s, (const float4*)x[k], (uint2*)y[k], n4), rot);
    if (D == 1024) {
      TIME("fixed RPW1 (grid rows/4)", hipLaunchKernelGGL((layernorm_fixed_kernel<f16, false, 4, 1>), dim3((unsigned)asis_cdiv(rows, 4)), dim3(256), 0, s, x[k], (int64_t)D, w, b, 1e-6f, y[k], (int64_t)D, rows), rot);
      TIME("fixed RPW2 (grid rows/8)  [current]", hipLaunchKernelGGL((layernorm_fixed_kernel<f16, false, 4, 2>), dim3((unsigned)asis_cdiv(rows, 8)), dim3(256), 0, s, x[k], (int64_t)D, w, b, 1e-6f, y[k], (int64_t)D, rows), rot);
      TIME("fixed RPW4 (grid rows/16)", hipLaunchKernelGGL((layernorm_fixed_kernel<f16, false, 4, 4>), dim3((unsigned)asis_cdiv(rows, 16)), dim3(256), 0, s, x[k], (int64_t)D, w, b, 1e-6f, y[k], (int64_t)D, rows), rot);
      TIME("wide RPW2 (16-byte stores)", hipLaunchKernelGGL((layernorm_wide_kernel<f16, 2, 2>), dim3((unsigned)asis_cdiv(rows, 8)), dim3(256), 0, s, x[k], (int64_t)D, w, b, 1e-6f, y[k], (int64_t)D, rows), rot);
      TIME("wide RPW4 (16-byte stores)", hipLaunchKernelGGL((layernorm_wide_kernel<f16, 2, 4>), dim3((unsigned)asis_cdiv(rows, 16)), dim3(256), 0, s, x[k], (int64_t)D, w, b, 1e-6f, y[k], (int64_t)D, rows), rot);
      for (int g : {512, 1024, 1536, 2048}) {
        char nm[64];
        snprintf(nm, sizeof nm, "stream grid %d", g);
        TIME(nm, hipLaunchKernelGGL((layernorm_stream_kernel<f16, 4, false>), dim3(g), dim3(256), 0, s, x[k], (int64_t)D, w, b, 1e-6f, y[k], (int64_t)D, rows), rot);
        snprintf(nm, sizeof nm, "stream dpp grid %d", g);
        TIME(nm, hipLaunchKernelGGL((layernorm_stream_kernel<f16, 4, true>), dim3(g), dim3(256), 0, s, x[k], (int64_t)D, w, b, 1e-6f, y[k], (int64_t)D, rows), rot);
      }
    }
  }
  return 0;
}

extern "C" void asis_set_error_(const char*, ...) {}
