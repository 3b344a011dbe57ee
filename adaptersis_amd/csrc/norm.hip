// HBM-bound row kernels: LayerNorm, cls/pos-embed add, fp32->16-bit packing, patch im2col.
// One wave64 per row, 16-byte accesses, fp32 statistics (cdna_hip_programming.md G13).
#include "asis_common.h"

namespace {

constexpr int LN_MAXC = 8;  // float4 chunks per lane -> D <= 2048

template <typename T, bool OUT_F32>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, void* __restrict__ y, int64_t ldy, int64_t rows,
                                                        int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
  float4 v[LN_MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      v[i] = xr[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float4 ww = w4[c], bb = b4[c];
      const float o0 = (v[i].x - mean) * rstd * ww.x + bb.x;
      const float o1 = (v[i].y - mean) * rstd * ww.y + bb.y;
      const float o2 = (v[i].z - mean) * rstd * ww.z + bb.z;
      const float o3 = (v[i].w - mean) * rstd * ww.w + bb.w;
      if (OUT_F32) {
        reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + row * ldy)[c] = make_float4(o0, o1, o2, o3);
      } else {
        uint2 p;
        p.x = pack2<T>(o0, o1);
        p.y = pack2<T>(o2, o3);
        reinterpret_cast<uint2*>(reinterpret_cast<T*>(y) + row * ldy)[c] = p;
      }
    }
  }
}

// D == 256 * NCH exactly: no predicates, RPW rows per wave so that RPW * NCH 16-byte loads per lane are in flight
// before the first reduction, affine parameters fetched under the reductions.
template <typename T, bool OUT_F32, int NCH, int RPW>
__global__ __launch_bounds__(256) void layernorm_fixed_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              float eps, void* __restrict__ y, int64_t ldy,
                                                              int64_t rows) {
  constexpr int D = 256 * NCH;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  float4 v[RPW][NCH];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[r][i] = xr[lane + 64 * i];
  }
  float4 ww[NCH], bb[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    ww[i] = reinterpret_cast<const float4*>(w)[lane + 64 * i];
    bb[i] = reinterpret_cast<const float4*>(b)[lane + 64 * i];
  }
  float mean[RPW], rstd[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    mean[r] = s;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) mean[r] = wave_sum(mean[r]) / (float)D;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float a0 = v[r][i].x - mean[r], a1 = v[r][i].y - mean[r], a2 = v[r][i].z - mean[r],
                  a3 = v[r][i].w - mean[r];
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
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      const float o0 = (v[r][i].x - mean[r]) * rstd[r] * ww[i].x + bb[i].x;
      const float o1 = (v[r][i].y - mean[r]) * rstd[r] * ww[i].y + bb[i].y;
      const float o2 = (v[r][i].z - mean[r]) * rstd[r] * ww[i].z + bb[i].z;
      const float o3 = (v[r][i].w - mean[r]) * rstd[r] * ww[i].w + bb[i].w;
      if (OUT_F32) {
        reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + row * ldy)[c] = make_float4(o0, o1, o2, o3);
      } else {
        uint2 p;
        p.x = pack2<T>(o0, o1);
        p.y = pack2<T>(o2, o3);
        reinterpret_cast<uint2*>(reinterpret_cast<T*>(y) + row * ldy)[c] = p;
      }
    }
  }
}

__global__ __launch_bounds__(256) void add_cls_pos_kernel(const float4* __restrict__ x, const float4* __restrict__ cls,
                                                          const float4* __restrict__ pos, float4* __restrict__ out,
                                                          int B, int N, int D4) {
  const int64_t total = (int64_t)B * (N + 1) * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D4);
    const int64_t r = i / D4;
    const int t = (int)(r % (N + 1));
    const int b = (int)(r / (N + 1));
    const float4 p = pos[(int64_t)t * D4 + c];
    const float4 a = (t == 0) ? cls[c] : x[((int64_t)b * N + (t - 1)) * D4 + c];
    out[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, int64_t ld_src,
                                                       T* __restrict__ dst, int64_t ld_dst, int64_t rows, int cols, float scale, int part) {
  const int cpr = (int)(ld_dst >> 3);  // 8-element chunks per row
  const int64_t total = rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c0 = (int)(i - r * cpr) * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = (c0 + j < cols) ? src[r * ld_src + c0 + j] * scale : 0.f;
      if (part) f[j] = lo_part<T>(f[j]);
    }
    uint4 p;
    p.x = pack2<T>(f[0], f[1]);
    p.y = pack2<T>(f[2], f[3]);
    p.z = pack2<T>(f[4], f[5]);
    p.w = pack2<T>(f[6], f[7]);
    *reinterpret_cast<uint4*>(dst + r * ld_dst + c0) = p;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img, int B, int Himg, int Wimg,
                                                           int P, T* __restrict__ out, int64_t ldk, T* __restrict__ out_lo) {
  const int gh = Himg / P, gw = Wimg / P;
  const int cpr = (int)(ldk >> 3);
  const int K = 3 * P * P;
  const int64_t total = (int64_t)B * gh * gw * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int k0 = (int)(i - r * cpr) * 8;
    const int pw = (int)(r % gw);
    const int ph = (int)((r / gw) % gh);
    const int b = (int)(r / ((int64_t)gw * gh));
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float v = 0.f;
      if (k < K) {
        const int c = k / (P * P);
        const int rem = k - c * P * P;
        const int ii = rem / P, jj = rem - ii * P;
        v = img[(((int64_t)b * 3 + c) * Himg + (ph * P + ii)) * Wimg + pw * P + jj];
      }
      f[j] = v;
    }
    uint4 p;
    p.x = pack2<T>(f[0], f[1]);
    p.y = pack2<T>(f[2], f[3]);
    p.z = pack2<T>(f[4], f[5]);
    p.w = pack2<T>(f[6], f[7]);
    *reinterpret_cast<uint4*>(out + r * ldk + k0) = p;
    if (out_lo) {  // rounding residuals: img ~= hi + lo (split-precision patch embedding)
      uint4 q;
      q.x = pack2<T>(lo_part<T>(f[0]), lo_part<T>(f[1]));
      q.y = pack2<T>(lo_part<T>(f[2]), lo_part<T>(f[3]));
      q.z = pack2<T>(lo_part<T>(f[4]), lo_part<T>(f[5]));
      q.w = pack2<T>(lo_part<T>(f[6]), lo_part<T>(f[7]));
      *reinterpret_cast<uint4*>(out_lo + r * ldk + k0) = q;
    }
  }
}

inline int grid_for(int64_t total, int block = 256, int cap = 256 * 16) {
  int64_t g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ---- LayerNorm-fold chain helpers (include/asis_hip.h: asis_gemm_desc.rowstats / ln_mr) ---------------------------------------
// partial (sum, sum of squares) per 64-column group -> (mean, rstd) per row; partials combined as (count, mean, M2) triples
// L (16 or 32) lanes per row, one 64-column group each: coalesced reads, xor-butterfly sums inside the L-lane group (a fixed
// order).  Round 5: the one-thread-per-row form (83 workgroups, 128-byte strided reads) took 36 us x 65 launches per step.
template <int L>
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ st, int64_t rows, int groups, int D,
                                                                float eps, float* __restrict__ mr) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = tid / L;
  const int g = (int)(tid % L);
  const bool ok = row < rows && g < groups;
  const float2 v = ok ? reinterpret_cast<const float2*>(st)[row * groups + g] : make_float2(0.f, 0.f);
  float tot = v.x;
#pragma unroll
  for (int o = L / 2; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, L);
  const float mean = tot / (float)D;
  const float n = (float)(g + 1 < groups ? 64 : D - 64 * (groups - 1));
  const float mg = v.x / n;
  const float dm = mg - mean;
  float m2 = ok ? fmaxf(v.y - v.x * mg, 0.f) + n * dm * dm : 0.f;
#pragma unroll
  for (int o = L / 2; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o, L);
  if (g == 0 && row < rows) reinterpret_cast<float2*>(mr)[row] = make_float2(mean, 1.0f / sqrtf(m2 / (float)D + eps));
}

// fp32 rows -> hi / lo 16-bit planes + (mean, rstd): one wave per row, the statistics of layernorm_kernel
template <typename T>
__global__ __launch_bounds__(256) void split_stats_kernel(const float* __restrict__ x, int64_t ldx, T* __restrict__ hi, T* __restrict__ lo,
                                                          int64_t ld16, float* __restrict__ mr, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
  float4 v[LN_MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      v[i] = xr[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      uint2 ph, pl;
      ph.x = pack2<T>(v[i].x, v[i].y);
      ph.y = pack2<T>(v[i].z, v[i].w);
      pl.x = pack2<T>(lo_part<T>(v[i].x), lo_part<T>(v[i].y));
      pl.y = pack2<T>(lo_part<T>(v[i].z), lo_part<T>(v[i].w));
      reinterpret_cast<uint2*>(hi + row * ld16)[c] = ph;
      reinterpret_cast<uint2*>(lo + row * ld16)[c] = pl;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) reinterpret_cast<float2*>(mr)[row] = make_float2(mean, rstd);
}

extern "C" int asis_ln_stats_finalize(void* stream, const float* rowstats, int64_t rows, int groups, int D, float eps, float* mr) {
  ASIS_REQUIRE(rowstats && mr && rows >= 0 && groups >= 1 && groups <= 32 && D > 64 * (groups - 1) && D <= 64 * groups,
               "asis_ln_stats_finalize: bad arguments (rows %ld, groups %d <= 32, D %d)", (long)rows, groups, D);
  if (rows == 0) return ASIS_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (groups <= 16)
    hipLaunchKernelGGL((ln_stats_finalize_kernel<16>), dim3((unsigned)asis_cdiv(rows * 16, 256)), dim3(256), 0, s, rowstats, rows, groups, D,
                       eps, mr);
  else
    hipLaunchKernelGGL((ln_stats_finalize_kernel<32>), dim3((unsigned)asis_cdiv(rows * 32, 256)), dim3(256), 0, s, rowstats, rows, groups, D,
                       eps, mr);
  ASIS_CHECK_LAUNCH("asis_ln_stats_finalize");
  return ASIS_OK;
}

extern "C" int asis_split_stats(void* stream, int dtype, const float* x, int64_t ldx, void* hi, void* lo, int64_t ld16, float* mr,
                                int64_t rows, int D, float eps) {
  ASIS_REQUIRE(x && hi && lo && mr, "asis_split_stats: null pointer");
  ASIS_REQUIRE(D > 0 && D % 4 == 0 && D <= 256 * LN_MAXC, "asis_split_stats: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_MAXC);
  ASIS_REQUIRE(ldx % 4 == 0 && ldx >= D && ld16 % 4 == 0 && ld16 >= D, "asis_split_stats: row strides must be multiples of 4 and >= D");
  ASIS_REQUIRE(asis_aligned16(x) && (((uintptr_t)hi) & 7) == 0 && (((uintptr_t)lo) & 7) == 0, "asis_split_stats: misaligned pointers");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_split_stats: bad dtype %d", dtype);
  if (rows <= 0) return ASIS_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)asis_cdiv(rows, 4)), block(256);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((split_stats_kernel<f16>), grid, block, 0, s, x, ldx, reinterpret_cast<f16*>(hi), reinterpret_cast<f16*>(lo), ld16, mr, rows, D, eps);
  else
    hipLaunchKernelGGL((split_stats_kernel<bf16>), grid, block, 0, s, x, ldx, reinterpret_cast<bf16*>(hi), reinterpret_cast<bf16*>(lo), ld16, mr, rows, D, eps);
  ASIS_CHECK_LAUNCH("asis_split_stats");
  return ASIS_OK;
}

extern "C" int asis_layernorm(void* stream, int dtype, const float* x, int64_t ldx, const float* w, const float* b,
                              float eps, void* y, int64_t ldy, int out_f32, int64_t rows, int D) {
  ASIS_REQUIRE(x && w && b && y, "asis_layernorm: null pointer");
  ASIS_REQUIRE(D > 0 && D % 4 == 0 && D <= 256 * LN_MAXC, "asis_layernorm: D=%d must be a multiple of 4 and <= %d", D,
               256 * LN_MAXC);
  ASIS_REQUIRE(ldx % 4 == 0 && ldx >= D && ldy % 4 == 0 && ldy >= D, "asis_layernorm: row strides must be multiples of 4 and >= D");
  ASIS_REQUIRE(asis_aligned16(x) && asis_aligned16(w) && asis_aligned16(b) && (((uintptr_t)y) & 7) == 0,
               "asis_layernorm: pointers must be 16-byte aligned");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_layernorm: bad dtype %d", dtype);
  if (rows <= 0) return ASIS_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)asis_cdiv(rows, 4)), block(256);
  static const int ln_fast = [] {
    const char* e = getenv("ASIS_LN_FAST");
    return e ? atoi(e) : 2;
  }();
  if (ln_fast && !out_f32 && (D == 768 || D == 1024 || D == 1536) && rows >= 4096) {
#define ASIS_LN_FIXED(T, NCH)                                                                                         \
  do {                                                                                                                \
    if (ln_fast == 1)                                                                                                 \
      hipLaunchKernelGGL((layernorm_fixed_kernel<T, false, NCH, 1>), grid, block, 0, s, x, ldx, w, b, eps, y, ldy,   \
                         rows);                                                                                       \
    else                                                                                                              \
      hipLaunchKernelGGL((layernorm_fixed_kernel<T, false, NCH, 2>), dim3((unsigned)asis_cdiv(rows, 8)), block, 0, s, \
                         x, ldx, w, b, eps, y, ldy, rows);                                                            \
  } while (0)
    if (dtype == ASIS_F16) {
      if (D == 768) ASIS_LN_FIXED(f16, 3);
      else if (D == 1024) ASIS_LN_FIXED(f16, 4);
      else ASIS_LN_FIXED(f16, 6);
    } else {
      if (D == 768) ASIS_LN_FIXED(bf16, 3);
      else if (D == 1024) ASIS_LN_FIXED(bf16, 4);
      else ASIS_LN_FIXED(bf16, 6);
    }
#undef ASIS_LN_FIXED
    ASIS_CHECK_LAUNCH("asis_layernorm");
    return ASIS_OK;
  }
  if (out_f32)
    hipLaunchKernelGGL((layernorm_kernel<f16, true>), grid, block, 0, s, x, ldx, w, b, eps, y, ldy, rows, D);
  else if (dtype == ASIS_F16)
    hipLaunchKernelGGL((layernorm_kernel<f16, false>), grid, block, 0, s, x, ldx, w, b, eps, y, ldy, rows, D);
  else
    hipLaunchKernelGGL((layernorm_kernel<bf16, false>), grid, block, 0, s, x, ldx, w, b, eps, y, ldy, rows, D);
  ASIS_CHECK_LAUNCH("asis_layernorm");
  return ASIS_OK;
}

// LayerNorm -> the 16-bit hi plane AND the MX plane of the output (asis_common.h: (hi8, lo8) per element) in one pass: the A operand
// pair of a split-precision linear layer (config.precise_level 2 on the MX correction pass).  amax = an upper bound of |y|.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_mx_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                           const float* __restrict__ b, float eps, T* __restrict__ y, T* __restrict__ y_mx,
                                                           int64_t ldy, const float* __restrict__ amax, int64_t rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const MxScale sc = mx_scales<T>(*amax);
  const int nchunk = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
  float4 v[LN_MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      v[i] = xr[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const float4 ww = w4[c], bb = b4[c];
      const float o0 = (v[i].x - mean) * rstd * ww.x + bb.x;
      const float o1 = (v[i].y - mean) * rstd * ww.y + bb.y;
      const float o2 = (v[i].z - mean) * rstd * ww.z + bb.z;
      const float o3 = (v[i].w - mean) * rstd * ww.w + bb.w;
      uint2 p, m;
      p.x = pack2<T>(o0, o1);
      p.y = pack2<T>(o2, o3);
      m.x = mx_pack2<T>(o0, o1, sc, false);
      m.y = mx_pack2<T>(o2, o3, sc, false);
      reinterpret_cast<uint2*>(y + row * ldy)[c] = p;
      reinterpret_cast<uint2*>(y_mx + row * ldy)[c] = m;
    }
  }
}

extern "C" int asis_layernorm_mx(void* stream, int dtype, const float* x, int64_t ldx, const float* w, const float* b, float eps, void* y,
                                 void* y_mx, int64_t ldy, const float* amax, int64_t rows, int D) {
  ASIS_REQUIRE(x && w && b && y && y_mx && amax, "asis_layernorm_mx: null pointer");
  ASIS_REQUIRE(D > 0 && D % 4 == 0 && D <= 256 * LN_MAXC && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= D && ldy >= D && asis_aligned16(x) &&
                   asis_aligned16(w) && asis_aligned16(b) && (reinterpret_cast<uintptr_t>(y) & 7) == 0 && (reinterpret_cast<uintptr_t>(y_mx) & 7) == 0,
               "asis_layernorm_mx: D=%d must be a multiple of 4 and <= %d, leading dimensions multiples of 4, pointers aligned", D, 256 * LN_MAXC);
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_layernorm_mx: bad dtype %d", dtype);
  if (rows == 0) return ASIS_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)asis_cdiv(rows, 4)), block(256);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((layernorm_mx_kernel<f16>), grid, block, 0, s, x, ldx, w, b, eps, reinterpret_cast<f16*>(y), reinterpret_cast<f16*>(y_mx), ldy, amax, rows, D);
  else
    hipLaunchKernelGGL((layernorm_mx_kernel<bf16>), grid, block, 0, s, x, ldx, w, b, eps, reinterpret_cast<bf16*>(y), reinterpret_cast<bf16*>(y_mx), ldy, amax, rows, D);
  ASIS_CHECK_LAUNCH("asis_layernorm_mx");
  return ASIS_OK;
}

extern "C" int asis_add_cls_pos(void* stream, const float* x, const float* cls, const float* pos, float* out, int B,
                                int N, int D) {
  ASIS_REQUIRE(x && cls && pos && out, "asis_add_cls_pos: null pointer");
  ASIS_REQUIRE(D % 4 == 0 && B > 0 && N > 0, "asis_add_cls_pos: bad shape B=%d N=%d D=%d", B, N, D);
  ASIS_REQUIRE(asis_aligned16(x) && asis_aligned16(cls) && asis_aligned16(pos) && asis_aligned16(out),
               "asis_add_cls_pos: pointers must be 16-byte aligned");
  const int64_t total = (int64_t)B * (N + 1) * (D / 4);
  hipLaunchKernelGGL(add_cls_pos_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(cls),
                     reinterpret_cast<const float4*>(pos), reinterpret_cast<float4*>(out), B, N, D / 4);
  ASIS_CHECK_LAUNCH("asis_add_cls_pos");
  return ASIS_OK;
}

extern "C" int asis_cast_pad(void* stream, int dtype, const float* src, int64_t ld_src, void* dst, int64_t ld_dst,
                             int64_t rows, int cols, float scale, int part) {
  ASIS_REQUIRE(src && dst, "asis_cast_pad: null pointer");
  ASIS_REQUIRE(ld_dst % 8 == 0 && ld_dst >= cols && ld_src >= cols, "asis_cast_pad: bad leading dims");
  ASIS_REQUIRE(asis_aligned16(dst), "asis_cast_pad: dst must be 16-byte aligned");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_cast_pad: bad dtype %d", dtype);
  if (rows <= 0) return ASIS_OK;
  const int64_t total = rows * (ld_dst / 8);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((cast_pad_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, src, ld_src,
                       reinterpret_cast<f16*>(dst), ld_dst, rows, cols, scale, part);
  else
    hipLaunchKernelGGL((cast_pad_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, src, ld_src,
                       reinterpret_cast<bf16*>(dst), ld_dst, rows, cols, scale, part);
  ASIS_CHECK_LAUNCH("asis_cast_pad");
  return ASIS_OK;
}

extern "C" int asis_im2col_patch_split(void* stream, int dtype, const float* img, int B, int Himg, int Wimg, int P, void* out,
                                       void* out_lo, int64_t ldk);
extern "C" int asis_im2col_patch(void* stream, int dtype, const float* img, int B, int Himg, int Wimg, int P, void* out,
                                 int64_t ldk) {
  return asis_im2col_patch_split(stream, dtype, img, B, Himg, Wimg, P, out, nullptr, ldk);
}

extern "C" int asis_im2col_patch_split(void* stream, int dtype, const float* img, int B, int Himg, int Wimg, int P, void* out,
                                       void* out_lo, int64_t ldk) {
  ASIS_REQUIRE(img && out, "asis_im2col_patch: null pointer");
  ASIS_REQUIRE(!out_lo || asis_aligned16(out_lo), "asis_im2col_patch: out_lo must be 16-byte aligned");
  ASIS_REQUIRE(P > 0 && Himg % P == 0 && Wimg % P == 0,
               "Input image size %dx%d is not a multiple of patch size %d", Himg, Wimg, P);
  ASIS_REQUIRE(ldk % 8 == 0 && ldk >= 3 * P * P, "asis_im2col_patch: ldk=%ld must be a multiple of 8 and >= 3*P*P", (long)ldk);
  ASIS_REQUIRE(asis_aligned16(out), "asis_im2col_patch: out must be 16-byte aligned");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_im2col_patch: bad dtype %d", dtype);
  const int64_t total = (int64_t)B * (Himg / P) * (Wimg / P) * (ldk / 8);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((im2col_patch_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, img, B, Himg, Wimg, P,
                       reinterpret_cast<f16*>(out), ldk, reinterpret_cast<f16*>(out_lo));
  else
    hipLaunchKernelGGL((im2col_patch_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, img, B, Himg, Wimg, P,
                       reinterpret_cast<bf16*>(out), ldk, reinterpret_cast<bf16*>(out_lo));
  ASIS_CHECK_LAUNCH("asis_im2col_patch");
  return ASIS_OK;
}
