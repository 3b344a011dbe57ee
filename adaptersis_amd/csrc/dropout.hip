// Dropout for the MaskTransformer decode head (`backbones/masktrans_block.py:11-89`: nn.Dropout(0.1) on the attention
// probabilities, on the projection output, behind GELU and behind fc2; the script builds the head with dropout=0.1,
// `eval/eval_dinov2_masktrans.py:136-139`).
//
// Masks are COUNTER-BASED: keep(seed, site, i) = Philox4x32-10(key = seed, counter = (i / 4, site))[i % 4] >= p * 2^32 —
// a pure function of (seed, site, element index), so the forward, every backward kernel and the mask export (the oracle
// replays exactly these masks: tests/test_gpu_masktrans.py) regenerate it instead of storing it, whatever their thread
// layout.  `site` numbers the dropout layer (block * 4 + {attention probabilities, proj, GELU, fc2}), `seed` changes every step.
// One Philox call serves four consecutive elements.
//
// The attention-probability dropout runs on an UNFUSED attention (scores and probabilities materialised per head: 150 MB per
// image and layer at N = 1766 — this head is two layers on a 288 GB part) built from batched asis_gemm launches and the two
// row kernels below; the fused flash kernels serve every path without dropout.
#include "asis_common.h"

namespace {

struct philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return philox4{c0, c1, c2, c3};
}

// the four keep flags of elements 4 * blk .. 4 * blk + 3 of dropout layer `site`
__device__ __forceinline__ void keep4(uint64_t seed, uint32_t site, uint64_t blk, uint32_t thr, bool k[4]) {
  const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  k[0] = r.x >= thr; k[1] = r.y >= thr; k[2] = r.z >= thr; k[3] = r.w >= thr;
}

__host__ uint32_t threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// out = (res ? res : 0) + (alpha * x + bias[col]) * keep / (1 - p), fp32 [rows, ncols] contiguous, ncols % 4 == 0
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float4* __restrict__ x, const float4* __restrict__ res,
                                                          float4* __restrict__ out, int64_t n4, uint64_t seed, uint32_t site,
                                                          uint32_t thr, float inv_keep, float alpha, const float4* __restrict__ bias,
                                                          int ncols4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    bool k[4];
    keep4(seed, site, (uint64_t)i, thr, k);
    float4 v = x[i];
    const float4 bb = bias ? bias[i % ncols4] : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = alpha * v.x + bb.x; v.y = alpha * v.y + bb.y; v.z = alpha * v.z + bb.z; v.w = alpha * v.w + bb.w;
    float4 r = res ? res[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    r.x += k[0] ? v.x * inv_keep : 0.f;
    r.y += k[1] ? v.y * inv_keep : 0.f;
    r.z += k[2] ? v.z * inv_keep : 0.f;
    r.w += k[3] ? v.w * inv_keep : 0.f;
    out[i] = r;
  }
}

// 16-bit in place: x = x * keep (/ (1 - p) when scale != 0: a gradient; hi / lo operand halves are only zeroed, their
// 1 / (1 - p) goes into the consuming GEMM's fp32 epilogue so that the hi + lo pair stays exact)
template <typename T>
__global__ __launch_bounds__(256) void dropout_t16_kernel(T* __restrict__ x, T* __restrict__ x_lo, int64_t n4, uint64_t seed,
                                                          uint32_t site, uint32_t thr, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    bool k[4];
    keep4(seed, site, (uint64_t)i, thr, k);
    uint2 w = reinterpret_cast<uint2*>(x)[i];
    float a, b, c, d;
    unpack2<T>(w.x, a, b);
    unpack2<T>(w.y, c, d);
    a = k[0] ? a * scale : 0.f; b = k[1] ? b * scale : 0.f; c = k[2] ? c * scale : 0.f; d = k[3] ? d * scale : 0.f;
    w.x = pack2<T>(a, b);
    w.y = pack2<T>(c, d);
    reinterpret_cast<uint2*>(x)[i] = w;
    if (x_lo) {
      uint2 l = reinterpret_cast<uint2*>(x_lo)[i];
      if (!k[0]) l.x &= 0xffff0000u;
      if (!k[1]) l.x &= 0x0000ffffu;
      if (!k[2]) l.y &= 0xffff0000u;
      if (!k[3]) l.y &= 0x0000ffffu;
      reinterpret_cast<uint2*>(x_lo)[i] = l;
    }
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out, int64_t n4, uint64_t seed, uint32_t site,
                                                           uint32_t thr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    bool k[4];
    keep4(seed, site, (uint64_t)i, thr, k);
    reinterpret_cast<uchar4*>(out)[i] = make_uchar4(k[0], k[1], k[2], k[3]);
  }
}

// One wave per row of the scores S fp32 [rows, ld] (columns >= N are padding): P = softmax(scale * S) over the first N columns;
// p16 = P, pd16 = P * keep / (1 - p) (padding written as 0).  Mask element index = row * ld + column.
template <typename T>
__global__ __launch_bounds__(256) void softmax_dropout_fwd_kernel(const float* __restrict__ S, T* __restrict__ p16,
                                                                  T* __restrict__ pd16, int64_t rows, int N, int ld, float scale_log2e,
                                                                  uint64_t seed, uint32_t site, uint32_t thr, float inv_keep) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = S + row * ld;
  float m = -3.0e38f;
  for (int c = lane * 4; c < N; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(s + c);
    m = fmaxf(m, v.x);
    if (c + 1 < N) m = fmaxf(m, v.y);
    if (c + 2 < N) m = fmaxf(m, v.z);
    if (c + 3 < N) m = fmaxf(m, v.w);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int c = lane * 4; c < N; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(s + c);
    l += __builtin_amdgcn_exp2f((v.x - m) * scale_log2e);
    if (c + 1 < N) l += __builtin_amdgcn_exp2f((v.y - m) * scale_log2e);
    if (c + 2 < N) l += __builtin_amdgcn_exp2f((v.z - m) * scale_log2e);
    if (c + 3 < N) l += __builtin_amdgcn_exp2f((v.w - m) * scale_log2e);
  }
  l = wave_sum(l);
  const float inv = 1.0f / l;
  for (int c = lane * 4; c < ld; c += 256) {
    float pr[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
      const float4 v = *reinterpret_cast<const float4*>(s + c);
      pr[0] = __builtin_amdgcn_exp2f((v.x - m) * scale_log2e) * inv;
      if (c + 1 < N) pr[1] = __builtin_amdgcn_exp2f((v.y - m) * scale_log2e) * inv;
      if (c + 2 < N) pr[2] = __builtin_amdgcn_exp2f((v.z - m) * scale_log2e) * inv;
      if (c + 3 < N) pr[3] = __builtin_amdgcn_exp2f((v.w - m) * scale_log2e) * inv;
    }
    bool k[4];
    keep4(seed, site, (uint64_t)(row * ld + c) >> 2, thr, k);
    uint2 w, wd;
    w.x = pack2<T>(pr[0], pr[1]);
    w.y = pack2<T>(pr[2], pr[3]);
    wd.x = pack2<T>(k[0] ? pr[0] * inv_keep : 0.f, k[1] ? pr[1] * inv_keep : 0.f);
    wd.y = pack2<T>(k[2] ? pr[2] * inv_keep : 0.f, k[3] ? pr[3] * inv_keep : 0.f);
    *reinterpret_cast<uint2*>(p16 + row * ld + c) = w;
    *reinterpret_cast<uint2*>(pd16 + row * ld + c) = wd;
  }
}

// dS = scale * P * (keep / (1 - p) * dPd - D),  D = sum_k Pd_k dPd_k (= dO . O): one wave per row; dS 16-bit, padding 0
template <typename T>
__global__ __launch_bounds__(256) void softmax_dropout_bwd_kernel(const T* __restrict__ p16, const T* __restrict__ pd16,
                                                                  const float* __restrict__ dPd, T* __restrict__ ds16, int64_t rows,
                                                                  int N, int ld, float scale, uint64_t seed, uint32_t site,
                                                                  uint32_t thr, float inv_keep) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* p = p16 + row * ld;
  const T* pd = pd16 + row * ld;
  const float* g = dPd + row * ld;
  float dsum = 0.f;
  for (int c = lane * 4; c < N; c += 256) {
    const uint2 w = *reinterpret_cast<const uint2*>(pd + c);
    const float4 v = *reinterpret_cast<const float4*>(g + c);
    float a, b, cc, d;
    unpack2<T>(w.x, a, b);
    unpack2<T>(w.y, cc, d);
    dsum += a * v.x;                                     // dPd's padding columns were never written: guard them
    if (c + 1 < N) dsum += b * v.y;
    if (c + 2 < N) dsum += cc * v.z;
    if (c + 3 < N) dsum += d * v.w;
  }
  dsum = wave_sum(dsum);
  for (int c = lane * 4; c < ld; c += 256) {
    const uint2 w = *reinterpret_cast<const uint2*>(p + c);
    const float4 v = *reinterpret_cast<const float4*>(g + c);
    float pa, pb, pc, pdv;
    unpack2<T>(w.x, pa, pb);
    unpack2<T>(w.y, pc, pdv);
    bool k[4];
    keep4(seed, site, (uint64_t)(row * ld + c) >> 2, thr, k);
    const float r0 = c < N ? scale * pa * ((k[0] ? v.x * inv_keep : 0.f) - dsum) : 0.f;
    const float r1 = c + 1 < N ? scale * pb * ((k[1] ? v.y * inv_keep : 0.f) - dsum) : 0.f;
    const float r2 = c + 2 < N ? scale * pc * ((k[2] ? v.z * inv_keep : 0.f) - dsum) : 0.f;
    const float r3 = c + 3 < N ? scale * pdv * ((k[3] ? v.w * inv_keep : 0.f) - dsum) : 0.f;
    uint2 o;
    o.x = pack2<T>(r0, r1);
    o.y = pack2<T>(r2, r3);
    *reinterpret_cast<uint2*>(ds16 + row * ld + c) = o;
  }
}

int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

#define DROP_ARGS_OK(name)                                                                             \
  ASIS_REQUIRE(p >= 0.f && p < 1.f, name ": dropout probability must be in [0, 1)");                   \
  ASIS_REQUIRE(n > 0 && n % 4 == 0, name ": element count must be a positive multiple of 4")

extern "C" int asis_dropout_f32(void* stream, const float* x, const float* res, float* out, int64_t n, uint64_t seed, int site,
                                float p, float alpha, const float* bias_n, int ncols) {
  ASIS_REQUIRE(x && out, "asis_dropout_f32: null pointer");
  DROP_ARGS_OK("asis_dropout_f32");
  ASIS_REQUIRE(asis_aligned16(x) && asis_aligned16(out) && (!res || asis_aligned16(res)), "asis_dropout_f32: 16-byte alignment");
  ASIS_REQUIRE(!bias_n || (ncols > 0 && ncols % 4 == 0 && n % ncols == 0 && asis_aligned16(bias_n)), "asis_dropout_f32: bias needs ncols % 4 == 0 dividing n");
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(res), reinterpret_cast<float4*>(out), n / 4,
                     seed, (uint32_t)site, threshold(p), 1.0f / (1.0f - p), alpha, reinterpret_cast<const float4*>(bias_n), bias_n ? ncols / 4 : 1);
  ASIS_CHECK_LAUNCH("asis_dropout_f32");
  return ASIS_OK;
}

extern "C" int asis_dropout_t16(void* stream, int dtype, void* x, void* x_lo, int64_t n, uint64_t seed, int site, float p, int rescale) {
  ASIS_REQUIRE(x, "asis_dropout_t16: null pointer");
  DROP_ARGS_OK("asis_dropout_t16");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_dropout_t16: bad dtype %d", dtype);
  ASIS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (!x_lo || (reinterpret_cast<uintptr_t>(x_lo) & 7) == 0), "asis_dropout_t16: 8-byte alignment");
  ASIS_REQUIRE(!(x_lo && rescale), "asis_dropout_t16: a hi / lo operand pair is only zeroed (its 1 / (1 - p) belongs to the consuming GEMM)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float scale = rescale ? 1.0f / (1.0f - p) : 1.0f;
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((dropout_t16_kernel<f16>), dim3(grid_for(n / 4)), dim3(256), 0, s, reinterpret_cast<f16*>(x), reinterpret_cast<f16*>(x_lo),
                       n / 4, seed, (uint32_t)site, threshold(p), scale);
  else
    hipLaunchKernelGGL((dropout_t16_kernel<bf16>), dim3(grid_for(n / 4)), dim3(256), 0, s, reinterpret_cast<bf16*>(x), reinterpret_cast<bf16*>(x_lo),
                       n / 4, seed, (uint32_t)site, threshold(p), scale);
  ASIS_CHECK_LAUNCH("asis_dropout_t16");
  return ASIS_OK;
}

extern "C" int asis_dropout_mask(void* stream, uint8_t* out, int64_t n, uint64_t seed, int site, float p) {
  ASIS_REQUIRE(out, "asis_dropout_mask: null pointer");
  DROP_ARGS_OK("asis_dropout_mask");
  ASIS_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3) == 0, "asis_dropout_mask: 4-byte alignment");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out, n / 4, seed,
                     (uint32_t)site, threshold(p));
  ASIS_CHECK_LAUNCH("asis_dropout_mask");
  return ASIS_OK;
}

extern "C" int asis_softmax_dropout_fwd(void* stream, int dtype, const float* S, void* p16, void* pd16, int64_t rows, int N, int ld,
                                        float scale, uint64_t seed, int site, float p) {
  ASIS_REQUIRE(S && p16 && pd16, "asis_softmax_dropout_fwd: null pointer");
  ASIS_REQUIRE(rows > 0 && N > 0 && ld >= N && ld % 4 == 0, "asis_softmax_dropout_fwd: bad shape (ld a multiple of 4, >= N)");
  ASIS_REQUIRE(p >= 0.f && p < 1.f, "asis_softmax_dropout_fwd: dropout probability must be in [0, 1)");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_softmax_dropout_fwd: bad dtype %d", dtype);
  ASIS_REQUIRE(asis_aligned16(S) && (reinterpret_cast<uintptr_t>(p16) & 7) == 0 && (reinterpret_cast<uintptr_t>(pd16) & 7) == 0,
               "asis_softmax_dropout_fwd: alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((rows + 3) / 4));
  const float sl = scale * 1.4426950408889634f, ik = 1.0f / (1.0f - p);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((softmax_dropout_fwd_kernel<f16>), grid, dim3(256), 0, s, S, reinterpret_cast<f16*>(p16), reinterpret_cast<f16*>(pd16),
                       rows, N, ld, sl, seed, (uint32_t)site, threshold(p), ik);
  else
    hipLaunchKernelGGL((softmax_dropout_fwd_kernel<bf16>), grid, dim3(256), 0, s, S, reinterpret_cast<bf16*>(p16), reinterpret_cast<bf16*>(pd16),
                       rows, N, ld, sl, seed, (uint32_t)site, threshold(p), ik);
  ASIS_CHECK_LAUNCH("asis_softmax_dropout_fwd");
  return ASIS_OK;
}

extern "C" int asis_softmax_dropout_bwd(void* stream, int dtype, const void* p16, const void* pd16, const float* dPd, void* ds16,
                                        int64_t rows, int N, int ld, float scale, uint64_t seed, int site, float p) {
  ASIS_REQUIRE(p16 && pd16 && dPd && ds16, "asis_softmax_dropout_bwd: null pointer");
  ASIS_REQUIRE(rows > 0 && N > 0 && ld >= N && ld % 4 == 0, "asis_softmax_dropout_bwd: bad shape (ld a multiple of 4, >= N)");
  ASIS_REQUIRE(p >= 0.f && p < 1.f, "asis_softmax_dropout_bwd: dropout probability must be in [0, 1)");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_softmax_dropout_bwd: bad dtype %d", dtype);
  ASIS_REQUIRE(asis_aligned16(dPd) && (reinterpret_cast<uintptr_t>(p16) & 7) == 0 && (reinterpret_cast<uintptr_t>(pd16) & 7) == 0 &&
               (reinterpret_cast<uintptr_t>(ds16) & 7) == 0, "asis_softmax_dropout_bwd: alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((rows + 3) / 4));
  const float ik = 1.0f / (1.0f - p);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((softmax_dropout_bwd_kernel<f16>), grid, dim3(256), 0, s, reinterpret_cast<const f16*>(p16), reinterpret_cast<const f16*>(pd16),
                       dPd, reinterpret_cast<f16*>(ds16), rows, N, ld, scale, seed, (uint32_t)site, threshold(p), ik);
  else
    hipLaunchKernelGGL((softmax_dropout_bwd_kernel<bf16>), grid, dim3(256), 0, s, reinterpret_cast<const bf16*>(p16), reinterpret_cast<const bf16*>(pd16),
                       dPd, reinterpret_cast<bf16*>(ds16), rows, N, ld, scale, seed, (uint32_t)site, threshold(p), ik);
  ASIS_CHECK_LAUNCH("asis_softmax_dropout_bwd");
  return ASIS_OK;
}
