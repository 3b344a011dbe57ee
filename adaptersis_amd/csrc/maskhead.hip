// Tail of the MaskTransformer decode head (Segmenter-style, reference eval/eval_dinov2_masktrans.py:452-462):
//     patches = x[:, :-C] @ proj_patch ; cls = x[:, -C:] @ proj_classes          (two GEMMs, asis_gemm)
//     patches /= ||patches|| ; cls /= ||cls||                                      (L2 over the feature axis)
//     masks = patches @ cls^T                                                      (B, N, C): cosines
//     masks = LayerNorm_C(masks)                                                   (mask_norm, affine, over the C classes)
// Rows live in the stacked token matrix of the head: batch b owns rows b*RB .. b*RB + N - 1 (patches) and
// b*RB + N .. b*RB + N + C - 1 (class tokens), RB = N + C.  C <= 16, one wave per row, D a multiple of 4.
// All of it is HBM-bound row streaming (the projected patch rows are read once per pass).
#include "asis_common.h"

namespace {

constexpr int MAXC = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// class rows: chat[b, k, :] = x[row] / ||x[row]||, inv_c[b*C + k] = 1 / ||x[row]||
__global__ __launch_bounds__(256) void cls_l2norm_kernel(const float* __restrict__ x, float* __restrict__ chat,
                                                         float* __restrict__ inv_c, int B, int N, int C, int D) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wid >= B * C) return;
  const int b = wid / C, k = wid - b * C;
  const float* r = x + ((int64_t)b * (N + C) + N + k) * D;
  float ss = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 v = *reinterpret_cast<const float4*>(r + d);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss);
  if (lane == 0) inv_c[wid] = inv;
  float* o = chat + (int64_t)wid * D;
  for (int d = lane * 4; d < D; d += 256) {
    float4 v = *reinterpret_cast<const float4*>(r + d);
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    *reinterpret_cast<float4*>(o + d) = v;
  }
}

// dx[row] = inv (dchat - chat (chat . dchat)) written into the class rows of dX (stacked layout)
__global__ __launch_bounds__(256) void cls_l2norm_bwd_kernel(const float* __restrict__ dchat, const float* __restrict__ chat,
                                                             const float* __restrict__ inv_c, float* __restrict__ dx, int B,
                                                             int N, int C, int D) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wid >= B * C) return;
  const int b = wid / C, k = wid - b * C;
  const float* dy = dchat + (int64_t)wid * D;
  const float* y = chat + (int64_t)wid * D;
  float dot = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 a = *reinterpret_cast<const float4*>(dy + d), c = *reinterpret_cast<const float4*>(y + d);
    dot += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
  dot = wave_sum(dot);
  const float inv = inv_c[wid];
  float* o = dx + ((int64_t)b * (N + C) + N + k) * D;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 a = *reinterpret_cast<const float4*>(dy + d), c = *reinterpret_cast<const float4*>(y + d);
    float4 v;
    v.x = inv * (a.x - c.x * dot); v.y = inv * (a.y - c.y * dot); v.z = inv * (a.z - c.z * dot); v.w = inv * (a.w - c.w * dot);
    *reinterpret_cast<float4*>(o + d) = v;
  }
}

// one wave per patch row: cos_k = (p . chat_k) / ||p||, logits = LayerNorm over k
template <int CC>
__global__ __launch_bounds__(256) void mask_logits_fwd_kernel(const float* __restrict__ P, const float* __restrict__ chat,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float* __restrict__ logits, float* __restrict__ cosm,
                                                              float* __restrict__ inv_p, int B, int N, int C, int D) {
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wid >= (int64_t)B * N) return;
  const int b = (int)(wid / N), n = (int)(wid - (int64_t)b * N);
  const float* p = P + ((int64_t)b * (N + C) + n) * D;
  const float* ch = chat + (int64_t)b * C * D;
  float ss = 0.f, dot[CC];
#pragma unroll
  for (int k = 0; k < CC; ++k) dot[k] = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 v = *reinterpret_cast<const float4*>(p + d);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int k = 0; k < CC; ++k)
      if (k < C) {
        const float4 c = *reinterpret_cast<const float4*>(ch + (int64_t)k * D + d);
        dot[k] += v.x * c.x + v.y * c.y + v.z * c.z + v.w * c.w;
      }
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss);
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < CC; ++k) {
    dot[k] = wave_sum(dot[k]) * inv;
    if (k < C) mean += dot[k];
  }
  mean /= (float)C;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < CC; ++k)
    if (k < C) var += (dot[k] - mean) * (dot[k] - mean);
  const float rstd = 1.0f / sqrtf(var / (float)C + eps);
  if (lane == 0) {
    inv_p[wid] = inv;
#pragma unroll
    for (int k = 0; k < CC; ++k)
      if (k < C) {
        cosm[wid * C + k] = dot[k];
        logits[wid * C + k] = (dot[k] - mean) * rstd * gamma[k] + beta[k];
      }
  }
}

// one wave per patch row: LayerNorm_C backward -> dcos; dP = inv_p (sum_k dcos_k chat_k - phat sum_k dcos_k cos_k);
// per-workgroup partial sums of (dy * xhat | dy) for mask_norm's weight / bias gradients
template <int CC>
__global__ __launch_bounds__(256) void mask_logits_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ cosm,
                                                              const float* __restrict__ inv_p, const float* __restrict__ P,
                                                              const float* __restrict__ chat, const float* __restrict__ gamma,
                                                              float eps, float* __restrict__ dP, float* __restrict__ dcos,
                                                              float* __restrict__ part, int B, int N, int C, int D) {
  __shared__ float red[4][2 * MAXC];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wv;
  float gsum[CC], bsum[CC];
#pragma unroll
  for (int k = 0; k < CC; ++k) gsum[k] = bsum[k] = 0.f;
  if (wid < (int64_t)B * N) {
    const int b = (int)(wid / N), n = (int)(wid - (int64_t)b * N);
    float m[CC], dy[CC], dm[CC];
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < CC; ++k) {
      m[k] = k < C ? cosm[wid * C + k] : 0.f;
      dy[k] = k < C ? dlogits[wid * C + k] : 0.f;
      mean += m[k];
    }
    mean /= (float)C;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < CC; ++k)
      if (k < C) var += (m[k] - mean) * (m[k] - mean);
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;   // sum of dxhat, sum of dxhat * xhat
#pragma unroll
    for (int k = 0; k < CC; ++k)
      if (k < C) {
        const float xh = (m[k] - mean) * rstd, dxh = dy[k] * gamma[k];
        s1 += dxh;
        s2 += dxh * xh;
        gsum[k] = dy[k] * xh;
        bsum[k] = dy[k];
      }
    float sdm = 0.f;            // sum_k dcos_k cos_k
#pragma unroll
    for (int k = 0; k < CC; ++k) {
      dm[k] = 0.f;
      if (k < C) {
        const float xh = (m[k] - mean) * rstd, dxh = dy[k] * gamma[k];
        dm[k] = rstd * (dxh - (s1 + xh * s2) / (float)C);
        sdm += dm[k] * m[k];
        if (lane == 0) dcos[wid * C + k] = dm[k];
      }
    }
    const float inv = inv_p[wid];
    const float* p = P + ((int64_t)b * (N + C) + n) * D;
    const float* ch = chat + (int64_t)b * C * D;
    float* o = dP + ((int64_t)b * (N + C) + n) * D;
    for (int d = lane * 4; d < D; d += 256) {
      const float4 v = *reinterpret_cast<const float4*>(p + d);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < CC; ++k)
        if (k < C) {
          const float4 c = *reinterpret_cast<const float4*>(ch + (int64_t)k * D + d);
          acc.x += dm[k] * c.x; acc.y += dm[k] * c.y; acc.z += dm[k] * c.z; acc.w += dm[k] * c.w;
        }
      const float q = inv * sdm;   // phat = p * inv
      float4 r;
      r.x = inv * (acc.x - v.x * q); r.y = inv * (acc.y - v.y * q); r.z = inv * (acc.z - v.z * q); r.w = inv * (acc.w - v.w * q);
      *reinterpret_cast<float4*>(o + d) = r;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < CC; ++k) {
      red[wv][k] = gsum[k];
      red[wv][MAXC + k] = bsum[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * MAXC) {
    const int k = threadIdx.x & (MAXC - 1), half = threadIdx.x / MAXC;
    if (k < C) {
      const int idx = half * MAXC + k;
      part[(int64_t)blockIdx.x * 2 * C + half * C + k] = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
    }
  }
}

// dchat[b, k, d] = sum_n dcos[b, n, k] * P[b, n, d] * inv_p[b, n]: one workgroup per (batch, 256-column chunk, row slice),
// slices summed by atomics into the zero-initialised output (C * D floats per batch: no contention to speak of)
template <int CC>
__global__ __launch_bounds__(256) void mask_dchat_kernel(const float* __restrict__ dcos, const float* __restrict__ P,
                                                         const float* __restrict__ inv_p, float* __restrict__ dchat, int B, int N,
                                                         int C, int D, int rows_per_slice) {
  const int b = blockIdx.z, d = blockIdx.x * 256 + threadIdx.x;
  const int n0 = blockIdx.y * rows_per_slice, n1 = min(N, n0 + rows_per_slice);
  if (d >= D) return;
  float acc[CC];
#pragma unroll
  for (int k = 0; k < CC; ++k) acc[k] = 0.f;
  for (int n = n0; n < n1; ++n) {
    const int64_t row = (int64_t)b * N + n;
    const float v = P[((int64_t)b * (N + C) + n) * D + d] * inv_p[row];
#pragma unroll
    for (int k = 0; k < CC; ++k)
      if (k < C) acc[k] += dcos[row * C + k] * v;
  }
#pragma unroll
  for (int k = 0; k < CC; ++k)
    if (k < C) atomicAdd(dchat + ((int64_t)b * C + k) * D + d, acc[k]);
}

}  // namespace

#define MH_SHAPE(name)                                                                                                       \
  ASIS_REQUIRE(B > 0 && N > 0 && C > 0 && C <= MAXC && D > 0 && D % 4 == 0, name ": bad shape B=%d N=%d C=%d (<= 16) D=%d (multiple of 4)", \
               B, N, C, D)

extern "C" int asis_cls_l2norm(void* stream, const float* x, float* chat, float* inv_c, int B, int N, int C, int D) {
  ASIS_REQUIRE(x && chat && inv_c, "asis_cls_l2norm: null pointer");
  MH_SHAPE("asis_cls_l2norm");
  hipLaunchKernelGGL(cls_l2norm_kernel, dim3((B * C + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, chat, inv_c,
                     B, N, C, D);
  ASIS_CHECK_LAUNCH("asis_cls_l2norm");
  return ASIS_OK;
}

extern "C" int asis_cls_l2norm_bwd(void* stream, const float* dchat, const float* chat, const float* inv_c, float* dx, int B, int N,
                                   int C, int D) {
  ASIS_REQUIRE(dchat && chat && inv_c && dx, "asis_cls_l2norm_bwd: null pointer");
  MH_SHAPE("asis_cls_l2norm_bwd");
  hipLaunchKernelGGL(cls_l2norm_bwd_kernel, dim3((B * C + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dchat, chat,
                     inv_c, dx, B, N, C, D);
  ASIS_CHECK_LAUNCH("asis_cls_l2norm_bwd");
  return ASIS_OK;
}

extern "C" int asis_mask_logits_fwd(void* stream, const float* P, const float* chat, const float* gamma, const float* beta, float eps,
                                    float* logits, float* cosm, float* inv_p, int B, int N, int C, int D) {
  ASIS_REQUIRE(P && chat && gamma && beta && logits && cosm && inv_p, "asis_mask_logits_fwd: null pointer");
  MH_SHAPE("asis_mask_logits_fwd");
  const int64_t rows = (int64_t)B * N;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (C <= 2) hipLaunchKernelGGL((mask_logits_fwd_kernel<2>), grid, block, 0, s, P, chat, gamma, beta, eps, logits, cosm, inv_p, B, N, C, D);
  else if (C <= 4) hipLaunchKernelGGL((mask_logits_fwd_kernel<4>), grid, block, 0, s, P, chat, gamma, beta, eps, logits, cosm, inv_p, B, N, C, D);
  else if (C <= 8) hipLaunchKernelGGL((mask_logits_fwd_kernel<8>), grid, block, 0, s, P, chat, gamma, beta, eps, logits, cosm, inv_p, B, N, C, D);
  else hipLaunchKernelGGL((mask_logits_fwd_kernel<16>), grid, block, 0, s, P, chat, gamma, beta, eps, logits, cosm, inv_p, B, N, C, D);
  ASIS_CHECK_LAUNCH("asis_mask_logits_fwd");
  return ASIS_OK;
}

extern "C" int asis_mask_logits_nblk(int B, int N) { return (int)(((int64_t)B * N + 3) / 4); }

extern "C" int asis_mask_logits_bwd(void* stream, const float* dlogits, const float* cosm, const float* inv_p, const float* P,
                                    const float* chat, const float* gamma, float eps, float* dP, float* dcos, float* part, int B,
                                    int N, int C, int D) {
  ASIS_REQUIRE(dlogits && cosm && inv_p && P && chat && gamma && dP && dcos && part, "asis_mask_logits_bwd: null pointer");
  MH_SHAPE("asis_mask_logits_bwd");
  dim3 grid((unsigned)asis_mask_logits_nblk(B, N)), block(256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (C <= 2) hipLaunchKernelGGL((mask_logits_bwd_kernel<2>), grid, block, 0, s, dlogits, cosm, inv_p, P, chat, gamma, eps, dP, dcos, part, B, N, C, D);
  else if (C <= 4) hipLaunchKernelGGL((mask_logits_bwd_kernel<4>), grid, block, 0, s, dlogits, cosm, inv_p, P, chat, gamma, eps, dP, dcos, part, B, N, C, D);
  else if (C <= 8) hipLaunchKernelGGL((mask_logits_bwd_kernel<8>), grid, block, 0, s, dlogits, cosm, inv_p, P, chat, gamma, eps, dP, dcos, part, B, N, C, D);
  else hipLaunchKernelGGL((mask_logits_bwd_kernel<16>), grid, block, 0, s, dlogits, cosm, inv_p, P, chat, gamma, eps, dP, dcos, part, B, N, C, D);
  ASIS_CHECK_LAUNCH("asis_mask_logits_bwd");
  return ASIS_OK;
}

extern "C" int asis_mask_dchat(void* stream, const float* dcos, const float* P, const float* inv_p, float* dchat, int B, int N, int C,
                               int D) {
  ASIS_REQUIRE(dcos && P && inv_p && dchat, "asis_mask_dchat: null pointer");
  MH_SHAPE("asis_mask_dchat");
  ASIS_REQUIRE(B <= 65535, "asis_mask_dchat: B too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(dchat, 0, sizeof(float) * (size_t)B * C * D, s) != hipSuccess) ASIS_FAIL(ASIS_ELAUNCH, "asis_mask_dchat: memset failed");
  const int slices = N >= 2048 ? 16 : (N >= 256 ? 8 : 1);
  const int rps = (N + slices - 1) / slices;
  dim3 grid((D + 255) / 256, slices, B), block(256);
  if (C <= 2) hipLaunchKernelGGL((mask_dchat_kernel<2>), grid, block, 0, s, dcos, P, inv_p, dchat, B, N, C, D, rps);
  else if (C <= 4) hipLaunchKernelGGL((mask_dchat_kernel<4>), grid, block, 0, s, dcos, P, inv_p, dchat, B, N, C, D, rps);
  else if (C <= 8) hipLaunchKernelGGL((mask_dchat_kernel<8>), grid, block, 0, s, dcos, P, inv_p, dchat, B, N, C, D, rps);
  else hipLaunchKernelGGL((mask_dchat_kernel<16>), grid, block, 0, s, dcos, P, inv_p, dchat, B, N, C, D, rps);
  ASIS_CHECK_LAUNCH("asis_mask_dchat");
  return ASIS_OK;
}
