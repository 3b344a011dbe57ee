// Per-pixel segmentation losses, fused with the final bilinear resize of the logits.
//
// Training loss of train.py:422-428:
//     out  = F.interpolate(logits, size=(H, W), mode="bilinear")      (align_corners=False)
//     prob = softmax_C(out)                                           train.py:424
//     loss = DC(C)(prob, target)  with  p = softmax_C(prob)  (second softmax, segloss/dice.py:23)
//            dice[b,c] = 2 sum(p t) / (sum p + sum t + 1e-19) ;  loss = 1 - mean_{b,c} dice
// One pass reads the NHWC logits (4 taps) and the int64 target once and keeps the C running
// sums per thread; wave64 shuffles + one LDS step reduce them per block; the per-block partials
// are summed in double by the finalize kernel (deterministic, no atomics).
// The backward recomputes the two softmaxes instead of storing probabilities:
//     dL/dp[b,c,pix] = a[b,c] * t + g[b,c]   with  a = -2/(BC S), g = 2 I/(BC S^2), S = sum p + sum t + eps
// then softmax^T twice, written at (H, W); asis_resize_bilinear_bwd gathers it back to the
// decoder's (h, w) grid as the 16-bit operand of the dgrad / wgrad GEMMs.
#include "asis_common.h"

namespace {

constexpr int MAXC = 16;

struct Tap {
  int i0, i1;
  float l0, l1;
};
// area_pixel_compute_source_index(align_corners=False): src = scale*(dst+0.5)-0.5, clamped at 0
__device__ __forceinline__ Tap tap_ac_false(int dst, float scale, int in) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  Tap t;
  t.i0 = (int)s;
  if (t.i0 > in - 1) t.i0 = in - 1;
  t.i1 = t.i0 + ((t.i0 < in - 1) ? 1 : 0);
  t.l1 = s - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

// interpolated logits of output pixel (y, x) -> z[C]
__device__ __forceinline__ void sample_logits(const float* __restrict__ lg, int h, int w, int C, int y, int x, float sh,
                                              float sw, float* z) {
  const Tap ty = tap_ac_false(y, sh, h), tx = tap_ac_false(x, sw, w);
  const float* p00 = lg + ((int64_t)ty.i0 * w + tx.i0) * C;
  const float* p01 = lg + ((int64_t)ty.i0 * w + tx.i1) * C;
  const float* p10 = lg + ((int64_t)ty.i1 * w + tx.i0) * C;
  const float* p11 = lg + ((int64_t)ty.i1 * w + tx.i1) * C;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) z[c] = ty.l0 * (tx.l0 * p00[c] + tx.l1 * p01[c]) + ty.l1 * (tx.l0 * p10[c] + tx.l1 * p11[c]);
}

__device__ __forceinline__ void softmax_c(float* z, int C) {
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) m = fmaxf(m, z[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) {
      z[c] = __expf(z[c] - m);
      s += z[c];
    }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) z[c] *= inv;
}

// partial[(b * nblk + blk) * C*3 + c*3 + {0: sum p t, 1: sum p, 2: sum t}]
__global__ __launch_bounds__(256) void dice_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                       int h, int w, int H, int W, int C, int n_softmax,
                                                       float* __restrict__ partial) {
  __shared__ float red[4][MAXC * 3];
  const int b = blockIdx.y, nblk = gridDim.x;
  const float* lg = logits + (int64_t)b * h * w * C;
  const int64_t* tg = target + (int64_t)b * H * W;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  float acc[MAXC * 3];
#pragma unroll
  for (int i = 0; i < MAXC * 3; ++i) acc[i] = 0.f;
  const int npix = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += nblk * blockDim.x) {
    const int y = p / W, x = p - y * W;
    float z[MAXC];
    sample_logits(lg, h, w, C, y, x, sh, sw, z);
    for (int k = 0; k < n_softmax; ++k) softmax_c(z, C);
    const int t = (int)tg[p];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        const float tt = (t == c) ? 1.f : 0.f;
        acc[c * 3 + 0] += z[c] * tt;
        acc[c * 3 + 1] += z[c];
        acc[c * 3 + 2] += tt;
      }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < MAXC * 3; ++i)
    if (i < C * 3) {
      const float v = wave_sum(acc[i]);
      if (lane == 0) red[wid][i] = v;
    }
  __syncthreads();
  if (threadIdx.x < C * 3)
    partial[((int64_t)b * nblk + blockIdx.x) * C * 3 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// sums[b][c][3] (fp32, from double accumulation), loss, and the backward coefficients coef[b][c][2]
// mode 0: Dice (segloss/dice.py:27-33)  loss = 1 - mean_{b,c} 2I/(Sp+St+eps)
// mode 1: soft IoU (segloss/iou_multi.py:38-49, eps = smooth)  loss = mean_c mean_b [1 - (I+eps)/(Sp+St-I+eps)]
__global__ void dice_finalize_kernel(const float* __restrict__ partial, int nblk, int B, int C, float eps, int mode,
                                     float grad_scale, float* __restrict__ sums, float* __restrict__ loss,
                                     float* __restrict__ coef) {
  __shared__ double dsum[256];
  const int i = threadIdx.x;  // one thread per (b, c)
  double dice = 0.0;
  if (i < B * C) {
    const int b = i / C, c = i - b * C;
    double s0 = 0, s1 = 0, s2 = 0;
    for (int k = 0; k < nblk; ++k) {
      const float* p = partial + ((int64_t)b * nblk + k) * C * 3 + c * 3;
      s0 += p[0];
      s1 += p[1];
      s2 += p[2];
    }
    if (sums) {
      sums[i * 3 + 0] = (float)s0;
      sums[i * 3 + 1] = (float)s1;
      sums[i * 3 + 2] = (float)s2;
    }
    const double bc = (double)B * C;
    if (mode == 0) {
      const double S = s1 + s2 + (double)eps;
      dice = 2.0 * s0 / S;
      coef[i * 2 + 0] = (float)(-2.0 / (bc * S) * grad_scale);          // multiplies t
      coef[i * 2 + 1] = (float)(2.0 * s0 / (bc * S * S) * grad_scale);  // constant term
    } else {
      const double In = s0 + (double)eps, U = s1 + s2 - s0 + (double)eps;
      dice = In / U;  // d(-In/U)/dp = t * (-(U + In)/U^2) + In/U^2
      coef[i * 2 + 0] = (float)(-(U + In) / (U * U) / bc * grad_scale);
      coef[i * 2 + 1] = (float)(In / (U * U) / bc * grad_scale);
    }
  }
  dsum[i] = dice;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (i < o) dsum[i] += dsum[i + o];
    __syncthreads();
  }
  if (i == 0) *loss = (float)(1.0 - dsum[0] / ((double)B * C));
}

// dz[b, y, x, c] = d loss / d (resized logits), fp32 at (H, W)
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                       const float* __restrict__ coef, int h, int w, int H, int W, int C,
                                                       int n_softmax, float* __restrict__ dz) {
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * h * w * C;
  const int64_t* tg = target + (int64_t)b * H * W;
  float* out = dz + (int64_t)b * H * W * C;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const int npix = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    float z[MAXC], p1[MAXC];
    sample_logits(lg, h, w, C, y, x, sh, sw, z);
    softmax_c(z, C);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) p1[c] = z[c];
    if (n_softmax > 1) softmax_c(z, C);  // z = p2
    const int t = (int)tg[p];
    float g[MAXC];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        g[c] = coef[(b * C + c) * 2 + 0] * ((t == c) ? 1.f : 0.f) + coef[(b * C + c) * 2 + 1];
        dot += g[c] * z[c];
      }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) g[c] = z[c] * (g[c] - dot);  // through the last softmax
    if (n_softmax > 1) {
      float dot1 = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) dot1 += g[c] * p1[c];
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) g[c] = p1[c] * (g[c] - dot1);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) out[(int64_t)p * C + c] = g[c];
  }
}

// Transpose of the bilinear resize (align_corners=False), gather form (deterministic):
// dsrc[b, sy, sx, c] = sum over output pixels whose taps touch (sy, sx).  Writes a 16-bit NHWC
// tensor with CP (>= C, multiple of 8) channels (pad channels zero) and per-block column sums
// (conv bias gradient): partial[blk][C].
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dz, int B, int H, int W, int h, int w,
                                                         int C, int CP, T* __restrict__ out, T* __restrict__ out_lo,
                                                         float* __restrict__ partial) {
  __shared__ float red[4][MAXC];
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const float ish = (float)H / (float)h, isw = (float)W / (float)w;
  const int64_t total = (int64_t)B * h * w;
  float csum[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) csum[c] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int sx = (int)(i % w);
    const int sy = (int)((i / w) % h);
    const int b = (int)(i / ((int64_t)w * h));
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    int y_lo = (int)floorf(((float)sy - 1.f + 0.5f) * ish - 0.5f) - 1, y_hi = (int)ceilf(((float)sy + 1.f + 0.5f) * ish - 0.5f) + 1;
    int x_lo = (int)floorf(((float)sx - 1.f + 0.5f) * isw - 0.5f) - 1, x_hi = (int)ceilf(((float)sx + 1.f + 0.5f) * isw - 0.5f) + 1;
    if (y_lo < 0) y_lo = 0;
    if (x_lo < 0) x_lo = 0;
    if (y_hi > H - 1) y_hi = H - 1;
    if (x_hi > W - 1) x_hi = W - 1;
    for (int y = y_lo; y <= y_hi; ++y) {
      const Tap ty = tap_ac_false(y, sh, h);
      const float wy = ((ty.i0 == sy) ? ty.l0 : 0.f) + ((ty.i1 == sy) ? ty.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int x = x_lo; x <= x_hi; ++x) {
        const Tap tx = tap_ac_false(x, sw, w);
        const float wx = ((tx.i0 == sx) ? tx.l0 : 0.f) + ((tx.i1 == sx) ? tx.l1 : 0.f);
        if (wx == 0.f) continue;
        const float* src = dz + (((int64_t)b * H + y) * W + x) * C;
        const float wt = wy * wx;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) acc[c] += wt * src[c];
      }
    }
    T* o = out + i * CP;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < CP) {
        const float v = c < C ? acc[c] : 0.f;
        o[c] = to_t16<T>(v);
        if (out_lo) out_lo[i * CP + c] = to_t16<T>(lo_part<T>(v));
      }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) csum[c] += acc[c];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) {
      const float v = wave_sum(csum[c]);
      if (lane == 0) red[wid][c] = v;
    }
  __syncthreads();
  if (threadIdx.x < C)
    partial[(int64_t)blockIdx.x * C + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[k] = scale * sum_n partial[n][k]   (double accumulate)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, int n, int K, float scale,
                                                          float* __restrict__ out) {
  __shared__ double red[256];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[(int64_t)i * K + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = (float)(red[0] * (double)scale);
}

// F.interpolate(x, size=(H, W), mode="bilinear") (align_corners=False) on NHWC fp32 (decoders.py:88, train.py:422)
__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* __restrict__ x, int B, int h, int w, int H, int W, int C,
                                                         float* __restrict__ out) {
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const int b = (int)(i / ((int64_t)W * H));
    float z[MAXC];
    sample_logits(x + (int64_t)b * h * w * C, h, w, C, yy, xx, sh, sw, z);
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) out[i * C + c] = z[c];
  }
}

// Validation metrics of train.py:616-642 in one pass over the (resized) logits: class-weighted cross entropy
// numerator / denominator (nn.CrossEntropyLoss(weight) = sum w[t]*nll / sum w[t]) and the number of pixels
// whose argmax equals the target.  partial[blk][3] = {sum w*nll, sum w, correct}.
__global__ __launch_bounds__(256) void ce_acc_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, int B, int h, int w, int H, int W,
                                                     int C, float* __restrict__ partial) {
  __shared__ float red[4][3];
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const int64_t total = (int64_t)B * H * W;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int b = (int)(i / ((int64_t)W * H));
    float z[MAXC];
    sample_logits(logits + (int64_t)b * h * w * C, h, w, C, y, x, sh, sw, z);
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C && z[c] > m) { m = z[c]; am = c; }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) se += __expf(z[c] - m);
    const int t = (int)target[i];
    float zt = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c == t) zt = z[c];
    const float wt = weight ? weight[t] : 1.f;
    a0 += wt * (m + __logf(se) - zt);
    a1 += wt;
    a2 += (am == t) ? 1.f : 0.f;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
  if (lane == 0) { red[wid][0] = a0; red[wid][1] = a1; red[wid][2] = a2; }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[(int64_t)blockIdx.x * 3 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// few rows, many columns (split-K slabs of the weight gradients): one thread per column, coalesced rows
__global__ __launch_bounds__(256) void reduce_rows_wide_kernel(const float* __restrict__ partial, int n, int64_t K,
                                                               float scale, float* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)partial[(int64_t)i * K + k];
    out[k] = (float)(s * (double)scale);
  }
}

}  // namespace

extern "C" int asis_dice_nblk(int H, int W) {
  int n = (H * W + 256 * 8 - 1) / (256 * 8);
  if (n > 512) n = 512;
  if (n < 1) n = 1;
  return n;
}

extern "C" int asis_dice_fwd(void* stream, const float* logits, const int64_t* target, int B, int h, int w, int H, int W,
                             int C, int n_softmax, float eps, int mode, float grad_scale, float* partial, float* sums,
                             float* loss, float* coef) {
  ASIS_REQUIRE(logits && target && partial && loss && coef, "asis_dice_fwd: null pointer");
  ASIS_REQUIRE(C >= 1 && C <= MAXC, "asis_dice_fwd: C=%d must be in 1..%d", C, MAXC);
  ASIS_REQUIRE(B * C <= 256 && B <= 65535, "asis_dice_fwd: B*C=%d must be <= 256", B * C);
  ASIS_REQUIRE(n_softmax >= 0 && n_softmax <= 2, "asis_dice_fwd: n_softmax must be 0, 1 or 2");
  ASIS_REQUIRE(mode == 0 || mode == 1, "asis_dice_fwd: mode must be 0 (Dice) or 1 (soft IoU)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nblk = asis_dice_nblk(H, W);
  hipLaunchKernelGGL(dice_fwd_kernel, dim3(nblk, B), dim3(256), 0, s, logits, target, h, w, H, W, C, n_softmax, partial);
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nblk, B, C, eps, mode, grad_scale, sums, loss, coef);
  ASIS_CHECK_LAUNCH("asis_dice_fwd");
  return ASIS_OK;
}

extern "C" int asis_dice_bwd(void* stream, const float* logits, const int64_t* target, const float* coef, int B, int h,
                             int w, int H, int W, int C, int n_softmax, float* dz) {
  ASIS_REQUIRE(logits && target && coef && dz, "asis_dice_bwd: null pointer");
  ASIS_REQUIRE(C >= 1 && C <= MAXC && n_softmax >= 1 && n_softmax <= 2, "asis_dice_bwd: bad C / n_softmax");
  hipLaunchKernelGGL(dice_bwd_kernel, dim3(asis_dice_nblk(H, W), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     logits, target, coef, h, w, H, W, C, n_softmax, dz);
  ASIS_CHECK_LAUNCH("asis_dice_bwd");
  return ASIS_OK;
}

extern "C" int asis_resize_bilinear_fwd(void* stream, const float* x, int B, int h, int w, int H, int W, int C, float* out) {
  ASIS_REQUIRE(x && out && C >= 1 && C <= MAXC, "asis_resize_bilinear_fwd: bad arguments (C <= %d)", MAXC);
  int64_t g = ((int64_t)B * H * W + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(resize_fwd_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, B, h, w, H,
                     W, C, out);
  ASIS_CHECK_LAUNCH("asis_resize_bilinear_fwd");
  return ASIS_OK;
}

extern "C" int asis_ce_acc_nblk(int64_t total_pixels) {
  int64_t n = (total_pixels + 256 * 8 - 1) / (256 * 8);
  if (n > 2048) n = 2048;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int asis_ce_acc(void* stream, const float* logits, const int64_t* target, const float* weight, int B, int h,
                           int w, int H, int W, int C, float* partial) {
  ASIS_REQUIRE(logits && target && partial, "asis_ce_acc: null pointer");
  ASIS_REQUIRE(C >= 1 && C <= MAXC, "asis_ce_acc: C=%d must be in 1..%d", C, MAXC);
  hipLaunchKernelGGL(ce_acc_kernel, dim3(asis_ce_acc_nblk((int64_t)B * H * W)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), logits, target, weight, B, h, w, H, W, C, partial);
  ASIS_CHECK_LAUNCH("asis_ce_acc");
  return ASIS_OK;
}

extern "C" int asis_resize_bwd_nblk(int64_t total_pixels) {
  int64_t n = (total_pixels + 255) / 256;
  if (n > 4096) n = 4096;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int asis_resize_bilinear_bwd(void* stream, int dtype, const float* dz, int B, int H, int W, int h, int w, int C,
                                        int CP, void* out, void* out_lo, float* partial) {
  ASIS_REQUIRE(dz && out && partial, "asis_resize_bilinear_bwd: null pointer");
  ASIS_REQUIRE(C >= 1 && C <= MAXC && CP >= C && CP <= MAXC, "asis_resize_bilinear_bwd: bad C=%d CP=%d", C, CP);
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16 || dtype == ASIS_F32, "asis_resize_bilinear_bwd: bad dtype %d", dtype);
  ASIS_REQUIRE(dtype == ASIS_F32 || CP % 8 == 0, "asis_resize_bilinear_bwd: 16-bit output needs CP %% 8 == 0");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nblk = asis_resize_bwd_nblk((int64_t)B * h * w);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((resize_bwd_kernel<f16>), dim3(nblk), dim3(256), 0, s, dz, B, H, W, h, w, C, CP,
                       reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), partial);
  else if (dtype == ASIS_BF16)
    hipLaunchKernelGGL((resize_bwd_kernel<bf16>), dim3(nblk), dim3(256), 0, s, dz, B, H, W, h, w, C, CP,
                       reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(out_lo), partial);
  else
    hipLaunchKernelGGL((resize_bwd_kernel<float>), dim3(nblk), dim3(256), 0, s, dz, B, H, W, h, w, C, CP,
                       reinterpret_cast<float*>(out), (float*)nullptr, partial);
  ASIS_CHECK_LAUNCH("asis_resize_bilinear_bwd");
  return ASIS_OK;
}

extern "C" int asis_reduce_rows(void* stream, const float* partial, int n, int K, float scale, float* out) {
  ASIS_REQUIRE(partial && out && n > 0 && K > 0, "asis_reduce_rows: bad arguments");
  if (n <= 64 || K > 65535) {
    int64_t g = ((int64_t)K + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(reduce_rows_wide_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       partial, n, (int64_t)K, scale, out);
  } else {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(K), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial, n, K,
                       scale, out);
  }
  ASIS_CHECK_LAUNCH("asis_reduce_rows");
  return ASIS_OK;
}
