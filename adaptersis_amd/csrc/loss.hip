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

// One generalised pass for every loss of segloss/: with x0 = resized logits, x1 = softmax(x0), x2 = softmax(x1)
//   region term on q = x_{n_region}:  per (b, c) sums I = sum q t, Sp = sum q, St = sum t   (tp = I, fp = Sp - I, fn = St - I)
//   CE term (n_ce >= 1): nll = -log softmax(x_{n_ce-1})[t], weighted mean with optional class weights
// partial[(b * nblk + blk) * (C*3 + 2) + {c*3 + {0,1,2}: I, Sp, St ; C*3: sum w nll ; C*3+1: sum w}]
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                           const float* __restrict__ ce_w, int h, int w, int H, int W, int C,
                                                           int n_region, int n_ce, float* __restrict__ partial) {
  __shared__ float red[4][MAXC * 3 + 2];
  const int b = blockIdx.y, nblk = gridDim.x;
  const float* lg = logits + (int64_t)b * h * w * C;
  const int64_t* tg = target + (int64_t)b * H * W;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  float acc[MAXC * 3 + 2];
#pragma unroll
  for (int i = 0; i < MAXC * 3 + 2; ++i) acc[i] = 0.f;
  const int npix = H * W;
  const int nmax = n_region > n_ce ? n_region : n_ce;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += nblk * blockDim.x) {
    const int y = p / W, x = p - y * W;
    float z[MAXC];
    sample_logits(lg, h, w, C, y, x, sh, sw, z);
    const int t = (int)tg[p];
    for (int k = 0; k <= nmax; ++k) {
      if (k == n_region) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) {
            const float tt = (t == c) ? 1.f : 0.f;
            acc[c * 3 + 0] += z[c] * tt;
            acc[c * 3 + 1] += z[c];
            acc[c * 3 + 2] += tt;
          }
      }
      if (k + 1 == n_ce) {  // log-sum-exp on the input of the CE's own softmax
        float m = -INFINITY, zt = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) {
            m = fmaxf(m, z[c]);
            if (c == t) zt = z[c];
          }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) se += expf(z[c] - m);
        const float wt = (ce_w && t >= 0 && t < C) ? ce_w[t] : 1.f;
        acc[MAXC * 3 + 0] += wt * (logf(se) - (zt - m));
        acc[MAXC * 3 + 1] += wt;
      }
      if (k < nmax) softmax_c(z, C);
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < MAXC * 3 + 2; ++i)
    if (i < C * 3 || i >= MAXC * 3) {
      const float v = wave_sum(acc[i]);
      if (lane == 0) red[wid][i] = v;
    }
  __syncthreads();
  const int tx = threadIdx.x;
  if (tx < C * 3 + 2) {
    const int src = tx < C * 3 ? tx : MAXC * 3 + (tx - C * 3);
    partial[((int64_t)b * nblk + blockIdx.x) * (C * 3 + 2) + tx] = (red[0][src] + red[1][src]) + (red[2][src] + red[3][src]);
  }
}

// sums[b][c][3] (fp32, from double accumulation), loss, and the backward coefficients:
//   coef[(b*C + c)*2 + {0,1}] : d region / d q[b,c,pix] = coef0 * t + coef1     (times grad_scale)
//   coef[B*C*2]               : grad_scale / sum of CE weights (0 without a CE term)
// region modes (all derive from I, Sp, St):
//   0 Dice      segloss/dice.py:27-33            1 - mean 2I/(Sp+St+eps)
//   1 soft IoU  segloss/iou_multi.py:38-49       mean [1 - (I+eps)/(Sp+St-I+eps)]
//   2 SoftDice  segloss/dice_loss.py:255-291     -mean (2I+eps)/(Sp+St+eps)            (2tp+fp+fn = Sp+St)
//   3 Tversky   segloss/dice_loss.py:333-372     -mean (I+eps)/((1-a-b)I + a Sp + b St + eps),  a=.3, b=.7
//   4 none      (CE only, segloss/ND_Crossentropy.py:11-32)
__global__ void seg_loss_finalize_kernel(const float* __restrict__ partial, int nblk, int B, int C, float eps, int mode,
                                         int n_ce, float grad_scale, float* __restrict__ sums, float* __restrict__ loss,
                                         float* __restrict__ coef) {
  __shared__ double dsum[256];
  __shared__ double ce_tot[2];
  const int i = threadIdx.x;  // one thread per (b, c)
  const int PS = C * 3 + 2;
  double term = 0.0;
  if (i < B * C) {
    const int b = i / C, c = i - b * C;
    double s0 = 0, s1 = 0, s2 = 0;
    for (int k = 0; k < nblk; ++k) {
      const float* p = partial + ((int64_t)b * nblk + k) * PS + c * 3;
      s0 += p[0];
      s1 += p[1];
      s2 += p[2];
    }
    if (sums) {
      sums[i * 3 + 0] = (float)s0;
      sums[i * 3 + 1] = (float)s1;
      sums[i * 3 + 2] = (float)s2;
    }
    const double bc = (double)B * C, e = (double)eps;
    double c0 = 0.0, c1 = 0.0;
    if (mode == 0) {
      const double S = s1 + s2 + e;
      term = 2.0 * s0 / S;
      c0 = -2.0 / (bc * S);
      c1 = 2.0 * s0 / (bc * S * S);
    } else if (mode == 1) {
      const double In = s0 + e, U = s1 + s2 - s0 + e;
      term = In / U;  // d(-In/U)/dq = t * (-(U + In)/U^2) + In/U^2
      c0 = -(U + In) / (U * U) / bc;
      c1 = In / (U * U) / bc;
    } else if (mode == 2) {
      const double Nn = 2.0 * s0 + e, S = s1 + s2 + e;
      term = Nn / S;
      c0 = -2.0 / (bc * S);
      c1 = Nn / (bc * S * S);
    } else if (mode == 3) {
      const double al = 0.3, be = 0.7, g = 1.0 - al - be;
      const double Nn = s0 + e, Dn = g * s0 + al * s1 + be * s2 + e;
      term = Nn / Dn;  // d(-Nn/Dn)/dq = -(t Dn - Nn (g t + al)) / Dn^2
      c0 = -(Dn - Nn * g) / (Dn * Dn) / bc;
      c1 = Nn * al / (Dn * Dn) / bc;
    }
    coef[i * 2 + 0] = (float)(c0 * grad_scale);
    coef[i * 2 + 1] = (float)(c1 * grad_scale);
  }
  dsum[i] = term;
  if (i < 2) {
    double s = 0.0;
    if (n_ce > 0)
      for (int k = 0; k < B * nblk; ++k) s += partial[(int64_t)k * PS + C * 3 + i];
    ce_tot[i] = s;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (i < o) dsum[i] += dsum[i + o];
    __syncthreads();
  }
  if (i == 0) {
    const double mean = dsum[0] / ((double)B * C);
    double l = (mode == 0 || mode == 1) ? 1.0 - mean : (mode == 4 ? 0.0 : -mean);
    if (n_ce > 0) l += ce_tot[0] / ce_tot[1];
    *loss = (float)l;
    coef[B * C * 2] = n_ce > 0 ? (float)((double)grad_scale / ce_tot[1]) : 0.f;
  }
}

__device__ __forceinline__ void softmax_bwd_c(float* g, const float* p, int C) {  // g <- p * (g - <g, p>)
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) dot += g[c] * p[c];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) g[c] = p[c] * (g[c] - dot);
}

// dz[b, y, x, c] = d loss / d (resized logits), fp32 at (H, W)
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                           const float* __restrict__ coef, const float* __restrict__ ce_w,
                                                           int B, int h, int w, int H, int W, int C, int n_region, int n_ce,
                                                           int mode, float* __restrict__ dz) {
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * h * w * C;
  const int64_t* tg = target + (int64_t)b * H * W;
  float* out = dz + (int64_t)b * H * W * C;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const int npix = H * W;
  const float ce_scale = coef[B * C * 2];
  const int nmax = n_region > n_ce ? n_region : n_ce;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    float p1[MAXC], p2[MAXC];
    sample_logits(lg, h, w, C, y, x, sh, sw, p1);
    if (nmax >= 1) softmax_c(p1, C);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) p2[c] = p1[c];
    if (nmax >= 2) softmax_c(p2, C);
    const int t = (int)tg[p];
    float r[MAXC];  // region gradient wrt q
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      r[c] = (c < C && mode != 4) ? coef[(b * C + c) * 2 + 0] * ((t == c) ? 1.f : 0.f) + coef[(b * C + c) * 2 + 1] : 0.f;
    const float cw = n_ce > 0 ? ce_scale * ((ce_w && t >= 0 && t < C) ? ce_w[t] : 1.f) : 0.f;
    float g[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) g[c] = 0.f;
    if (nmax >= 2) {  // level x1: softmax2 backward of the region term + region at x1 + CE over softmax(x1)
      if (n_region == 2) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) g[c] = r[c];
        softmax_bwd_c(g, p2, C);
      }
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) {
          if (n_region == 1) g[c] += r[c];
          if (n_ce == 2) g[c] += cw * (p2[c] - ((t == c) ? 1.f : 0.f));
        }
      softmax_bwd_c(g, p1, C);
    } else if (nmax == 1 && n_region == 1) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) g[c] = r[c];
      softmax_bwd_c(g, p1, C);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        if (n_region == 0) g[c] += r[c];
        if (n_ce == 1) g[c] += cw * (p1[c] - ((t == c) ? 1.f : 0.f));
        out[(int64_t)p * C + c] = g[c];
      }
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

// out[k] = scale * sum_n partial[n][k]   (double accumulate).  32 row groups x 8 columns per block (K / 8 blocks: 256 for the
// 2048 columns of a LayerNorm-backward partial, one per CU), four independent loads per thread in flight, one LDS pass over the
// row groups at the end.  The partials were just written: the 32-byte row segments come out of L2.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, int n, int K, float scale,
                                                          float* __restrict__ out) {
  constexpr int NG = 32, NCOL = 8;
  __shared__ double red[NG][NCOL + 1];
  const int c = threadIdx.x & (NCOL - 1), g = threadIdx.x / NCOL;
  const int k = blockIdx.x * NCOL + c;
  double s = 0.0;
  if (k < K) {
    const float* p = partial + k;
    int i = g;
    for (; i + 3 * NG < n; i += 4 * NG) {
      const float a0 = p[(int64_t)i * K], a1 = p[(int64_t)(i + NG) * K], a2 = p[(int64_t)(i + 2 * NG) * K],
                  a3 = p[(int64_t)(i + 3 * NG) * K];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; i < n; i += NG) s += (double)p[(int64_t)i * K];
  }
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && k < K) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < NG; ++j) t += red[j][c];
    out[k] = (float)(t * (double)scale);
  }
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
                                                     int C, float* __restrict__ partial, int* __restrict__ counts) {
  __shared__ float red[4][3];
  __shared__ int cnt[MAXC * 3];  // per class: #(target == c), #(argmax == c), #(both)  (ch_iou / isi_iou inputs)
  if (counts) {
    if (threadIdx.x < MAXC * 3) cnt[threadIdx.x] = 0;
    __syncthreads();
  }
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
    if (counts) {
      if (t >= 0 && t < C) atomicAdd(&cnt[t * 3 + 0], 1);
      atomicAdd(&cnt[am * 3 + 1], 1);
      if (am == t) atomicAdd(&cnt[am * 3 + 2], 1);
    }
  }
  if (counts) {
    __syncthreads();
    if (threadIdx.x < C * 3 && cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]);
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

extern "C" int asis_seg_loss_fwd(void* stream, const float* logits, const int64_t* target, const float* ce_weight, int B,
                                 int h, int w, int H, int W, int C, int n_region, int mode, float eps, int n_ce,
                                 float grad_scale, float* partial, float* sums, float* loss, float* coef) {
  ASIS_REQUIRE(logits && target && partial && loss && coef, "asis_seg_loss_fwd: null pointer");
  ASIS_REQUIRE(C >= 1 && C <= MAXC, "asis_seg_loss_fwd: C=%d must be in 1..%d", C, MAXC);
  ASIS_REQUIRE(B * C <= 256 && B <= 65535, "asis_seg_loss_fwd: B*C=%d must be <= 256", B * C);
  ASIS_REQUIRE(n_region >= 0 && n_region <= 2, "asis_seg_loss_fwd: n_region must be 0, 1 or 2");
  ASIS_REQUIRE(n_ce >= 0 && n_ce <= 2, "asis_seg_loss_fwd: n_ce must be 0 (no CE term), 1 or 2");
  ASIS_REQUIRE(mode >= 0 && mode <= 4, "asis_seg_loss_fwd: mode must be 0..4");
  ASIS_REQUIRE(mode != 4 || n_ce > 0, "asis_seg_loss_fwd: mode 4 (no region term) needs a CE term");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nblk = asis_dice_nblk(H, W);
  hipLaunchKernelGGL(seg_loss_fwd_kernel, dim3(nblk, B), dim3(256), 0, s, logits, target, ce_weight, h, w, H, W, C, n_region,
                     n_ce, partial);
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nblk, B, C, eps, mode, n_ce, grad_scale,
                     sums, loss, coef);
  ASIS_CHECK_LAUNCH("asis_seg_loss_fwd");
  return ASIS_OK;
}

extern "C" int asis_seg_loss_bwd(void* stream, const float* logits, const int64_t* target, const float* coef,
                                 const float* ce_weight, int B, int h, int w, int H, int W, int C, int n_region, int mode,
                                 int n_ce, float* dz) {
  ASIS_REQUIRE(logits && target && coef && dz, "asis_seg_loss_bwd: null pointer");
  ASIS_REQUIRE(C >= 1 && C <= MAXC && n_region >= 0 && n_region <= 2 && n_ce >= 0 && n_ce <= 2 && mode >= 0 && mode <= 4,
               "asis_seg_loss_bwd: bad C / n_region / n_ce / mode");
  hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3(asis_dice_nblk(H, W), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     logits, target, coef, ce_weight, B, h, w, H, W, C, n_region, n_ce, mode, dz);
  ASIS_CHECK_LAUNCH("asis_seg_loss_bwd");
  return ASIS_OK;
}

// the two-mode entry points of the train.py / train_mla.py losses (region term only)
extern "C" int asis_dice_fwd(void* stream, const float* logits, const int64_t* target, int B, int h, int w, int H, int W,
                             int C, int n_softmax, float eps, int mode, float grad_scale, float* partial, float* sums,
                             float* loss, float* coef) {
  ASIS_REQUIRE(mode == 0 || mode == 1, "asis_dice_fwd: mode must be 0 (Dice) or 1 (soft IoU)");
  return asis_seg_loss_fwd(stream, logits, target, nullptr, B, h, w, H, W, C, n_softmax, mode, eps, 0, grad_scale, partial,
                           sums, loss, coef);
}

extern "C" int asis_dice_bwd(void* stream, const float* logits, const int64_t* target, const float* coef, int B, int h,
                             int w, int H, int W, int C, int n_softmax, float* dz) {
  ASIS_REQUIRE(n_softmax >= 1 && n_softmax <= 2, "asis_dice_bwd: bad n_softmax");
  return asis_seg_loss_bwd(stream, logits, target, coef, nullptr, B, h, w, H, W, C, n_softmax, 0, 0, dz);
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

extern "C" int asis_ce_acc_counts(void* stream, const float* logits, const int64_t* target, const float* weight, int B,
                                  int h, int w, int H, int W, int C, float* partial, int32_t* counts) {
  ASIS_REQUIRE(logits && target && partial, "asis_ce_acc: null pointer");
  ASIS_REQUIRE(C >= 1 && C <= MAXC, "asis_ce_acc: C=%d must be in 1..%d", C, MAXC);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (counts && hipMemsetAsync(counts, 0, sizeof(int32_t) * C * 3, s) != hipSuccess)
    ASIS_FAIL(ASIS_ELAUNCH, "asis_ce_acc_counts: hipMemsetAsync failed");
  hipLaunchKernelGGL(ce_acc_kernel, dim3(asis_ce_acc_nblk((int64_t)B * H * W)), dim3(256), 0, s, logits, target, weight, B,
                     h, w, H, W, C, partial, counts);
  ASIS_CHECK_LAUNCH("asis_ce_acc");
  return ASIS_OK;
}

extern "C" int asis_ce_acc(void* stream, const float* logits, const int64_t* target, const float* weight, int B, int h,
                           int w, int H, int W, int C, float* partial) {
  return asis_ce_acc_counts(stream, logits, target, weight, B, h, w, H, W, C, partial, nullptr);
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
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((K + 7) / 8), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial, n,
                       K, scale, out);
  }
  ASIS_CHECK_LAUNCH("asis_reduce_rows");
  return ASIS_OK;
}
