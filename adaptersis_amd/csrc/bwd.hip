// Backward companions of the decoder's conv -> BatchNorm(train) -> ReLU -> bilinear-upsample
// stages (backbones/decoders.py:109-135) and the SGD step (train.py:178-191).
//
// Stage backward, given dU = dL/d(upsampled output) [B, fH, fW, C] (fp32, from the next conv's
// dgrad GEMM):
//   1. asis_upsample_bn_relu_bwd:  g = relu'(bn(x)) * upsample^T(dU)     (gather form: every input
//      pixel sums the <= (2f+1)^2 output pixels whose taps touch it -> deterministic, no atomics)
//      + per-block partial sums of  sum g  and  sum g*xhat   (BatchNorm dbeta / dgamma)
//   2. asis_bn_bwd_apply:  dx = gamma*invstd * (g - dbeta/n - xhat*dgamma/n)  -> 16-bit operand of
//      the conv dgrad / wgrad GEMMs, + partial column sums of dx (conv bias gradient)
#include "asis_common.h"

namespace {

inline int grid_for(int64_t total, int block = 256, int cap = 256 * 16) {
  int64_t g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// source taps of output index o for align_corners=True (as upsample kernel in convmisc.hip)
__device__ __forceinline__ void tap_ac_true(int o, float r, int in, int& i0, int& i1, float& l0, float& l1) {
  const float s = r * (float)o;
  i0 = (int)s;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void upsample_bn_relu_bwd_kernel(const float* __restrict__ dU, const float* __restrict__ x,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   float* __restrict__ g, float* __restrict__ partial, int B,
                                                                   int H, int W, int OH, int OW, int C, int CW, int RW) {
  __shared__ float red[2][256 * 4];
  // Tap tables in LDS, built once per workgroup: the source index / fraction of every output row and column, and for every
  // source row / column the exact run of output indices that touch it.  The pixel loop then looks weights up (same values,
  // same ascending (oy, ox) order as evaluating tap_ac_true per candidate pair: bit-identical) instead of spending ~10
  // instructions per candidate on int -> float arithmetic.
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  int* const a0y = reinterpret_cast<int*>(dyn_lds);
  float* const l1y = reinterpret_cast<float*>(a0y + OH);
  int* const a0x = reinterpret_cast<int*>(l1y + OH);
  float* const l1x = reinterpret_cast<float*>(a0x + OW);
  int* const runy = reinterpret_cast<int*>(l1x + OW);  // [H][2] first / last contributing output row
  int* const runx = runy + 2 * H;                      // [W][2]
  const int cpt = C >> 2;
  const float rh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const float irh = rh > 0.f ? 1.f / rh : 0.f, irw = rw > 0.f ? 1.f / rw : 0.f;
  for (int i = threadIdx.x; i < OH; i += blockDim.x) {
    int a0, a1; float l0, l1;
    tap_ac_true(i, rh, H, a0, a1, l0, l1);
    a0y[i] = a0; l1y[i] = l1;
  }
  for (int i = threadIdx.x; i < OW; i += blockDim.x) {
    int a0, a1; float l0, l1;
    tap_ac_true(i, rw, W, a0, a1, l0, l1);
    a0x[i] = a0; l1x[i] = l1;
  }
  __syncthreads();
  auto wgt = [](const int* a0t, const float* l1t, int o, int in, int i) {
    const int a0 = a0t[o];
    const int a1 = a0 + ((a0 < in - 1) ? 1 : 0);
    const float l1 = l1t[o];
    return ((a0 == i) ? 1.f - l1 : 0.f) + ((a1 == i) ? l1 : 0.f);
  };
  auto run = [&](const int* a0t, const float* l1t, int i, int in, int on, float r, float ir, int* out2) {
    int lo, hi;
    if (r > 0.f) {
      lo = (int)floorf(((float)i - 1.f) * ir) - 1;
      hi = (int)ceilf(((float)i + 1.f) * ir) + 1;
    } else { lo = 0; hi = on - 1; }
    if (lo < 0) lo = 0;
    if (hi > on - 1) hi = on - 1;
    while (lo <= hi && wgt(a0t, l1t, lo, in, i) == 0.f) ++lo;
    while (hi >= lo && wgt(a0t, l1t, hi, in, i) == 0.f) --hi;
    out2[0] = lo; out2[1] = hi;
  };
  for (int i = threadIdx.x; i < H; i += blockDim.x) run(a0y, l1y, i, H, OH, rh, irh, runy + 2 * i);
  for (int i = threadIdx.x; i < W; i += blockDim.x) run(a0x, l1x, i, W, OW, rw, irw, runx + 2 * i);
  __syncthreads();
  const int64_t rows = (int64_t)B * H * W;
  // block = CW channel chunks (blockIdx.y picks the channel tile) x RW pixel rows: a thread keeps its channel chunk
  const int cx = threadIdx.x % CW, ry = threadIdx.x / CW;
  const int c = blockIdx.y * CW + cx;
  const bool live = c < cpt;
  const int cc = live ? c : 0;
  const float4 sc = reinterpret_cast<const float4*>(scale)[cc], sh = reinterpret_cast<const float4*>(shift)[cc];
  const float4 mu = reinterpret_cast<const float4*>(mean)[cc], is = reinterpret_cast<const float4*>(invstd)[cc];
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int64_t pix = (int64_t)blockIdx.x * RW + ry; live && pix < rows; pix += (int64_t)gridDim.x * RW) {
    const int64_t i = pix * cpt + c;
    const int iw = (int)(pix % W);
    const int ih = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    const int oy_lo = runy[2 * ih], oy_hi = runy[2 * ih + 1], ox_lo = runx[2 * iw], ox_hi = runx[2 * iw + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // Loads in batches of NR rows x NB columns (the runs are <= 5 x 5 at x2): one load per trip of a data-dependent loop
    // leaves every miss exposed.  Same values; the same ascending (oy, ox) order of the sum while a column run fits one batch
    // (every x2 stage), by row pair and column batch beyond that.
    constexpr int NB = 5, NR = 2;
    for (int oy0 = oy_lo; oy0 <= oy_hi; oy0 += NR) {
      for (int ox0 = ox_lo; ox0 <= ox_hi; ox0 += NB) {
        float wy[NR], wx[NB];
        float4 v[NR][NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) wx[m] = ox0 + m <= ox_hi ? wgt(a0x, l1x, ox0 + m, W, iw) : 0.f;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          const int oy = oy0 + k;
          const bool oky = oy <= oy_hi;
          wy[k] = oky ? wgt(a0y, l1y, oy, H, ih) : 0.f;
          const float* drow = dU + (((int64_t)b * OH + (oky ? oy : oy_hi)) * OW) * C;
#pragma unroll
          for (int m = 0; m < NB; ++m) {
            const int ox = ox0 + m;
            v[k][m] = (oky && ox <= ox_hi) ? reinterpret_cast<const float4*>(drow + (int64_t)ox * C)[c]
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          if (wy[k] == 0.f) continue;
#pragma unroll
          for (int m = 0; m < NB; ++m) {
            if (wx[m] == 0.f) continue;
            const float wt = wy[k] * wx[m];
            acc.x += wt * v[k][m].x; acc.y += wt * v[k][m].y; acc.z += wt * v[k][m].z; acc.w += wt * v[k][m].w;
          }
        }
      }
    }
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    float4 gg;
    gg.x = (xv.x * sc.x + sh.x > 0.f) ? acc.x : 0.f;
    gg.y = (xv.y * sc.y + sh.y > 0.f) ? acc.y : 0.f;
    gg.z = (xv.z * sc.z + sh.z > 0.f) ? acc.z : 0.f;
    gg.w = (xv.w * sc.w + sh.w > 0.f) ? acc.w : 0.f;
    reinterpret_cast<float4*>(g)[i] = gg;
    s1.x += gg.x; s1.y += gg.y; s1.z += gg.z; s1.w += gg.w;
    s2.x += gg.x * (xv.x - mu.x) * is.x;
    s2.y += gg.y * (xv.y - mu.y) * is.y;
    s2.z += gg.z * (xv.z - mu.z) * is.z;
    s2.w += gg.w * (xv.w - mu.w) * is.w;
  }
  reinterpret_cast<float4*>(red[0])[threadIdx.x] = s1;
  reinterpret_cast<float4*>(red[1])[threadIdx.x] = s2;
  __syncthreads();
  if (ry == 0 && live) {
    for (int t = cx + CW; t < CW * RW; t += CW) {
      const float4 a = reinterpret_cast<const float4*>(red[0])[t], bq = reinterpret_cast<const float4*>(red[1])[t];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += bq.x; s2.y += bq.y; s2.z += bq.z; s2.w += bq.w;
    }
    reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * C)[c] = s1;
    reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * C)[c] = s2;
  }
}

// Transpose of BN + ReLU + MaxPool2d(3, stride 2, pad 1) (encoders.py:17-18, the stem's tail), gather form:
// g[b,ih,iw,c] = relu'(bn(x)) * sum over the <= 4 pooling windows that contain (ih,iw) and whose FIRST maximum (scan
// order, as ATen) is this pixel of dy[b,oh,ow,c]; + the BatchNorm partial sums like upsample_bn_relu_bwd_kernel.
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  float* __restrict__ g, float* __restrict__ partial, int B,
                                                                  int H, int W, int OH, int OW, int C, int CW, int RW) {
  __shared__ float red[2][256 * 4];
  const int cpt = C >> 2;
  const int64_t rows = (int64_t)B * H * W;
  const int cx = threadIdx.x % CW, ry = threadIdx.x / CW;
  const int c = blockIdx.y * CW + cx;
  const bool live = c < cpt;
  const int cc = live ? c : 0;
  const float4 sc = reinterpret_cast<const float4*>(scale)[cc], sh = reinterpret_cast<const float4*>(shift)[cc];
  const float4 mu = reinterpret_cast<const float4*>(mean)[cc], is = reinterpret_cast<const float4*>(invstd)[cc];
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  auto act = [&](const float4 v) {
    return make_float4(fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f), fmaxf(v.z * sc.z + sh.z, 0.f),
                       fmaxf(v.w * sc.w + sh.w, 0.f));
  };
  for (int64_t pix = (int64_t)blockIdx.x * RW + ry; live && pix < rows; pix += (int64_t)gridDim.x * RW) {
    const int iw = (int)(pix % W);
    const int ih = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    const float* xb = x + (int64_t)b * H * W * C;
    const float4 xv = reinterpret_cast<const float4*>(xb + ((int64_t)ih * W + iw) * C)[c];
    const float4 me = act(xv);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oh = (ih - 1 + 1) >> 1; oh <= (ih + 1) >> 1; ++oh) {   // windows with 2*oh - 1 <= ih <= 2*oh + 1
      if (oh < 0 || oh >= OH || 2 * oh - 1 > ih || ih > 2 * oh + 1) continue;
      for (int ow = (iw - 1 + 1) >> 1; ow <= (iw + 1) >> 1; ++ow) {
        if (ow < 0 || ow >= OW || 2 * ow - 1 > iw || iw > 2 * ow + 1) continue;
        // is (ih, iw) the first maximum of window (oh, ow)?  per channel: no earlier element >=, no later element >
        bool w0 = true, w1 = true, w2 = true, w3 = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int yy = 2 * oh - 1 + kh, xx = 2 * ow - 1 + kw;
            if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W || (yy == ih && xx == iw)) continue;
            const float4 o = act(reinterpret_cast<const float4*>(xb + ((int64_t)yy * W + xx) * C)[c]);
            const bool earlier = yy < ih || (yy == ih && xx < iw);
            if (earlier) { w0 &= o.x < me.x; w1 &= o.y < me.y; w2 &= o.z < me.z; w3 &= o.w < me.w; }
            else { w0 &= o.x <= me.x; w1 &= o.y <= me.y; w2 &= o.z <= me.z; w3 &= o.w <= me.w; }
          }
        const float4 d = reinterpret_cast<const float4*>(dy + (((int64_t)b * OH + oh) * OW + ow) * C)[c];
        if (w0) acc.x += d.x;
        if (w1) acc.y += d.y;
        if (w2) acc.z += d.z;
        if (w3) acc.w += d.w;
      }
    }
    float4 gg;
    gg.x = me.x > 0.f ? acc.x : 0.f; gg.y = me.y > 0.f ? acc.y : 0.f;
    gg.z = me.z > 0.f ? acc.z : 0.f; gg.w = me.w > 0.f ? acc.w : 0.f;
    reinterpret_cast<float4*>(g + pix * C)[c] = gg;
    s1.x += gg.x; s1.y += gg.y; s1.z += gg.z; s1.w += gg.w;
    s2.x += gg.x * (xv.x - mu.x) * is.x; s2.y += gg.y * (xv.y - mu.y) * is.y;
    s2.z += gg.z * (xv.z - mu.z) * is.z; s2.w += gg.w * (xv.w - mu.w) * is.w;
  }
  reinterpret_cast<float4*>(red[0])[threadIdx.x] = s1;
  reinterpret_cast<float4*>(red[1])[threadIdx.x] = s2;
  __syncthreads();
  if (ry == 0 && live) {
    for (int t = cx + CW; t < CW * RW; t += CW) {
      const float4 a = reinterpret_cast<const float4*>(red[0])[t], bq = reinterpret_cast<const float4*>(red[1])[t];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += bq.x; s2.y += bq.y; s2.z += bq.z; s2.w += bq.w;
    }
    reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * C)[c] = s1;
    reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * C)[c] = s2;
  }
}

// zero-insertion of a stride-2 gradient: out[b, 2i, 2j, :] = in[b, i, j, :], everything else 0 (out is Hd x Wd); a
// stride-1 conv of it with the mirrored weights is the stride-2 conv's input gradient (conv_transpose2d)
template <typename T>
__global__ __launch_bounds__(256) void dilate2_kernel(const T* __restrict__ in, const T* __restrict__ in_lo, T* __restrict__ out,
                                                      T* __restrict__ out_lo, int B, int OH, int OW, int Hd, int Wd, int C) {
  const int cpt = C >> 3;
  const int64_t total = (int64_t)B * Hd * Wd * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt);
    const int64_t pix = i / cpt;
    const int xx = (int)(pix % Wd);
    const int yy = (int)((pix / Wd) % Hd);
    const int b = (int)(pix / ((int64_t)Wd * Hd));
    uint4 v = make_uint4(0, 0, 0, 0), l = v;
    if (!(yy & 1) && !(xx & 1) && (yy >> 1) < OH && (xx >> 1) < OW) {
      const int64_t src = ((((int64_t)b * OH + (yy >> 1)) * OW + (xx >> 1)) * C) / 8 + c;
      v = reinterpret_cast<const uint4*>(in)[src];
      if (in_lo) l = reinterpret_cast<const uint4*>(in_lo)[src];
    }
    reinterpret_cast<uint4*>(out)[i] = v;
    if (out_lo) reinterpret_cast<uint4*>(out_lo)[i] = l;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, float inv_n, T* __restrict__ out,
                                                           T* __restrict__ out_lo, float* __restrict__ partial, int64_t R,
                                                           int C, int CW, int RW, const float* __restrict__ mx_amax = nullptr) {
  __shared__ float red[256 * 4];
  const int cpt = C >> 2;
  const int cx = threadIdx.x % CW, ry = threadIdx.x / CW;
  const int c = blockIdx.y * CW + cx;
  const bool live = c < cpt;
  const int cc = live ? c : 0;
  const float4 mu = reinterpret_cast<const float4*>(mean)[cc], is = reinterpret_cast<const float4*>(invstd)[cc];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[cc];
  const float4 dg = reinterpret_cast<const float4*>(dgamma)[cc], db = reinterpret_cast<const float4*>(dbeta)[cc];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * RW + ry; live && row < R; row += (int64_t)gridDim.x * RW) {
    const int64_t i = row * cpt + c;
    const float4 gv = reinterpret_cast<const float4*>(g)[i], xv = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = ga.x * is.x * (gv.x - db.x * inv_n - (xv.x - mu.x) * is.x * dg.x * inv_n);
    o.y = ga.y * is.y * (gv.y - db.y * inv_n - (xv.y - mu.y) * is.y * dg.y * inv_n);
    o.z = ga.z * is.z * (gv.z - db.z * inv_n - (xv.z - mu.z) * is.z * dg.z * inv_n);
    o.w = ga.w * is.w * (gv.w - db.w * inv_n - (xv.w - mu.w) * is.w * dg.w * inv_n);
    uint2 pk;
    pk.x = pack2<T>(o.x, o.y);
    pk.y = pack2<T>(o.z, o.w);
    reinterpret_cast<uint2*>(out)[i] = pk;
    if (out_lo) {   // rounding residuals, or their MX form (asis_common.h) when the tensor's absolute maximum is given
      pk.x = lo_word2<T>(o.x, o.y, mx_amax);
      pk.y = lo_word2<T>(o.z, o.w, mx_amax);
      reinterpret_cast<uint2*>(out_lo)[i] = pk;
    }
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  reinterpret_cast<float4*>(red)[threadIdx.x] = s;
  __syncthreads();
  if (ry == 0 && live) {
    for (int t = cx + CW; t < CW * RW; t += CW) {
      const float4 a = reinterpret_cast<const float4*>(red)[t];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * C)[c] = s;
  }
}

// max |dx| of the same expression (nothing written): the per-tensor scale of dx's MX form (asis_bn_bwd_apply_mx), one atomic per block
__global__ __launch_bounds__(256) void bn_bwd_absmax_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                            const float* __restrict__ dbeta, float inv_n, float* __restrict__ amax, int64_t R,
                                                            int C, int CW, int RW) {
  __shared__ float red[4];
  const int cpt = C >> 2;
  const int cx = threadIdx.x % CW, ry = threadIdx.x / CW;
  const int c = blockIdx.y * CW + cx;
  const bool live = c < cpt;
  const int cc = live ? c : 0;
  const float4 mu = reinterpret_cast<const float4*>(mean)[cc], is = reinterpret_cast<const float4*>(invstd)[cc];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[cc];
  const float4 dg = reinterpret_cast<const float4*>(dgamma)[cc], db = reinterpret_cast<const float4*>(dbeta)[cc];
  float m = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * RW + ry; live && row < R; row += (int64_t)gridDim.x * RW) {
    const int64_t i = row * cpt + c;
    const float4 gv = reinterpret_cast<const float4*>(g)[i], xv = reinterpret_cast<const float4*>(x)[i];
    const float o0 = ga.x * is.x * (gv.x - db.x * inv_n - (xv.x - mu.x) * is.x * dg.x * inv_n);
    const float o1 = ga.y * is.y * (gv.y - db.y * inv_n - (xv.y - mu.y) * is.y * dg.y * inv_n);
    const float o2 = ga.z * is.z * (gv.z - db.z * inv_n - (xv.z - mu.z) * is.z * dg.z * inv_n);
    const float o3 = ga.w * is.w * (gv.w - db.w * inv_n - (xv.w - mu.w) * is.w * dg.w * inv_n);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(o0), fabsf(o1))), fmaxf(fabsf(o2), fabsf(o3)));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(amax), __builtin_bit_cast(unsigned int, m));
  }
}

// torch.optim.SGD (dampening 0, no Nesterov): g' = g*inv_scale + wd*p ; buf = first ? g' : mom*buf + g' ; p -= lr*buf
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n, float lr, float momentum, float wd, float inv_scale, int first) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pv = p[i];
    const float gv = g[i] * inv_scale + wd * pv;
    const float bv = first ? gv : momentum * buf[i] + gv;
    buf[i] = bv;
    p[i] = pv - lr * bv;
  }
}

// Overflow guard of the static loss scale (16-bit gradient tensors saturate to inf): guard[0] = 1 when any gradient element
// is not finite.  The guarded SGD kernel then leaves p / buf untouched and counts the skipped step in guard[1].
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, int64_t n, int* __restrict__ guard) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = g[i];
    bad |= !(fabsf(v) <= 3.402823466e38f);   // false for inf and NaN
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(guard, 1);
}

__global__ __launch_bounds__(256) void sgd_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                          int64_t n, float lr, float momentum, float wd, float inv_scale, int first,
                                                          int* __restrict__ guard, int count_skip) {
  if (guard[0] != 0) {   // uniform over the grid: every thread reads the same word, written by the kernel before this one
    if (count_skip && blockIdx.x == 0 && threadIdx.x == 0) guard[1] += 1;
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pv = p[i];
    const float gv = g[i] * inv_scale + wd * pv;
    const float bv = first ? gv : momentum * buf[i] + gv;
    buf[i] = bv;
    p[i] = pv - lr * bv;
  }
}

// 16-byte forms of the three kernels above (flat parameter buckets are 16-byte aligned; the scalar forms remain for views that
// are not): two float4 per array in flight per lane.  Round 5: the scalar forms ran the 31 M-parameter UNet bucket at 0.7 TB/s.
__device__ __forceinline__ void sgd4(float4& p, const float4 g, float4& b, float lr, float momentum, float wd, float inv_scale,
                                     int first) {
  float* pp = reinterpret_cast<float*>(&p);
  float* bb = reinterpret_cast<float*>(&b);
  const float* gg = reinterpret_cast<const float*>(&g);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gv = gg[k] * inv_scale + wd * pp[k];
    const float bv = first ? gv : momentum * bb[k] + gv;
    bb[k] = bv;
    pp[k] = pp[k] - lr * bv;
  }
}
__global__ __launch_bounds__(256) void sgd_vec_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                      int64_t n, float lr, float momentum, float wd, float inv_scale, int first,
                                                      int* __restrict__ guard, int count_skip) {
  if (guard && guard[0] != 0) {
    if (count_skip && blockIdx.x == 0 && threadIdx.x == 0) guard[1] += 1;
    return;
  }
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* b4 = reinterpret_cast<float4*>(buf);
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    float4 pa = p4[i], pb = p4[i + stride];
    const float4 ga = g4[i], gb = g4[i + stride];
    float4 ba = first ? make_float4(0.f, 0.f, 0.f, 0.f) : b4[i], bb = first ? make_float4(0.f, 0.f, 0.f, 0.f) : b4[i + stride];
    sgd4(pa, ga, ba, lr, momentum, wd, inv_scale, first);
    sgd4(pb, gb, bb, lr, momentum, wd, inv_scale, first);
    b4[i] = ba; b4[i + stride] = bb;
    p4[i] = pa; p4[i + stride] = pb;
  }
  if (i < n4) {
    float4 pa = p4[i];
    float4 ba = first ? make_float4(0.f, 0.f, 0.f, 0.f) : b4[i];
    sgd4(pa, g4[i], ba, lr, momentum, wd, inv_scale, first);
    b4[i] = ba;
    p4[i] = pa;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {   // the last n % 4 elements
    const int64_t j = (n4 << 2) + threadIdx.x;
    const float pv = p[j];
    const float gv = g[j] * inv_scale + wd * pv;
    const float bv = first ? gv : momentum * buf[j] + gv;
    buf[j] = bv;
    p[j] = pv - lr * bv;
  }
}
// not finite <=> (bits & 0x7fffffff) >= 0x7f800000: an unsigned maximum over the masked words keeps inf AND NaN (fmaxf drops NaN)
__device__ __forceinline__ unsigned absbits_max4(const float4 v, unsigned m) {
  const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
  const unsigned c = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
  return q > m ? q : m;
}
__global__ __launch_bounds__(256) void nonfinite_vec_kernel(const float* __restrict__ g, int64_t n, int* __restrict__ guard) {
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  unsigned m = 0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
    m = absbits_max4(a, m); m = absbits_max4(b, m); m = absbits_max4(c, m); m = absbits_max4(d, m);
  }
  for (; i < n4; i += stride) m = absbits_max4(g4[i], m);
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const unsigned t = __builtin_bit_cast(unsigned, g[(n4 << 2) + threadIdx.x]) & 0x7fffffffu;
    m = t > m ? t : m;
  }
  const bool bad = m >= 0x7f800000u;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(guard, 1);
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, int64_t n, float a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= a;
}

}  // namespace

extern "C" int asis_ew_blocks(int64_t total_chunks) { return grid_for(total_chunks); }

// block shape of the BatchNorm-backward kernels: CW channel chunks (4 channels each) x RW rows, CW*RW <= 256.
// CW = C/4 when that divides 256 (every power-of-two width), else the largest divisor of C/4 that is <= 64
// (UNet widths 96..1536), so each thread keeps one channel chunk for its whole row loop.
static void bn_bwd_shape(int C, int* CW, int* RW, int* tiles) {
  const int cpt = C / 4;
  int cw = 0;
  if (cpt <= 256 && 256 % cpt == 0) cw = cpt;
  else
    for (int k = cpt < 64 ? cpt : 64; k >= 1; --k)
      if (cpt % k == 0) { cw = k; break; }
  if (cw < 8 && cpt > 64) cw = 64;  // awkward widths: partial last tile, guarded in the kernel
  *CW = cw;
  *RW = 256 / cw;
  *tiles = (cpt + cw - 1) / cw;
}
// gridDim.x of those kernels = number of partial-sum rows they write
extern "C" int asis_bn_bwd_nblk(int64_t rows, int C) {
  int CW, RW, tiles;
  if (C < 4 || C % 4) return 1;
  bn_bwd_shape(C, &CW, &RW, &tiles);
  int64_t g = (rows + RW - 1) / RW;
  const int64_t cap = 4096 / tiles > 64 ? 4096 / tiles : 64;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int asis_upsample_bn_relu_bwd(void* stream, const float* dU, const float* x, const float* scale,
                                         const float* shift, const float* mean, const float* invstd, float* g,
                                         float* partial, int B, int H, int W, int C, int factor) {
  ASIS_REQUIRE(dU && x && scale && shift && mean && invstd && g && partial, "asis_upsample_bn_relu_bwd: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C >= 4, "asis_upsample_bn_relu_bwd: C=%d must be a positive multiple of 4", C);
  ASIS_REQUIRE(factor >= 1, "asis_upsample_bn_relu_bwd: bad factor");
  int CW, RW, tiles;
  bn_bwd_shape(C, &CW, &RW, &tiles);
  const int nblk = asis_bn_bwd_nblk((int64_t)B * H * W, C);
  const size_t tables = (size_t)(2 * (H * factor) + 2 * (W * factor) + 2 * H + 2 * W) * 4;  // tap tables, see the kernel
  ASIS_REQUIRE(tables <= 56 * 1024, "asis_upsample_bn_relu_bwd: %d x %d x%d needs %zu bytes of tap tables (limit 56 KiB)", H, W,
               factor, tables);
  hipLaunchKernelGGL(upsample_bn_relu_bwd_kernel, dim3(nblk, tiles), dim3(CW * RW), tables, reinterpret_cast<hipStream_t>(stream),
                     dU, x, scale, shift, mean, invstd, g, partial, B, H, W, H * factor, W * factor, C, CW, RW);
  ASIS_CHECK_LAUNCH("asis_upsample_bn_relu_bwd");
  return ASIS_OK;
}

extern "C" int asis_maxpool_bn_relu_bwd(void* stream, const float* dy, const float* x, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, float* g, float* partial, int B, int H, int W,
                                        int C) {
  ASIS_REQUIRE(dy && x && scale && shift && mean && invstd && g && partial, "asis_maxpool_bn_relu_bwd: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C >= 4, "asis_maxpool_bn_relu_bwd: C=%d must be a positive multiple of 4", C);
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  int CW, RW, tiles;
  bn_bwd_shape(C, &CW, &RW, &tiles);
  const int nblk = asis_bn_bwd_nblk((int64_t)B * H * W, C);
  hipLaunchKernelGGL(maxpool_bn_relu_bwd_kernel, dim3(nblk, tiles), dim3(CW * RW), 0, reinterpret_cast<hipStream_t>(stream), dy, x,
                     scale, shift, mean, invstd, g, partial, B, H, W, OH, OW, C, CW, RW);
  ASIS_CHECK_LAUNCH("asis_maxpool_bn_relu_bwd");
  return ASIS_OK;
}

extern "C" int asis_dilate2(void* stream, int dtype, const void* in, const void* in_lo, void* out, void* out_lo, int B, int OH,
                            int OW, int Hd, int Wd, int C) {
  ASIS_REQUIRE(in && out && (in_lo == nullptr) == (out_lo == nullptr), "asis_dilate2: null pointer / lo halves go together");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_dilate2: bad dtype %d", dtype);
  ASIS_REQUIRE(C % 8 == 0 && C > 0 && Hd >= 2 * OH - 1 && Wd >= 2 * OW - 1, "asis_dilate2: C %% 8 == 0 and Hd >= 2*OH-1 needed");
  const int64_t total = (int64_t)B * Hd * Wd * (C / 8);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((dilate2_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, (const f16*)in, (const f16*)in_lo, (f16*)out,
                       (f16*)out_lo, B, OH, OW, Hd, Wd, C);
  else
    hipLaunchKernelGGL((dilate2_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)in, (const bf16*)in_lo,
                       (bf16*)out, (bf16*)out_lo, B, OH, OW, Hd, Wd, C);
  ASIS_CHECK_LAUNCH("asis_dilate2");
  return ASIS_OK;
}

static int bn_bwd_apply_impl(void* stream, int dtype, const float* g, const float* x, const float* mean, const float* invstd,
                             const float* gamma, const float* dgamma, const float* dbeta, double count, void* out, void* out_lo,
                             const float* mx_amax, float* partial, int64_t R, int C) {
  ASIS_REQUIRE(g && x && mean && invstd && gamma && dgamma && dbeta && out && partial, "asis_bn_bwd_apply: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C >= 4, "asis_bn_bwd_apply: C=%d must be a positive multiple of 4", C);
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_bn_bwd_apply: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float inv_n = (float)(1.0 / count);
  int CW, RW, tiles;
  bn_bwd_shape(C, &CW, &RW, &tiles);
  const int nblk = asis_bn_bwd_nblk(R, C);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<f16>), dim3(nblk, tiles), dim3(CW * RW), 0, s, g, x, mean, invstd, gamma, dgamma,
                       dbeta, inv_n, reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), partial, R, C, CW, RW, mx_amax);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16>), dim3(nblk, tiles), dim3(CW * RW), 0, s, g, x, mean, invstd, gamma, dgamma,
                       dbeta, inv_n, reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(out_lo), partial, R, C, CW, RW, mx_amax);
  ASIS_CHECK_LAUNCH("asis_bn_bwd_apply");
  return ASIS_OK;
}
extern "C" int asis_bn_bwd_apply(void* stream, int dtype, const float* g, const float* x, const float* mean,
                                 const float* invstd, const float* gamma, const float* dgamma, const float* dbeta,
                                 double count, void* out, void* out_lo, float* partial, int64_t R, int C) {
  return bn_bwd_apply_impl(stream, dtype, g, x, mean, invstd, gamma, dgamma, dbeta, count, out, out_lo, nullptr, partial, R, C);
}
extern "C" int asis_bn_bwd_apply_mx(void* stream, int dtype, const float* g, const float* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* dgamma, const float* dbeta, double count, void* out, void* out_mx,
                                    const float* amax, float* partial, int64_t R, int C) {
  ASIS_REQUIRE(out_mx && amax, "asis_bn_bwd_apply_mx: null pointer");
  return bn_bwd_apply_impl(stream, dtype, g, x, mean, invstd, gamma, dgamma, dbeta, count, out, out_mx, amax, partial, R, C);
}
extern "C" int asis_bn_bwd_absmax(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* gamma,
                                  const float* dgamma, const float* dbeta, double count, float* amax, int64_t R, int C) {
  ASIS_REQUIRE(g && x && mean && invstd && gamma && dgamma && dbeta && amax, "asis_bn_bwd_absmax: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C >= 4, "asis_bn_bwd_absmax: C=%d must be a positive multiple of 4", C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ASIS_REQUIRE(hipMemsetAsync(amax, 0, sizeof(float), s) == hipSuccess, "asis_bn_bwd_absmax: memset failed");
  if (R == 0) return ASIS_OK;
  const float inv_n = (float)(1.0 / count);
  int CW, RW, tiles;
  bn_bwd_shape(C, &CW, &RW, &tiles);
  int nblk = asis_bn_bwd_nblk(R, C);
  if (nblk * tiles > 2048) nblk = 2048 / tiles > 0 ? 2048 / tiles : 1;     // one atomic per block: keep their number small
  hipLaunchKernelGGL(bn_bwd_absmax_kernel, dim3(nblk, tiles), dim3(CW * RW), 0, s, g, x, mean, invstd, gamma, dgamma, dbeta, inv_n, amax,
                     R, C, CW, RW);
  ASIS_CHECK_LAUNCH("asis_bn_bwd_absmax");
  return ASIS_OK;
}

extern "C" int asis_sgd_momentum(void* stream, float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                                 float weight_decay, float inv_scale, int first_step) {
  ASIS_REQUIRE(p && g && buf && n >= 0, "asis_sgd_momentum: bad arguments");
  if (n == 0) return ASIS_OK;
  if (asis_aligned16(p) && asis_aligned16(g) && asis_aligned16(buf))
    hipLaunchKernelGGL(sgd_vec_kernel, dim3(grid_for(n / 8 + 1, 256, 256 * 32)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p,
                       g, buf, n, lr, momentum, weight_decay, inv_scale, first_step, (int*)nullptr, 0);
  else
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, buf, n, lr,
                       momentum, weight_decay, inv_scale, first_step);
  ASIS_CHECK_LAUNCH("asis_sgd_momentum");
  return ASIS_OK;
}

extern "C" int asis_grad_guard(void* stream, const float* g, int64_t n, int32_t* guard, int reset) {
  ASIS_REQUIRE(g && guard && n >= 0, "asis_grad_guard: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (reset) ASIS_REQUIRE(hipMemsetAsync(guard, 0, sizeof(int32_t), s) == hipSuccess, "asis_grad_guard: memset failed");
  if (n == 0) return ASIS_OK;
  if (asis_aligned16(g))
    hipLaunchKernelGGL(nonfinite_vec_kernel, dim3(grid_for(n / 16 + 1, 256, 256 * 32)), dim3(256), 0, s, g, n, reinterpret_cast<int*>(guard));
  else
    hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for(n)), dim3(256), 0, s, g, n, reinterpret_cast<int*>(guard));
  ASIS_CHECK_LAUNCH("asis_grad_guard");
  return ASIS_OK;
}

extern "C" int asis_sgd_momentum_guarded(void* stream, float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                                         float weight_decay, float inv_scale, int first_step, int32_t* guard, int count_skip) {
  ASIS_REQUIRE(p && g && buf && guard && n >= 0, "asis_sgd_momentum_guarded: bad arguments");
  if (n == 0) return ASIS_OK;
  if (asis_aligned16(p) && asis_aligned16(g) && asis_aligned16(buf))
    hipLaunchKernelGGL(sgd_vec_kernel, dim3(grid_for(n / 8 + 1, 256, 256 * 32)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p,
                       g, buf, n, lr, momentum, weight_decay, inv_scale, first_step, reinterpret_cast<int*>(guard), count_skip);
  else
    hipLaunchKernelGGL(sgd_guarded_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, buf, n,
                       lr, momentum, weight_decay, inv_scale, first_step, reinterpret_cast<int*>(guard), count_skip);
  ASIS_CHECK_LAUNCH("asis_sgd_momentum_guarded");
  return ASIS_OK;
}

// 16-bit transport form of a gradient range for the all-reduce (parallel.StageReducer(compress=...)): fp32 -> bf16 (RNE; bf16
// keeps fp32's exponent range, so the unscaled ~1e-7 Dice gradients survive) and back.  n % 4 == 0 (bucket ranges are 16-byte
// aligned), 16 bytes in / 8 bytes out per lane.
__global__ void grad_pack_bf16_kernel(const float4* __restrict__ g, uint2* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = g[i];
    uint2 w;
    w.x = pack2<bf16>(v.x, v.y);
    w.y = pack2<bf16>(v.z, v.w);
    out[i] = w;
  }
}
__global__ void grad_unpack_bf16_kernel(const uint2* __restrict__ in, float4* __restrict__ g, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const uint2 w = in[i];
    float4 v;
    unpack2<bf16>(w.x, v.x, v.y);
    unpack2<bf16>(w.y, v.z, v.w);
    g[i] = v;
  }
}

extern "C" int asis_grad_pack_bf16(void* stream, const float* g, int64_t n, void* out) {
  ASIS_REQUIRE(g && out && n >= 0 && n % 4 == 0, "asis_grad_pack_bf16: null pointer or n=%ld not a multiple of 4", (long)n);
  ASIS_REQUIRE(asis_aligned16(g) && (reinterpret_cast<uintptr_t>(out) & 7) == 0, "asis_grad_pack_bf16: misaligned buffers");
  if (n == 0) return ASIS_OK;
  hipLaunchKernelGGL(grad_pack_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(g), reinterpret_cast<uint2*>(out), n / 4);
  ASIS_CHECK_LAUNCH("asis_grad_pack_bf16");
  return ASIS_OK;
}

extern "C" int asis_grad_unpack_bf16(void* stream, const void* in, int64_t n, float* g) {
  ASIS_REQUIRE(g && in && n >= 0 && n % 4 == 0, "asis_grad_unpack_bf16: null pointer or n=%ld not a multiple of 4", (long)n);
  ASIS_REQUIRE(asis_aligned16(g) && (reinterpret_cast<uintptr_t>(in) & 7) == 0, "asis_grad_unpack_bf16: misaligned buffers");
  if (n == 0) return ASIS_OK;
  hipLaunchKernelGGL(grad_unpack_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint2*>(in), reinterpret_cast<float4*>(g), n / 4);
  ASIS_CHECK_LAUNCH("asis_grad_unpack_bf16");
  return ASIS_OK;
}

extern "C" int asis_zero(void* stream, void* p, int64_t bytes) {
  ASIS_REQUIRE(p && bytes >= 0, "asis_zero: bad arguments");
  if (bytes == 0) return ASIS_OK;
  ASIS_REQUIRE(hipMemsetAsync(p, 0, (size_t)bytes, reinterpret_cast<hipStream_t>(stream)) == hipSuccess, "asis_zero: hipMemsetAsync failed");
  return ASIS_OK;
}

extern "C" int asis_scale_f32(void* stream, float* x, int64_t n, float a) {
  ASIS_REQUIRE(x && n >= 0, "asis_scale_f32: bad arguments");
  if (n == 0) return ASIS_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n, a);
  ASIS_CHECK_LAUNCH("asis_scale_f32");
  return ASIS_OK;
}
