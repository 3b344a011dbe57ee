// HBM-bound companions of the implicit-GEMM convolutions: stem conv (Cin=3), BatchNorm
// (train-mode batch statistics) finalisation and fused apply kernels, weight packing, and the
// decoder-input assembly.  All tensors between kernels are NHWC; one thread owns 4-8 contiguous
// channels so every access is 16 bytes.
//
// Reference sites: backbones/encoders.py:9-47 (stem / SyncBatchNorm / ReLU / MaxPool),
// backbones/decoders.py:109-135 (conv -> BatchNorm2d -> ReLU -> Upsample(2, bilinear,
// align_corners=True)), train.py:389-406 (rearrange + pad + concat).
#include "asis_common.h"

namespace {

inline int grid_for(int64_t total, int block = 256, int cap = 256 * 32) {
  int64_t g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- stem conv: img fp32 NCHW [B,3,H,W] -> fp32 NHWC [B,OH,OW,Cout]; w [Cout,3,3,3] ----------
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                         float* __restrict__ out, int B, int H, int W, int OH, int OW,
                                                         int Cout, int stride, int pad) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [27][Cout]
  for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) {
    const int co = i % Cout, k = i / Cout;  // k = ci*9 + kh*3 + kw
    wl[i] = w[co * 27 + k];
  }
  __syncthreads();
  // grid = (column chunks, output rows, images): no 64-bit index division; the 27 taps are fetched unconditionally from
  // coordinates clamped into the image and an outside tap is zeroed afterwards (a load under `if (inside)` is waited for on
  // the spot: 27 exposed latencies per thread)
  const int cpp = Cout >> 2;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= OW * cpp) return;
  const int ow = j / cpp, c = j - ow * cpp;
  const int oh = blockIdx.y, b = blockIdx.z;
  float v[27];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
        const bool in = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        const float x = img[(((int64_t)b * 3 + ci) * H + ihc) * W + iwc];
        v[ci * 9 + kh * 3 + kw] = in ? x : 0.f;
      }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float4 ww = reinterpret_cast<const float4*>(wl + k * Cout)[c];
    acc.x += v[k] * ww.x;
    acc.y += v[k] * ww.y;
    acc.z += v[k] * ww.z;
    acc.w += v[k] * ww.w;
  }
  reinterpret_cast<float4*>(out + (((int64_t)b * OH + oh) * OW + ow) * Cout)[c] = acc;
}

// ---- column statistics of an fp32 [R, C] matrix: partial[blk][{sum,sumsq}][C] ------------------
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, int64_t R, int C,
                                                       int64_t rows_per_block, float* __restrict__ partial) {
  __shared__ float red[2][256 * 4];
  const int cpt = C >> 2;
  const int cw = cpt < 256 ? cpt : 256;
  const int nrg = 256 / cw;
  const int rg = threadIdx.x / cw, cl = threadIdx.x - rg * cw;
  const bool active = rg < nrg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  for (int cb = 0; cb < cpt; cb += cw) {
    const int c = cb + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (active && c < cpt) {
      for (int64_t r = r0 + rg; r < r1; r += nrg) {
        const float4 v = reinterpret_cast<const float4*>(x + r * C)[c];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
      }
    }
    reinterpret_cast<float4*>(red[0])[threadIdx.x] = s;
    reinterpret_cast<float4*>(red[1])[threadIdx.x] = q;
    __syncthreads();
    if (rg == 0 && c < cpt) {
      for (int g = 1; g < nrg; ++g) {
        const float4 a = reinterpret_cast<const float4*>(red[0])[g * cw + cl];
        const float4 bq = reinterpret_cast<const float4*>(red[1])[g * cw + cl];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        q.x += bq.x; q.y += bq.y; q.z += bq.z; q.w += bq.w;
      }
      reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * C)[c] = s;
      reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * C)[c] = q;
    }
    __syncthreads();
  }
}

// ---- reduce partial[nparts][2][C] -> sums[2][C] (double accumulate, fp32 result pairs hi/lo) ----
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nparts, int C,
                                                              double* __restrict__ sums) {
  __shared__ double red[2][256];
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) {
    s += (double)partial[((int64_t)p * 2 + 0) * C + c];
    q += (double)partial[((int64_t)p * 2 + 1) * C + c];
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[c] = red[0][0];
    sums[C + c] = red[1][0];
  }
}

// ---- BatchNorm train-mode finalisation from global sums ---------------------------------------
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   int64_t* __restrict__ nbt, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  const double mean = sums[c] / count;
  double var = sums[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
  scale[c] = g * invstd;
  shift[c] = bb - (float)mean * g * invstd;
  if (mean_out) mean_out[c] = (float)mean;
  if (invstd_out) invstd_out[c] = invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// ---- BatchNorm eval-mode affine from the running statistics (decoders in validate_network, train.py:451) ----
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - rm[c] * g * is;
}

// ---- y = [relu](x*scale + shift) -> 16-bit --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int relu, T* __restrict__ out,
                                                     T* __restrict__ out_lo, int64_t R, int C, const float* __restrict__ mx_amax = nullptr) {
  const int cpt = C >> 2;
  const int64_t total = R * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[c], sh = reinterpret_cast<const float4*>(shift)[c];
    float a0 = v.x * sc.x + sh.x, a1 = v.y * sc.y + sh.y, a2 = v.z * sc.z + sh.z, a3 = v.w * sc.w + sh.w;
    if (relu) {
      a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); a2 = fmaxf(a2, 0.f); a3 = fmaxf(a3, 0.f);
    }
    uint2 o;
    o.x = pack2<T>(a0, a1);
    o.y = pack2<T>(a2, a3);
    reinterpret_cast<uint2*>(out)[i] = o;
    if (out_lo) {   // rounding residuals, or their MX form (asis_common.h) when the tensor's absolute maximum is given
      o.x = lo_word2<T>(a0, a1, mx_amax);
      o.y = lo_word2<T>(a2, a3, mx_amax);
      reinterpret_cast<uint2*>(out_lo)[i] = o;
    }
  }
}

// ---- BN + ReLU + MaxPool(3, stride 2, pad 1) ------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, T* __restrict__ out,
                                                              T* __restrict__ out_lo, int B, int H, int W, int OH,
                                                              int OW, int C) {
  const int cpt = C >> 2;
  const int64_t total = (int64_t)B * OH * OW * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt);
    const int64_t pix = i / cpt;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const int b = (int)(pix / ((int64_t)OW * OH));
    const float4 sc = reinterpret_cast<const float4*>(scale)[c], sh = reinterpret_cast<const float4*>(shift)[c];
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const float4 v = reinterpret_cast<const float4*>(x + (((int64_t)b * H + ih) * W + iw) * C)[c];
          m.x = fmaxf(m.x, fmaxf(v.x * sc.x + sh.x, 0.f));
          m.y = fmaxf(m.y, fmaxf(v.y * sc.y + sh.y, 0.f));
          m.z = fmaxf(m.z, fmaxf(v.z * sc.z + sh.z, 0.f));
          m.w = fmaxf(m.w, fmaxf(v.w * sc.w + sh.w, 0.f));
        }
      }
    uint2 o;
    o.x = pack2<T>(m.x, m.y);
    o.y = pack2<T>(m.z, m.w);
    reinterpret_cast<uint2*>(out + pix * C)[c] = o;
    if (out_lo) {
      o.x = pack2<T>(lo_part<T>(m.x), lo_part<T>(m.y));
      o.y = pack2<T>(lo_part<T>(m.z), lo_part<T>(m.w));
      reinterpret_cast<uint2*>(out_lo + pix * C)[c] = o;
    }
  }
}

// ---- BN + ReLU + bilinear upsample (align_corners=True), integer factor ---------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_upsample_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, T* __restrict__ out,
                                                               T* __restrict__ out_lo, const float* __restrict__ mx_amax, int B, int H, int W, int OH,
                                                               int OW, int C) {
  const int cpt = C >> 2;
  const float rh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const int64_t total = (int64_t)B * OH * OW * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt);
    const int64_t pix = i / cpt;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const int b = (int)(pix / ((int64_t)OW * OH));
    const float h1r = rh * oh, w1r = rw * ow;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < H - 1) ? 1 : 0, w1p = (w1 < W - 1) ? 1 : 0;
    const float hl = h1r - h1, wl = w1r - w1;
    const float4 sc = reinterpret_cast<const float4*>(scale)[c], sh = reinterpret_cast<const float4*>(shift)[c];
    const float* base = x + (((int64_t)b * H + h1) * W + w1) * C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int dh = (t >> 1) ? h1p : 0, dw = (t & 1) ? w1p : 0;
      const float wt = ((t >> 1) ? hl : 1.f - hl) * ((t & 1) ? wl : 1.f - wl);
      const float4 v = reinterpret_cast<const float4*>(base + ((int64_t)dh * W + dw) * C)[c];
      acc.x += wt * fmaxf(v.x * sc.x + sh.x, 0.f);
      acc.y += wt * fmaxf(v.y * sc.y + sh.y, 0.f);
      acc.z += wt * fmaxf(v.z * sc.z + sh.z, 0.f);
      acc.w += wt * fmaxf(v.w * sc.w + sh.w, 0.f);
    }
    uint2 o;
    o.x = pack2<T>(acc.x, acc.y);
    o.y = pack2<T>(acc.z, acc.w);
    reinterpret_cast<uint2*>(out + pix * C)[c] = o;
    if (out_lo) {
      o.x = lo_word2<T>(acc.x, acc.y, mx_amax);
      o.y = lo_word2<T>(acc.z, acc.w, mx_amax);
      reinterpret_cast<uint2*>(out_lo + pix * C)[c] = o;
    }
  }
}

// Same arithmetic (tap order, weights) with the index algebra taken out: grid = (column chunks, output rows, images), eight
// channels per thread -> no 64-bit division per element and 16-byte stores for both halves.  The generic kernel above spends
// most of its time on i / cpt, pix / OW, pix / (OW * OH) in 64 bits for 8 bytes of output (2.7 TB/s of writes where a bare
// 16-byte fill runs at 4.4 - 5.8 TB/s, scripts/ln_lab.hip).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_upsample8_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, T* __restrict__ out,
                                                                T* __restrict__ out_lo, int H, int W, int OH, int OW, int C,
                                                                const float* __restrict__ mx_amax) {
  constexpr int ROWS = 4;  // output rows per thread: 4x fewer waves to launch for the same bytes, source rows re-read from L1
  const int cp8 = C >> 3;
  // XCD-aware order: consecutive output rows re-read the same two source rows, so every XCD (one L2 each, blocks dealt
  // round-robin) gets a contiguous run of (image, row group, column chunk) instead of every eighth block
  const int gx = gridDim.x, gy = gridDim.y;
  const int lin = xcd_remap(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
  const int bx = lin % gx, og = (lin / gx) % gy, b = lin / (gx * gy);
  const int j = bx * 256 + threadIdx.x;
  if (j >= OW * cp8) return;
  const int ow = j / cp8, c8 = j - ow * cp8;
  const float rh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const float w1r = rw * ow;
  const int w1 = (int)w1r;
  const int w1p = (w1 < W - 1) ? 1 : 0;
  const float wl = w1r - w1;
  float4 sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sc[h] = reinterpret_cast<const float4*>(scale)[2 * c8 + h];
    sh[h] = reinterpret_cast<const float4*>(shift)[2 * c8 + h];
  }
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    const int oh = og * ROWS + rr;
    if (oh >= OH) break;
    const float h1r = rh * oh;
    const int h1 = (int)h1r;
    const int h1p = (h1 < H - 1) ? 1 : 0;
    const float hl = h1r - h1;
    const float* base = x + (((int64_t)b * H + h1) * W + w1) * C + 8 * c8;
    float4 v[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int dh = (t >> 1) ? h1p : 0, dw = (t & 1) ? w1p : 0;
      const float4* p = reinterpret_cast<const float4*>(base + ((int64_t)dh * W + dw) * C);
      v[t][0] = p[0];
      v[t][1] = p[1];
    }
    float acc[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float wt = ((t >> 1) ? hl : 1.f - hl) * ((t & 1) ? wl : 1.f - wl);
        a.x += wt * fmaxf(v[t][h].x * sc[h].x + sh[h].x, 0.f);
        a.y += wt * fmaxf(v[t][h].y * sc[h].y + sh[h].y, 0.f);
        a.z += wt * fmaxf(v[t][h].z * sc[h].z + sh[h].z, 0.f);
        a.w += wt * fmaxf(v[t][h].w * sc[h].w + sh[h].w, 0.f);
      }
      acc[4 * h + 0] = a.x; acc[4 * h + 1] = a.y; acc[4 * h + 2] = a.z; acc[4 * h + 3] = a.w;
    }
    const int64_t o8 = ((((int64_t)b * OH + oh) * OW + ow) * C >> 3) + c8;
    uint4 o;
    o.x = pack2<T>(acc[0], acc[1]); o.y = pack2<T>(acc[2], acc[3]);
    o.z = pack2<T>(acc[4], acc[5]); o.w = pack2<T>(acc[6], acc[7]);
    reinterpret_cast<uint4*>(out)[o8] = o;
    if (out_lo) {   // rounding residuals, or their MX form (asis_common.h) when the tensor's absolute maximum is given
      o.x = lo_word2<T>(acc[0], acc[1], mx_amax); o.y = lo_word2<T>(acc[2], acc[3], mx_amax);
      o.z = lo_word2<T>(acc[4], acc[5], mx_amax); o.w = lo_word2<T>(acc[6], acc[7], mx_amax);
      reinterpret_cast<uint4*>(out_lo)[o8] = o;
    }
  }
}

// ---- conv weight packing: fp32 [Cout,Cin,KH,KW] -> 16-bit GEMM B operand --------------------------
// mode 0 (forward):  out[co][(kh*KW+kw)*Cin + ci]                       rows Cout, ld = ldo
// mode 1 (dgrad):    out[ci][((KH-1-kh)*KW + (KW-1-kw))*CoP + co]       rows Cin,  CoP = Cout padded to 8
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ out,
                                                               int Cout, int Cin, int KH, int KW, int mode, int CoP,
                                                               int64_t ldo, int rows, int part, const float* __restrict__ mx_amax,
                                                               T* __restrict__ out2 = nullptr) {
  const int64_t total = (int64_t)rows * ldo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ldo);
    const int k = (int)(i - (int64_t)r * ldo);
    float v = 0.f;
    if (mode == 0) {
      if (k < KH * KW * Cin) {
        const int tap = k / Cin, ci = k - tap * Cin;
        const int kh = tap / KW, kw = tap - kh * KW;
        v = w[(((int64_t)r * Cin + ci) * KH + kh) * KW + kw];
      }
    } else {
      if (k < KH * KW * CoP) {
        const int tap = k / CoP, co = k - tap * CoP;
        const int kh = KH - 1 - tap / KW, kw = KW - 1 - (tap % KW);
        if (co < Cout) v = w[(((int64_t)co * Cin + r) * KH + kh) * KW + kw];
      }
    }
    if (out2) {        // pair form (round 5): the 16-bit weight AND its lo operand (residual, or MX when an amax is given) in one pass
      out[i] = to_t16<T>(v);
      if (mx_amax) {
        const uint32_t wd = mx_pack2<T>(v, 0.f, mx_scales<T>(*mx_amax), true);
        reinterpret_cast<uint16_t*>(out2)[i] = (uint16_t)(wd & 0xFFFFu);
      } else {
        out2[i] = to_t16<T>(lo_part<T>(v));
      }
    } else if (part == 2) {   // MX form of the lo operand, weight side: (lo8, hi8) in the element's two bytes
      const uint32_t wd = mx_pack2<T>(v, 0.f, mx_scales<T>(*mx_amax), true);
      reinterpret_cast<uint16_t*>(out)[i] = (uint16_t)(wd & 0xFFFFu);
    } else {
      out[i] = to_t16<T>(part ? lo_part<T>(v) : v);
    }
  }
}

// ---- decoder input: [adapter-stream tokens | zero-padded c4 | pass-A tokens] -> NHWC 16-bit -------
template <typename T>
__global__ __launch_bounds__(256) void decoder_input_kernel(const float* __restrict__ xs, int64_t xs_bstride,
                                                            const float* __restrict__ c4, int64_t c4_bstride,
                                                            const float* __restrict__ vit, int64_t vit_bstride,
                                                            T* __restrict__ out, T* __restrict__ out_lo, int B, int h,
                                                            int w, int h4, int w4, int D, const float* __restrict__ mx_amax) {
  const int cpt = (3 * D) >> 2;
  const int dq = D >> 2;
  const int py = (h - h4) / 2, px = (w - w4) / 2;  // F.pad([dx//2, dx-dx//2, dy//2, dy-dy//2])
  const int64_t total = (int64_t)B * h * w * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt);
    const int64_t pix = i / cpt;
    const int x = (int)(pix % w);
    const int y = (int)((pix / w) % h);
    const int b = (int)(pix / ((int64_t)w * h));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int seg = c / dq, cc = c - seg * dq;
    const int64_t pin = (int64_t)y * w + x;
    if (seg == 0) v = reinterpret_cast<const float4*>(xs + (int64_t)b * xs_bstride + pin * D)[cc];
    else if (seg == 2) v = reinterpret_cast<const float4*>(vit + (int64_t)b * vit_bstride + pin * D)[cc];
    else {
      const int yy = y - py, xx = x - px;
      if ((unsigned)yy < (unsigned)h4 && (unsigned)xx < (unsigned)w4)
        v = reinterpret_cast<const float4*>(c4 + (int64_t)b * c4_bstride + ((int64_t)yy * w4 + xx) * D)[cc];
    }
    uint2 o;
    o.x = pack2<T>(v.x, v.y);
    o.y = pack2<T>(v.z, v.w);
    reinterpret_cast<uint2*>(out + pix * 3 * D)[c] = o;
    if (out_lo) {
      o.x = lo_word2<T>(v.x, v.y, mx_amax);
      o.y = lo_word2<T>(v.z, v.w, mx_amax);
      reinterpret_cast<uint2*>(out_lo + pix * 3 * D)[c] = o;
    }
  }
}

// |x| maximum of an fp32 tensor [rows, cols] (row stride ld) -> atomicMax on the bit pattern (non-negative floats order like
// their bits); the caller zeroes *amax first (asis_absmax_f32 does, unless it accumulates over several tensors).
// ONE atomic per workgroup and at most ~1024 workgroups: same-address atomics retire one at a time in L2 (the first version's
// 16-24 k per-wave atomics cost 190-285 us whatever the tensor's size).
__device__ __forceinline__ void block_amax_commit(float m, float* __restrict__ amax) {
  __shared__ float red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(amax), __builtin_bit_cast(unsigned int, m));
  }
}
__device__ __forceinline__ float amax4(float m, const float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t rows, int cols, int64_t ld, float* __restrict__ amax) {
  const int c4 = cols >> 2;
  float m = 0.f;
  // grid.y strides the rows, grid.x the 16-byte chunks of a row: no division per element; four independent loads per trip from
  // clamped addresses (a chunk read twice does not change a maximum)
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float4* xr = reinterpret_cast<const float4*>(x + r * ld);
    const int step = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c4; i += 4 * step) {
      const float4 v0 = xr[i], v1 = xr[min(i + step, c4 - 1)], v2 = xr[min(i + 2 * step, c4 - 1)], v3 = xr[min(i + 3 * step, c4 - 1)];
      m = amax4(amax4(amax4(amax4(m, v0), v1), v2), v3);
    }
  }
  block_amax_commit(m, amax);
}
// |x| maximum of a 16-bit tensor [rows, cols] (row stride ld, cols % 8 == 0): the hi plane of a split-precision operand pair
template <typename T>
__global__ __launch_bounds__(256) void absmax16_kernel(const T* __restrict__ x, int64_t rows, int cols, int64_t ld, float* __restrict__ amax) {
  const int c8 = cols >> 3;
  float m = 0.f;
  auto one = [&](const uint4 v) {
    float a, b;
    unpack2<T>(v.x, a, b); m = fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
    unpack2<T>(v.y, a, b); m = fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
    unpack2<T>(v.z, a, b); m = fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
    unpack2<T>(v.w, a, b); m = fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
  };
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const uint4* xr = reinterpret_cast<const uint4*>(x + r * ld);
    const int step = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c8; i += 4 * step) {
      const uint4 v0 = xr[i], v1 = xr[min(i + step, c8 - 1)], v2 = xr[min(i + 2 * step, c8 - 1)], v3 = xr[min(i + 3 * step, c8 - 1)];
      one(v0); one(v1); one(v2); one(v3);
    }
  }
  block_amax_commit(m, amax);
}
// (hi, lo) 16-bit planes of a split-precision operand -> its MX plane (asis_common.h: two e4m3 bytes per element; activations
// (hi8, lo8), ``wside`` (lo8, hi8)); amax = the tensor's absolute maximum (device float)
template <typename T>
__global__ __launch_bounds__(256) void mx_from_pair_kernel(const T* __restrict__ hi, const T* __restrict__ lo, int64_t ld_in, T* __restrict__ out,
                                                           int64_t ld_out, int64_t rows, int cols, const float* __restrict__ amax, int wside) {
  const int c8 = cols >> 3;
  const MxScale sc = mx_scales<T>(*amax);
  const int64_t total = rows * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c8;
    const int c = (int)(i - r * c8);
    const uint4 h = reinterpret_cast<const uint4*>(hi + r * ld_in)[c], l = reinterpret_cast<const uint4*>(lo + r * ld_in)[c];
    uint4 o;
    float h0, h1, l0, l1;
    unpack2<T>(h.x, h0, h1); unpack2<T>(l.x, l0, l1); o.x = mx_pack2_pair(h0, l0, h1, l1, sc, wside != 0);
    unpack2<T>(h.y, h0, h1); unpack2<T>(l.y, l0, l1); o.y = mx_pack2_pair(h0, l0, h1, l1, sc, wside != 0);
    unpack2<T>(h.z, h0, h1); unpack2<T>(l.z, l0, l1); o.z = mx_pack2_pair(h0, l0, h1, l1, sc, wside != 0);
    unpack2<T>(h.w, h0, h1); unpack2<T>(l.w, l0, l1); o.w = mx_pack2_pair(h0, l0, h1, l1, sc, wside != 0);
    reinterpret_cast<uint4*>(out + r * ld_out)[c] = o;
  }
}
// the same of relu(x * scale[c] + shift[c]) (the tensor a BatchNorm + ReLU (+ bilinear upsampling: a convex combination) kernel
// is about to write), x fp32 [P, C].  The grid's stride is a multiple of C / 4 whenever C / 4 divides 256, so a thread stays on
// one channel group and keeps its scale / shift in registers.
__global__ __launch_bounds__(256) void bn_relu_absmax_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int64_t P, int C, int relu, float* __restrict__ amax) {
  const int c4 = C >> 2;
  const int64_t total = P * c4;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  auto one = [&](const float4 v, const float4 sc, const float4 sh) {
    float a0 = v.x * sc.x + sh.x, a1 = v.y * sc.y + sh.y, a2 = v.z * sc.z + sh.z, a3 = v.w * sc.w + sh.w;
    if (!relu) { a0 = fabsf(a0); a1 = fabsf(a1); a2 = fabsf(a2); a3 = fabsf(a3); }
    m = fmaxf(fmaxf(m, fmaxf(a0, a1)), fmaxf(a2, a3));
  };
  if (256 % c4 == 0) {
    const int c = (int)(i0 % c4);
    const float4 sc = reinterpret_cast<const float4*>(scale)[c], sh = reinterpret_cast<const float4*>(shift)[c];
    const float4* xv = reinterpret_cast<const float4*>(x);
    const int64_t last = i0 < total ? total - 1 - (total - 1 - i0) % step : 0;   // this thread's last element (same channel group)
    for (int64_t i = i0; i < total; i += 4 * step) {
      const float4 v0 = xv[i], v1 = xv[min(i + step, last)], v2 = xv[min(i + 2 * step, last)], v3 = xv[min(i + 3 * step, last)];
      one(v0, sc, sh); one(v1, sc, sh); one(v2, sc, sh); one(v3, sc, sh);
    }
  } else {
    for (int64_t i = i0; i < total; i += step) {
      const int c = (int)(i % c4);
      one(reinterpret_cast<const float4*>(x)[i], reinterpret_cast<const float4*>(scale)[c], reinterpret_cast<const float4*>(shift)[c]);
    }
  }
  block_amax_commit(m, amax);
}

// SwiGLU gate (dinov2/layers/swiglu_ffn.py:30-34): x12 fp32 [R, 2*Hd] -> silu(x1) * x2 as 16-bit [R, Hd]
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const float* __restrict__ x12, T* __restrict__ out, T* __restrict__ out_lo,
                                                     int64_t R, int Hd) {
  const int cpr = Hd >> 2;
  const int64_t total = R * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr);
    const float4 a = reinterpret_cast<const float4*>(x12 + r * 2 * Hd)[c];
    const float4 b = reinterpret_cast<const float4*>(x12 + r * 2 * Hd + Hd)[c];
    const float h0 = silu_mul(a.x, b.x), h1 = silu_mul(a.y, b.y);
    const float h2 = silu_mul(a.z, b.z), h3 = silu_mul(a.w, b.w);
    uint2 o;
    o.x = pack2<T>(h0, h1);
    o.y = pack2<T>(h2, h3);
    reinterpret_cast<uint2*>(out + r * Hd)[c] = o;
    if (out_lo) {     // rounding residual: the second half of a split-precision operand (config.precise_level 2)
      o.x = pack2<T>(lo_part<T>(h0), lo_part<T>(h1));
      o.y = pack2<T>(lo_part<T>(h2), lo_part<T>(h3));
      reinterpret_cast<uint2*>(out_lo + r * Hd)[c] = o;
    }
  }
}

// strided channel-slice copy (torch.cat / split along channels of NHWC tensors): 16-byte units
__global__ __launch_bounds__(256) void copy_channels_kernel(const uint4* __restrict__ src, int64_t src_ld, const uint4* unused,
                                                            uint4* __restrict__ dst, int64_t dst_ld, int64_t rows, int n16) {
  const int64_t total = rows * n16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / n16;
    const int c = (int)(i - r * n16);
    dst[r * dst_ld + c] = src[r * src_ld + c];
  }
}

// out[b][i] = a[b][i] + c[b][i] (fp32, float4), each operand with its own batch stride (in float4 units)
__global__ __launch_bounds__(256) void add_f32_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                      float4* __restrict__ out, int64_t n4, int batch, int64_t sa,
                                                      int64_t sb, int64_t so) {
  const int64_t total = n4 * batch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = i / n4, j = i - bi * n4;
    const float4 x = a[bi * sa + j], y = b[bi * sb + j];
    out[bi * so + j] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

}  // namespace

#define DT_OK(dtype, name) ASIS_REQUIRE((dtype) == ASIS_F16 || (dtype) == ASIS_BF16, name ": bad dtype %d", dtype)

extern "C" int asis_conv3x3_c3(void* stream, const float* img, const float* w, float* out, int B, int H, int W, int Cout,
                               int stride, int pad) {
  ASIS_REQUIRE(img && w && out, "asis_conv3x3_c3: null pointer");
  ASIS_REQUIRE(Cout % 4 == 0 && Cout > 0 && Cout <= 512, "asis_conv3x3_c3: Cout=%d must be a multiple of 4, <= 512", Cout);
  ASIS_REQUIRE(stride > 0 && pad >= 0 && H + 2 * pad >= 3 && W + 2 * pad >= 3, "asis_conv3x3_c3: bad geometry");
  ASIS_REQUIRE(asis_aligned16(out), "asis_conv3x3_c3: out must be 16-byte aligned");
  const int OH = (H + 2 * pad - 3) / stride + 1, OW = (W + 2 * pad - 3) / stride + 1;
  ASIS_REQUIRE(OH <= 65535 && B <= 65535, "asis_conv3x3_c3: OH=%d / B=%d exceed the grid", OH, B);
  const dim3 grid((unsigned)asis_cdiv((long)OW * (Cout / 4), 256), (unsigned)OH, (unsigned)B);
  hipLaunchKernelGGL(conv3x3_c3_kernel, grid, dim3(256), 27 * Cout * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), img, w, out, B, H, W, OH, OW, Cout, stride, pad);
  ASIS_CHECK_LAUNCH("asis_conv3x3_c3");
  return ASIS_OK;
}

extern "C" int asis_colstats_nparts(int64_t R) {
  int64_t n = asis_cdiv(R, 256);
  if (n > 2048) n = 2048;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int asis_colstats(void* stream, const float* x, int64_t R, int C, float* partial) {
  ASIS_REQUIRE(x && partial, "asis_colstats: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C > 0 && R > 0, "asis_colstats: bad shape R=%ld C=%d", (long)R, C);
  ASIS_REQUIRE(asis_aligned16(x) && asis_aligned16(partial), "asis_colstats: pointers must be 16-byte aligned");
  const int nparts = asis_colstats_nparts(R);
  const int64_t rpb = asis_cdiv(R, nparts);
  hipLaunchKernelGGL(colstats_kernel, dim3(nparts), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, R, C, rpb,
                     partial);
  ASIS_CHECK_LAUNCH("asis_colstats");
  return ASIS_OK;
}

extern "C" int asis_reduce_partials(void* stream, const float* partial, int nparts, int C, double* sums) {
  ASIS_REQUIRE(partial && sums && nparts > 0 && C > 0, "asis_reduce_partials: bad arguments");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial,
                     nparts, C, sums);
  ASIS_CHECK_LAUNCH("asis_reduce_partials");
  return ASIS_OK;
}

extern "C" int asis_bn_finalize(void* stream, const double* sums, double count, int C, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* scale, float* shift, float* mean_out,
                                float* invstd_out) {
  ASIS_REQUIRE(sums && scale && shift && C > 0 && count > 0, "asis_bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     sums, count, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, scale,
                     shift, mean_out, invstd_out);
  ASIS_CHECK_LAUNCH("asis_bn_finalize");
  return ASIS_OK;
}

extern "C" int asis_bn_eval_affine(void* stream, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, int C, float* scale, float* shift) {
  ASIS_REQUIRE(running_mean && running_var && scale && shift && C > 0, "asis_bn_eval_affine: bad arguments");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     gamma, beta, running_mean, running_var, eps, C, scale, shift);
  ASIS_CHECK_LAUNCH("asis_bn_eval_affine");
  return ASIS_OK;
}

static int bn_act_impl(void* stream, int dtype, const float* x, const float* scale, const float* shift, int relu, void* out, void* out_lo,
                       const float* mx_amax, int64_t R, int C) {
  ASIS_REQUIRE(x && scale && shift && out, "asis_bn_act: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C > 0, "asis_bn_act: C=%d must be a multiple of 4", C);
  DT_OK(dtype, "asis_bn_act");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = R * (C / 4);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((bn_act_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, x, scale, shift, relu,
                       reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), R, C, mx_amax);
  else
    hipLaunchKernelGGL((bn_act_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, x, scale, shift, relu,
                       reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(out_lo), R, C, mx_amax);
  ASIS_CHECK_LAUNCH("asis_bn_act");
  return ASIS_OK;
}
extern "C" int asis_bn_act(void* stream, int dtype, const float* x, const float* scale, const float* shift, int relu,
                           void* out, void* out_lo, int64_t R, int C) {
  return bn_act_impl(stream, dtype, x, scale, shift, relu, out, out_lo, nullptr, R, C);
}
extern "C" int asis_bn_act_mx(void* stream, int dtype, const float* x, const float* scale, const float* shift, int relu, void* out,
                              void* out_mx, const float* amax, int64_t R, int C) {
  ASIS_REQUIRE(out_mx && amax, "asis_bn_act_mx: null pointer");
  return bn_act_impl(stream, dtype, x, scale, shift, relu, out, out_mx, amax, R, C);
}

extern "C" int asis_bn_relu_maxpool(void* stream, int dtype, const float* x, const float* scale, const float* shift,
                                    void* out, void* out_lo, int B, int H, int W, int C) {
  ASIS_REQUIRE(x && scale && shift && out, "asis_bn_relu_maxpool: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C > 0, "asis_bn_relu_maxpool: C=%d must be a multiple of 4", C);
  DT_OK(dtype, "asis_bn_relu_maxpool");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)B * OH * OW * (C / 4);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((bn_relu_maxpool_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, x, scale, shift,
                       reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), B, H, W, OH, OW, C);
  else
    hipLaunchKernelGGL((bn_relu_maxpool_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, x, scale, shift,
                       reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(out_lo), B, H, W, OH, OW, C);
  ASIS_CHECK_LAUNCH("asis_bn_relu_maxpool");
  return ASIS_OK;
}

static int bn_relu_upsample_impl(void* stream, int dtype, const float* x, const float* scale, const float* shift, void* out, void* out_lo,
                                 const float* mx_amax, int B, int H, int W, int C, int factor) {
  ASIS_REQUIRE(x && scale && shift && out, "asis_bn_relu_upsample: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && C > 0 && factor >= 1, "asis_bn_relu_upsample: bad C=%d / factor=%d", C, factor);
  DT_OK(dtype, "asis_bn_relu_upsample");
  const int OH = H * factor, OW = W * factor;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)B * OH * OW * (C / 4);
  const long gx8 = asis_cdiv((long)OW * (C / 8), 256);
  if (C % 8 == 0 && OH <= 65535 && B <= 65535 && gx8 * OH * B < (1L << 31) && asis_aligned16(out) &&
      (!out_lo || asis_aligned16(out_lo)) && asis_aligned16(x)) {
    const dim3 grid8((unsigned)gx8, (unsigned)asis_cdiv(OH, 4), (unsigned)B);  // 4 = ROWS of the kernel
    if (dtype == ASIS_F16)
      hipLaunchKernelGGL((bn_relu_upsample8_kernel<f16>), grid8, dim3(256), 0, s, x, scale, shift, reinterpret_cast<f16*>(out),
                         reinterpret_cast<f16*>(out_lo), H, W, OH, OW, C, mx_amax);
    else
      hipLaunchKernelGGL((bn_relu_upsample8_kernel<bf16>), grid8, dim3(256), 0, s, x, scale, shift, reinterpret_cast<bf16*>(out),
                         reinterpret_cast<bf16*>(out_lo), H, W, OH, OW, C, mx_amax);
    ASIS_CHECK_LAUNCH("asis_bn_relu_upsample");
    return ASIS_OK;
  }
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((bn_relu_upsample_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, x, scale, shift,
                       reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), mx_amax, B, H, W, OH, OW, C);
  else
    hipLaunchKernelGGL((bn_relu_upsample_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, x, scale, shift,
                       reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(out_lo), mx_amax, B, H, W, OH, OW, C);
  ASIS_CHECK_LAUNCH("asis_bn_relu_upsample");
  return ASIS_OK;
}

extern "C" int asis_bn_relu_upsample(void* stream, int dtype, const float* x, const float* scale, const float* shift,
                                     void* out, void* out_lo, int B, int H, int W, int C, int factor) {
  return bn_relu_upsample_impl(stream, dtype, x, scale, shift, out, out_lo, nullptr, B, H, W, C, factor);
}
extern "C" int asis_bn_relu_upsample_mx(void* stream, int dtype, const float* x, const float* scale, const float* shift,
                                        void* out, void* out_mx, const float* amax, int B, int H, int W, int C, int factor) {
  ASIS_REQUIRE(out_mx && amax, "asis_bn_relu_upsample_mx: null pointer");
  return bn_relu_upsample_impl(stream, dtype, x, scale, shift, out, out_mx, amax, B, H, W, C, factor);
}

extern "C" int asis_absmax_f32(void* stream, const float* x, int64_t rows, int cols, int64_t ld, float* amax, int reset) {
  ASIS_REQUIRE(x && amax && rows >= 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ld >= cols && asis_aligned16(x),
               "asis_absmax_f32: null pointer, or cols / ld not multiples of 4, or misaligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (reset) ASIS_REQUIRE(hipMemsetAsync(amax, 0, sizeof(float), s) == hipSuccess, "asis_absmax_f32: memset failed");
  if (rows == 0) return ASIS_OK;
  // ~1024 workgroups (one atomic each): gx over a row's chunks (four per thread and trip), gy over rows
  int gx = (int)asis_cdiv(cols / 4, 256 * 4);
  if (gx > 1024) gx = 1024;
  int64_t gy = asis_cdiv(1024, gx);
  if (gy > rows) gy = rows;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, x, rows, cols, ld, amax);
  ASIS_CHECK_LAUNCH("asis_absmax_f32");
  return ASIS_OK;
}

extern "C" int asis_bn_relu_absmax(void* stream, const float* x, const float* scale, const float* shift, int64_t P, int C, int relu,
                                   float* amax) {
  ASIS_REQUIRE(x && scale && shift && amax && P >= 0 && C > 0 && C % 4 == 0 && asis_aligned16(x) && asis_aligned16(scale) && asis_aligned16(shift),
               "asis_bn_relu_absmax: null pointer, C not a multiple of 4, or misaligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ASIS_REQUIRE(hipMemsetAsync(amax, 0, sizeof(float), s) == hipSuccess, "asis_bn_relu_absmax: memset failed");
  if (P == 0) return ASIS_OK;
  hipLaunchKernelGGL(bn_relu_absmax_kernel, dim3(grid_for(asis_cdiv(P * (C / 4), 4), 256, 1024)), dim3(256), 0, s, x, scale, shift, P, C, relu, amax);
  ASIS_CHECK_LAUNCH("asis_bn_relu_absmax");
  return ASIS_OK;
}

extern "C" int asis_absmax_16(void* stream, int dtype, const void* x, int64_t rows, int cols, int64_t ld, float* amax) {
  ASIS_REQUIRE(x && amax && rows >= 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols && asis_aligned16(x),
               "asis_absmax_16: null pointer, or cols / ld not multiples of 8, or misaligned");
  DT_OK(dtype, "asis_absmax_16");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ASIS_REQUIRE(hipMemsetAsync(amax, 0, sizeof(float), s) == hipSuccess, "asis_absmax_16: memset failed");
  if (rows == 0) return ASIS_OK;
  int gx = (int)asis_cdiv(cols / 8, 256 * 4);
  if (gx > 1024) gx = 1024;
  int64_t gy = asis_cdiv(1024, gx);
  if (gy > rows) gy = rows;
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((absmax16_kernel<f16>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, reinterpret_cast<const f16*>(x), rows, cols, ld, amax);
  else
    hipLaunchKernelGGL((absmax16_kernel<bf16>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, reinterpret_cast<const bf16*>(x), rows, cols, ld, amax);
  ASIS_CHECK_LAUNCH("asis_absmax_16");
  return ASIS_OK;
}

extern "C" int asis_mx_from_pair(void* stream, int dtype, const void* hi, const void* lo, int64_t ld_in, void* out_mx, int64_t ld_out,
                                 int64_t rows, int cols, const float* amax, int wside) {
  ASIS_REQUIRE(hi && lo && out_mx && amax && rows >= 0 && cols > 0 && cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ld_in >= cols &&
                   ld_out >= cols && asis_aligned16(hi) && asis_aligned16(lo) && asis_aligned16(out_mx),
               "asis_mx_from_pair: null pointer, or cols / leading dimensions not multiples of 8, or misaligned");
  DT_OK(dtype, "asis_mx_from_pair");
  if (rows == 0) return ASIS_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = rows * (cols / 8);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((mx_from_pair_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, reinterpret_cast<const f16*>(hi),
                       reinterpret_cast<const f16*>(lo), ld_in, reinterpret_cast<f16*>(out_mx), ld_out, rows, cols, amax, wside);
  else
    hipLaunchKernelGGL((mx_from_pair_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, reinterpret_cast<const bf16*>(hi),
                       reinterpret_cast<const bf16*>(lo), ld_in, reinterpret_cast<bf16*>(out_mx), ld_out, rows, cols, amax, wside);
  ASIS_CHECK_LAUNCH("asis_mx_from_pair");
  return ASIS_OK;
}

static int pack_conv_weight_impl(void* stream, int dtype, const float* w, void* out, int Cout, int Cin, int KH, int KW, int mode,
                                 int64_t ldo, int part, const float* mx_amax) {
  ASIS_REQUIRE(w && out, "asis_pack_conv_weight: null pointer");
  ASIS_REQUIRE(part >= 0 && part <= 2 && (part != 2 || mx_amax), "asis_pack_conv_weight: part 2 (MX weight side) needs amax");
  ASIS_REQUIRE(mode == 0 || mode == 1, "asis_pack_conv_weight: bad mode %d", mode);
  DT_OK(dtype, "asis_pack_conv_weight");
  const int CoP = (Cout + 7) / 8 * 8;
  const int rows = mode == 0 ? Cout : Cin;
  const int K = mode == 0 ? KH * KW * Cin : KH * KW * CoP;
  ASIS_REQUIRE(ldo >= K && ldo % 8 == 0, "asis_pack_conv_weight: ldo=%ld must be a multiple of 8 and >= %d", (long)ldo, K);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)rows * ldo;
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((pack_conv_weight_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, w,
                       reinterpret_cast<f16*>(out), Cout, Cin, KH, KW, mode, CoP, ldo, rows, part, mx_amax);
  else
    hipLaunchKernelGGL((pack_conv_weight_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, w,
                       reinterpret_cast<bf16*>(out), Cout, Cin, KH, KW, mode, CoP, ldo, rows, part, mx_amax);
  ASIS_CHECK_LAUNCH("asis_pack_conv_weight");
  return ASIS_OK;
}

// Tiled form of the pair pack for 3x3 weights (round 5: the per-element gather above read 4 bytes at a 36-byte stride and ran
// the UNet's 250 M parameters at a third of the HBM rate): contiguous fp32 runs in, through LDS, contiguous 16-bit runs out.
//   mode 0: one block per (co, 128 input channels): 1152 contiguous floats -> 9 runs of 128 (x 2 planes)
//   mode 1: one block per (64 output channels, 16 input channels): 64 runs of 144 floats -> 144 runs of 64 (rows padded to 145
//           floats in LDS: the column reads are 2-way conflicted at worst)
template <typename T>
__global__ __launch_bounds__(256) void pack_conv3x3_pair_tiled_kernel(const float* __restrict__ w, T* __restrict__ out, T* __restrict__ out2,
                                                                      int Cout, int Cin, int mode, int64_t ldo,
                                                                      const float* __restrict__ mx_amax) {
  __shared__ float tile[64 * 145];
  const int tid = threadIdx.x;
  uint32_t* const o1 = reinterpret_cast<uint32_t*>(out);
  uint32_t* const o2 = reinterpret_cast<uint32_t*>(out2);
  if (mode == 0) {
    const int chunks = Cin >> 7;
    const int co = blockIdx.x / chunks, ci0 = (blockIdx.x - co * chunks) << 7;
    const float4* src = reinterpret_cast<const float4*>(w + ((int64_t)co * Cin + ci0) * 9);
    for (int i = tid; i < 288; i += 256) reinterpret_cast<float4*>(tile)[i] = src[i];
    __syncthreads();
    for (int p = tid; p < 576; p += 256) {
      const int tap = p >> 6, c = (p & 63) << 1;
      const float v0 = tile[c * 9 + tap], v1 = tile[(c + 1) * 9 + tap];
      const int64_t idx = ((int64_t)co * ldo + (int64_t)tap * Cin + ci0 + c) >> 1;
      o1[idx] = pack2<T>(v0, v1);
      o2[idx] = lo_word2<T>(v0, v1, mx_amax, true);
    }
  } else {
    const int chunks = Cin >> 4;
    const int cb = blockIdx.x / chunks, ci0 = (blockIdx.x - cb * chunks) << 4, co0 = cb << 6;
    for (int i = tid; i < 64 * 36; i += 256) {
      const int co = i / 36, j = i - co * 36;
      const float4 v = reinterpret_cast<const float4*>(w + ((int64_t)(co0 + co) * Cin + ci0) * 9)[j];
      float* t = tile + co * 145 + 4 * j;
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
    const int pair = tid & 31, slot = tid >> 5;
    for (int row = slot; row < 144; row += 8) {
      const int ci = row / 9, tapf = row - ci * 9;       // out[ci][(2 - kh) * 3 + (2 - kw)][co] = w[co][ci][kh][kw]
      const int k = ci * 9 + (8 - tapf);
      const float v0 = tile[(2 * pair) * 145 + k], v1 = tile[(2 * pair + 1) * 145 + k];
      const int64_t idx = ((int64_t)(ci0 + ci) * ldo + (int64_t)tapf * Cout + co0 + 2 * pair) >> 1;
      o1[idx] = pack2<T>(v0, v1);
      o2[idx] = lo_word2<T>(v0, v1, mx_amax, true);
    }
  }
}

extern "C" int asis_pack_conv_weight_pair(void* stream, int dtype, const float* w, void* out_hi, void* out_lo, int Cout, int Cin,
                                          int KH, int KW, int mode, int64_t ldo, const float* amax) {
  ASIS_REQUIRE(w && out_hi && out_lo, "asis_pack_conv_weight_pair: null pointer");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_pack_conv_weight_pair: bad dtype %d", dtype);
  ASIS_REQUIRE(mode == 0 || mode == 1, "asis_pack_conv_weight_pair: mode must be 0 (forward) or 1 (dgrad)");
  const int CoP = (Cout + 7) / 8 * 8;
  const int rows = mode == 0 ? Cout : Cin;
  const int64_t K = mode == 0 ? (int64_t)KH * KW * Cin : (int64_t)KH * KW * CoP;
  ASIS_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && ldo >= K, "asis_pack_conv_weight_pair: bad shape / ldo");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)rows * ldo;
  static const int tiled_on = [] { const char* e = getenv("ASIS_PACK_TILED"); return e ? atoi(e) : 1; }();
  const bool tiled = tiled_on && KH == 3 && KW == 3 && ldo == K && ldo % 2 == 0 && asis_aligned16(w) &&
                     (reinterpret_cast<uintptr_t>(out_hi) & 3) == 0 && (reinterpret_cast<uintptr_t>(out_lo) & 3) == 0 &&
                     (mode == 0 ? Cin % 128 == 0 : (Cout % 64 == 0 && Cin % 16 == 0));
  if (tiled) {
    const int64_t nblk = mode == 0 ? (int64_t)Cout * (Cin / 128) : (int64_t)(Cout / 64) * (Cin / 16);
    ASIS_REQUIRE(nblk < (1ll << 31), "asis_pack_conv_weight_pair: too many tiles");
    if (dtype == ASIS_F16)
      hipLaunchKernelGGL((pack_conv3x3_pair_tiled_kernel<f16>), dim3((unsigned)nblk), dim3(256), 0, s, w, reinterpret_cast<f16*>(out_hi),
                         reinterpret_cast<f16*>(out_lo), Cout, Cin, mode, ldo, amax);
    else
      hipLaunchKernelGGL((pack_conv3x3_pair_tiled_kernel<bf16>), dim3((unsigned)nblk), dim3(256), 0, s, w, reinterpret_cast<bf16*>(out_hi),
                         reinterpret_cast<bf16*>(out_lo), Cout, Cin, mode, ldo, amax);
    ASIS_CHECK_LAUNCH("asis_pack_conv_weight_pair");
    return ASIS_OK;
  }
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((pack_conv_weight_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, w, reinterpret_cast<f16*>(out_hi), Cout,
                       Cin, KH, KW, mode, CoP, ldo, rows, 0, amax, reinterpret_cast<f16*>(out_lo));
  else
    hipLaunchKernelGGL((pack_conv_weight_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, w, reinterpret_cast<bf16*>(out_hi), Cout,
                       Cin, KH, KW, mode, CoP, ldo, rows, 0, amax, reinterpret_cast<bf16*>(out_lo));
  ASIS_CHECK_LAUNCH("asis_pack_conv_weight_pair");
  return ASIS_OK;
}

extern "C" int asis_pack_conv_weight(void* stream, int dtype, const float* w, void* out, int Cout, int Cin, int KH,
                                     int KW, int mode, int64_t ldo, int part) {
  ASIS_REQUIRE(part == 0 || part == 1, "asis_pack_conv_weight: part must be 0 or 1");
  return pack_conv_weight_impl(stream, dtype, w, out, Cout, Cin, KH, KW, mode, ldo, part, nullptr);
}
extern "C" int asis_pack_conv_weight_mx(void* stream, int dtype, const float* w, void* out_mx, int Cout, int Cin, int KH,
                                        int KW, int mode, int64_t ldo, const float* amax) {
  return pack_conv_weight_impl(stream, dtype, w, out_mx, Cout, Cin, KH, KW, mode, ldo, 2, amax);
}

static int decoder_input_impl(void* stream, int dtype, const float* xs, int64_t xs_bstride, const float* c4,
                              int64_t c4_bstride, const float* vit, int64_t vit_bstride, void* out, void* out_lo,
                              int B, int h, int w, int h4, int w4, int D, const float* mx_amax) {
  ASIS_REQUIRE(xs && c4 && vit && out, "asis_decoder_input: null pointer");
  ASIS_REQUIRE(D % 4 == 0 && h4 <= h && w4 <= w, "asis_decoder_input: bad shape");
  ASIS_REQUIRE(c4_bstride % 4 == 0 && xs_bstride % 4 == 0 && vit_bstride % 4 == 0 && asis_aligned16(xs) &&
                   asis_aligned16(c4) && asis_aligned16(vit), "asis_decoder_input: alignment");
  DT_OK(dtype, "asis_decoder_input");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)B * h * w * (3 * D / 4);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((decoder_input_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, xs, xs_bstride, c4, c4_bstride, vit,
                       vit_bstride, reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), B, h, w, h4, w4, D, mx_amax);
  else
    hipLaunchKernelGGL((decoder_input_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, xs, xs_bstride, c4, c4_bstride, vit,
                       vit_bstride, reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(out_lo), B, h, w, h4, w4, D, mx_amax);
  ASIS_CHECK_LAUNCH("asis_decoder_input");
  return ASIS_OK;
}

extern "C" int asis_decoder_input(void* stream, int dtype, const float* xs, int64_t xs_bstride, const float* c4,
                                  int64_t c4_bstride, const float* vit, int64_t vit_bstride, void* out, void* out_lo,
                                  int B, int h, int w, int h4, int w4, int D) {
  return decoder_input_impl(stream, dtype, xs, xs_bstride, c4, c4_bstride, vit, vit_bstride, out, out_lo, B, h, w, h4, w4, D, nullptr);
}
extern "C" int asis_decoder_input_mx(void* stream, int dtype, const float* xs, int64_t xs_bstride, const float* c4,
                                     int64_t c4_bstride, const float* vit, int64_t vit_bstride, void* out, void* out_mx,
                                     const float* amax, int B, int h, int w, int h4, int w4, int D) {
  ASIS_REQUIRE(out_mx && amax, "asis_decoder_input_mx: null pointer");
  return decoder_input_impl(stream, dtype, xs, xs_bstride, c4, c4_bstride, vit, vit_bstride, out, out_mx, B, h, w, h4, w4, D, amax);
}

extern "C" int asis_swiglu_split(void* stream, int dtype, const float* x12, void* out, void* out_lo, int64_t R, int Hd) {
  ASIS_REQUIRE(x12 && out && Hd % 4 == 0 && Hd > 0, "asis_swiglu: bad arguments");
  ASIS_REQUIRE(asis_aligned16(x12) && (((uintptr_t)out) & 7) == 0 && (((uintptr_t)out_lo) & 7) == 0, "asis_swiglu: alignment");
  DT_OK(dtype, "asis_swiglu");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = R * (Hd / 4);
  if (dtype == ASIS_F16) hipLaunchKernelGGL((swiglu_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, x12, reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), R, Hd);
  else hipLaunchKernelGGL((swiglu_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, x12, reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(out_lo), R, Hd);
  ASIS_CHECK_LAUNCH("asis_swiglu");
  return ASIS_OK;
}

extern "C" int asis_swiglu(void* stream, int dtype, const float* x12, void* out, int64_t R, int Hd) {
  return asis_swiglu_split(stream, dtype, x12, out, nullptr, R, Hd);
}

extern "C" int asis_copy_channels(void* stream, const void* src, int64_t src_ld_bytes, void* dst, int64_t dst_ld_bytes,
                                  int64_t rows, int64_t row_bytes) {
  ASIS_REQUIRE(src && dst, "asis_copy_channels: null pointer");
  ASIS_REQUIRE(row_bytes % 16 == 0 && src_ld_bytes % 16 == 0 && dst_ld_bytes % 16 == 0 && asis_aligned16(src) && asis_aligned16(dst),
               "asis_copy_channels: everything must be 16-byte granular");
  const int64_t total = rows * (row_bytes / 16);
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint4*>(src), src_ld_bytes / 16, (const uint4*)nullptr, reinterpret_cast<uint4*>(dst),
                     dst_ld_bytes / 16, rows, (int)(row_bytes / 16));
  ASIS_CHECK_LAUNCH("asis_copy_channels");
  return ASIS_OK;
}

extern "C" int asis_add_f32(void* stream, const float* a, const float* b, float* out, int64_t n, int batch,
                            int64_t stride_a, int64_t stride_b, int64_t stride_out) {
  ASIS_REQUIRE(a && b && out, "asis_add_f32: null pointer");
  ASIS_REQUIRE(n % 4 == 0 && batch >= 1 && stride_a % 4 == 0 && stride_b % 4 == 0 && stride_out % 4 == 0 &&
                   asis_aligned16(a) && asis_aligned16(b) && asis_aligned16(out), "asis_add_f32: alignment");
  hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n / 4 * batch)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                     reinterpret_cast<float4*>(out), n / 4, batch, stride_a / 4, stride_b / 4, stride_out / 4);
  ASIS_CHECK_LAUNCH("asis_add_f32");
  return ASIS_OK;
}
