// Data-movement kernels of the UNet decode head (backbones/unet_parts.py): MaxPool2d(2) forward / backward on
// split-precision NHWC maps and the 2x2 stride-2 pixel shuffle that turns ConvTranspose2d(k=2, s=2) into one
// MFMA GEMM:
//     y[b, 2i+di, 2j+dj, co] = bias[co] + sum_ci x[b, i, j, ci] * w[ci, co, di, dj]
//  => G[p, n] = X[p, :] . Wt[n, :]  with p = (b, i, j), n = co*4 + di*2 + dj   (asis_gemm, N = 4*Cout)
//     forward : asis_convt2x2_scatter  G (fp32) -> the 16-bit (hi, lo) operand of the next conv, written straight
//               into the channel-concat buffer of Up (unet_parts.py:54-63: F.pad + torch.cat([x2, x1])).
//     backward: asis_convt2x2_gather   d cat (fp32) -> dG (hi, lo) [P, 4*Cout] for the dgrad GEMM (dX = dG . W) and
//               the weight gradient (dW[ci, n] = sum_p x[p, ci] dG[p, n], asis_wgrad with a 1x1 geometry);
//               asis_convt2x2_bias_grad: per-channel sums of the same slice.
// All of it is HBM-bound streaming: 16-byte lanes, channel-fastest indexing, no atomics.
#include "asis_common.h"

namespace {

inline int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  if (g > 65535 * 4) g = 65535 * 4;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- MaxPool2d(2) (floor mode: an odd last row / column is dropped, unet_parts.py:31) ------------------------------
// value of a pixel = hi + lo (fp32); the first maximum in scan order (0,0),(0,1),(1,0),(1,1) wins like ATen's kernel.
// idx[b, oh, ow, c] in 0..3 feeds the backward.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, const T* __restrict__ x_lo,
                                                           T* __restrict__ out, T* __restrict__ out_lo,
                                                           uint8_t* __restrict__ idx, int B, int H, int W, int OH, int OW,
                                                           int C) {
  typedef typename T16<T>::v8 v8;
  const int cpt = C >> 3;
  const int64_t total = (int64_t)B * OH * OW * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt) << 3;
    const int64_t pix = i / cpt;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const int b = (int)(pix / ((int64_t)OW * OH));
    float best[8];
    v8 bh, bl;
    uint8_t bi[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t off = (((int64_t)b * H + (oh * 2 + (q >> 1))) * W + (ow * 2 + (q & 1))) * C + c;
      const v8 h = *reinterpret_cast<const v8*>(x + off);
      v8 l;
      if (x_lo) l = *reinterpret_cast<const v8*>(x_lo + off);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = (float)h[k] + (x_lo ? (float)l[k] : 0.f);
        if (q == 0 || v > best[k]) {
          best[k] = v;
          bh[k] = h[k];
          if (x_lo) bl[k] = l[k];
          bi[k] = (uint8_t)q;
        }
      }
    }
    const int64_t o = pix * C + c;
    *reinterpret_cast<v8*>(out + o) = bh;
    if (out_lo) *reinterpret_cast<v8*>(out_lo + o) = bl;
    if (idx) {
      uint2 w;
      w.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((uint32_t)bi[3] << 24);
      w.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((uint32_t)bi[7] << 24);
      *reinterpret_cast<uint2*>(idx + o) = w;
    }
  }
}

// dx[b, 2oh+di, 2ow+dj, c] += dy[b, oh, ow, c] where (di, dj) == idx  (dx already holds the other gradient path of
// the pooled tensor — the skip connection — or zeros)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                           float* __restrict__ dx, int B, int H, int W, int OH, int OW,
                                                           int C) {
  const int cpt = C >> 2;
  const int64_t total = (int64_t)B * OH * OW * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt) << 2;
    const int64_t pix = i / cpt;
    const int ow = (int)(pix % OW);
    const int oh = (int)((pix / OW) % OH);
    const int b = (int)(pix / ((int64_t)OW * OH));
    const float4 g = *reinterpret_cast<const float4*>(dy + pix * C + c);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(idx + pix * C + c);
    const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* p = dx + (((int64_t)b * H + (oh * 2 + (q >> 1))) * W + (ow * 2 + (q & 1))) * C + c;
      float4 v = *reinterpret_cast<float4*>(p);
      float* vv = reinterpret_cast<float*>(&v);
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (((w >> (8 * k)) & 0xff) == (uint32_t)q) {
          vv[k] += gv[k];
          any = true;
        }
      if (any) *reinterpret_cast<float4*>(p) = v;
    }
  }
}

// ---- ConvTranspose2d(k=2, s=2) pixel shuffle -----------------------------------------------------------------------
// G fp32 [P, 4*Cout] (n = co*4 + di*2 + dj) -> dst (hi, lo) [B, H2, W2, Ctot] at channels [coff, coff+Cout) and pixels
// (padT + 2i + di, padL + 2j + dj).  One thread: one source pixel x 8 output channels (128 B read, 4 x 16 B written).
template <typename T>
__global__ __launch_bounds__(256) void convt2x2_scatter_kernel(const float* __restrict__ G, T* __restrict__ dst,
                                                               T* __restrict__ dst_lo, int B, int H, int W, int Cout,
                                                               int H2, int W2, int Ctot, int coff, int padT, int padL) {
  typedef typename T16<T>::v8 v8;
  const int cpt = Cout >> 3;
  const int64_t total = (int64_t)B * H * W * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt) << 3;
    const int64_t p = i / cpt;
    const int j = (int)(p % W);
    const int ii = (int)((p / W) % H);
    const int b = (int)(p / ((int64_t)W * H));
    float v[32];
    const float4* src = reinterpret_cast<const float4*>(G + p * 4 * Cout + c * 4);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 t = src[k];
      v[k * 4 + 0] = t.x; v[k * 4 + 1] = t.y; v[k * 4 + 2] = t.z; v[k * 4 + 3] = t.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v8 h, l;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float f = v[k * 4 + q];
        h[k] = to_t16<T>(f);
        l[k] = to_t16<T>(lo_part<T>(f));
      }
      const int64_t o = (((int64_t)b * H2 + (padT + 2 * ii + (q >> 1))) * W2 + (padL + 2 * j + (q & 1))) * Ctot + coff + c;
      *reinterpret_cast<v8*>(dst + o) = h;
      if (dst_lo) *reinterpret_cast<v8*>(dst_lo + o) = l;
    }
  }
}

// the transpose: d cat fp32 [B, H2, W2, Ctot] -> dG (hi, lo) [P, 4*Cout]
template <typename T>
__global__ __launch_bounds__(256) void convt2x2_gather_kernel(const float* __restrict__ dcat, T* __restrict__ dG,
                                                              T* __restrict__ dG_lo, int B, int H, int W, int Cout, int H2,
                                                              int W2, int Ctot, int coff, int padT, int padL) {
  typedef typename T16<T>::v8 v8;
  const int cpt = Cout >> 3;
  const int64_t total = (int64_t)B * H * W * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt) << 3;
    const int64_t p = i / cpt;
    const int j = (int)(p % W);
    const int ii = (int)((p / W) % H);
    const int b = (int)(p / ((int64_t)W * H));
    float v[32];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4* s = reinterpret_cast<const float4*>(
          dcat + (((int64_t)b * H2 + (padT + 2 * ii + (q >> 1))) * W2 + (padL + 2 * j + (q & 1))) * Ctot + coff + c);
      const float4 t0 = s[0], t1 = s[1];
      v[0 * 4 + q] = t0.x; v[1 * 4 + q] = t0.y; v[2 * 4 + q] = t0.z; v[3 * 4 + q] = t0.w;
      v[4 * 4 + q] = t1.x; v[5 * 4 + q] = t1.y; v[6 * 4 + q] = t1.z; v[7 * 4 + q] = t1.w;
    }
    T* o = dG + p * 4 * Cout + c * 4;
    T* ol = dG_lo ? dG_lo + p * 4 * Cout + c * 4 : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        h[e] = to_t16<T>(v[k * 8 + e]);
        l[e] = to_t16<T>(lo_part<T>(v[k * 8 + e]));
      }
      reinterpret_cast<v8*>(o)[k] = h;
      if (ol) reinterpret_cast<v8*>(ol)[k] = l;
    }
  }
}

// partial[blk][co] = sum over this block's rows of the pixels (padT + y, padL + x), y < 2H, x < 2W, of
// dcat[..., coff + co]: the ConvTranspose2d bias gradient.  64 / 32 / 16 float4 channel lanes (by Cout) x 4 / 8 / 16 row lanes
// per block: a wave reads whole contiguous channel runs, four rows in flight per thread; the offsets of 256 rows at a time go
// through LDS (one index division per row, not per element), and the row lanes are folded through LDS in a fixed order.
// (Round 5: the one-thread-per-channel form — a 64-bit division and ONE 4-byte load in flight per iteration — took 505 us per
// call, 3 % of the config-2 step.)
__global__ __launch_bounds__(256) void convt2x2_bias_grad_kernel(const float* __restrict__ dcat, float* __restrict__ partial,
                                                                 int B, int OH, int OW, int Cout, int H2, int W2, int Ctot,
                                                                 int coff, int padT, int padL, int rows_per_blk) {
  __shared__ int64_t roff[256];
  __shared__ float4 red[256];
  const int64_t rows = (int64_t)B * OH * OW;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk;
  if (r1 > rows) r1 = rows;
  const int c4n = Cout >> 2;
  const int clog = c4n >= 64 ? 6 : (c4n >= 32 ? 5 : 4);     // channel lanes 64 / 32 / 16 -> row lanes 4 / 8 / 16
  const int cl = 1 << clog, rl = 256 >> clog;
  const int cx = threadIdx.x & (cl - 1), ry = threadIdx.x >> clog;
  for (int cbase = 0; cbase < c4n; cbase += cl) {
    const int c4 = cbase + cx;
    const bool cok = c4 < c4n;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t rb = r0; rb < r1; rb += 256) {
      __syncthreads();
      {
        const int64_t r = rb + threadIdx.x;
        if (r < r1) {
          const int x = (int)(r % OW);
          const int64_t t = r / OW;
          const int y = (int)(t % OH), b = (int)(t / OH);
          roff[threadIdx.x] = (((int64_t)b * H2 + padT + y) * W2 + padL + x) * Ctot + coff;
        }
      }
      __syncthreads();
      const int n = (int)((r1 - rb) < 256 ? (r1 - rb) : 256);
      if (cok) {
        int i = ry;
        for (; i + 3 * rl < n; i += 4 * rl) {
          const float4 a0 = *reinterpret_cast<const float4*>(dcat + roff[i] + 4 * c4);
          const float4 a1 = *reinterpret_cast<const float4*>(dcat + roff[i + rl] + 4 * c4);
          const float4 a2 = *reinterpret_cast<const float4*>(dcat + roff[i + 2 * rl] + 4 * c4);
          const float4 a3 = *reinterpret_cast<const float4*>(dcat + roff[i + 3 * rl] + 4 * c4);
          s.x += (a0.x + a1.x) + (a2.x + a3.x); s.y += (a0.y + a1.y) + (a2.y + a3.y);
          s.z += (a0.z + a1.z) + (a2.z + a3.z); s.w += (a0.w + a1.w) + (a2.w + a3.w);
        }
        for (; i < n; i += rl) {
          const float4 a0 = *reinterpret_cast<const float4*>(dcat + roff[i] + 4 * c4);
          s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
        }
      }
    }
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    if (ry == 0 && cok) {
      float4 a = red[cx];
      for (int j = 1; j < rl; ++j) {          // fixed order: the sum does not depend on the schedule
        const float4 t = red[(j << clog) + cx];
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      float* o = partial + (int64_t)blockIdx.x * Cout + 4 * c4;
      o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    }
  }
}

// dU[m][c] = sum_{k < C} (d_hi[m][k] + d_lo[m][k]) * w[k][c]: the 1x1 classifier's input gradient as one pass over dU.  Lane =
// 4 consecutive channels (its C x 4 weights stay in registers), 256 / (Cq / 4) rows per block pass; a row's gradient halves are
// the same 2 x C values for all of its lanes (one cached line).
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_dgrad_small_kernel(const T* __restrict__ dh, const T* __restrict__ dl, int64_t ldd,
                                                                  const float* __restrict__ w, float* __restrict__ dU, int64_t M,
                                                                  int Cq, int C) {
  const int cg = Cq >> 2, rpb = 256 / cg;
  const int c4 = threadIdx.x % cg, rl = threadIdx.x / cg;
  if (rl >= rpb) return;
  float4 wr[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) wr[k] = k < C ? *reinterpret_cast<const float4*>(w + (int64_t)k * Cq + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t m = (int64_t)blockIdx.x * rpb + rl; m < M; m += (int64_t)gridDim.x * rpb) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < C) {
        const float d = (float)dh[m * ldd + k] + (dl ? (float)dl[m * ldd + k] : 0.f);
        a.x = __builtin_fmaf(d, wr[k].x, a.x); a.y = __builtin_fmaf(d, wr[k].y, a.y);
        a.z = __builtin_fmaf(d, wr[k].z, a.z); a.w = __builtin_fmaf(d, wr[k].w, a.w);
      }
    *reinterpret_cast<float4*>(dU + m * Cq + 4 * c4) = a;
  }
}

// ---- FCUUp + FusionModel of the OR-UNet fuse head (eval/eval_dinov2_or_unet_fuse.py:502-530) -------------------------
// x <- relu(x + nearest(r)) in place on the split-precision map x [B,H,W,C]; r [B,h,w,C] is the projected ViT map after
// BatchNorm + ReLU, F.interpolate(size=(H, W)) in its default 'nearest' mode = source pixel (ys[y], xs[x]), the tables made
// on the host with ATen's own float arithmetic.  Both addends are post-ReLU (>= 0), so the ReLU of the sum never clips; it
// stays in the formula for the -0.0 / NaN behaviour only.
template <typename T>
__global__ __launch_bounds__(256) void nearest_add_relu_kernel(T* __restrict__ x, T* __restrict__ x_lo,
                                                               const T* __restrict__ r, const T* __restrict__ r_lo,
                                                               const int* __restrict__ ys, const int* __restrict__ xs, int B,
                                                               int H, int W, int h, int w, int C) {
  typedef typename T16<T>::v8 v8;
  const int cpt = C >> 3;
  const int64_t total = (int64_t)B * H * W * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt) << 3;
    const int64_t pix = i / cpt;
    const int xx = (int)(pix % W);
    const int yy = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    const int64_t so = (((int64_t)b * h + ys[yy]) * w + xs[xx]) * C + c;
    const int64_t o = pix * C + c;
    v8 xh = *reinterpret_cast<const v8*>(x + o), xl, rh = *reinterpret_cast<const v8*>(r + so), rl;
    if (x_lo) xl = *reinterpret_cast<const v8*>(x_lo + o);
    if (r_lo) rl = *reinterpret_cast<const v8*>(r_lo + so);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = ((float)xh[k] + (x_lo ? (float)xl[k] : 0.f)) + ((float)rh[k] + (r_lo ? (float)rl[k] : 0.f));
      v = v > 0.f ? v : 0.f;
      const T hi = (T)v;
      xh[k] = hi;
      if (x_lo) xl[k] = (T)(v - (float)hi);
    }
    *reinterpret_cast<v8*>(x + o) = xh;
    if (x_lo) *reinterpret_cast<v8*>(x_lo + o) = xl;
  }
}

// transpose of the nearest resize: dr[b, sy, sx, :] = sum of g over the destination pixels that read source (sy, sx) —
// rows y0[sy] .. y0[sy+1]-1, columns x0[sx] .. x0[sx+1]-1 (the tables are monotone, so the pre-image is a rectangle)
__global__ __launch_bounds__(256) void nearest_sum_kernel(const float* __restrict__ g, float* __restrict__ dr,
                                                          const int* __restrict__ y0, const int* __restrict__ x0, int B, int H,
                                                          int W, int h, int w, int C) {
  const int cpt = C >> 2;
  const int64_t total = (int64_t)B * h * w * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt) << 2;
    const int64_t pix = i / cpt;
    const int sx = (int)(pix % w);
    const int sy = (int)((pix / w) % h);
    const int b = (int)(pix / ((int64_t)w * h));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int yy = y0[sy]; yy < y0[sy + 1]; ++yy)
      for (int xx = x0[sx]; xx < x0[sx + 1]; ++xx) {
        const float4 v = *reinterpret_cast<const float4*>(g + (((int64_t)b * H + yy) * W + xx) * C + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    *reinterpret_cast<float4*>(dr + pix * C + c) = acc;
  }
}

}  // namespace

#define DT_OK(dtype, name) ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, name ": bad dtype %d", dtype)

extern "C" int asis_maxpool2_fwd(void* stream, int dtype, const void* x, const void* x_lo, void* out, void* out_lo,
                                 uint8_t* idx, int B, int H, int W, int C) {
  ASIS_REQUIRE(x && out, "asis_maxpool2_fwd: null pointer");
  ASIS_REQUIRE((x_lo == nullptr) == (out_lo == nullptr), "asis_maxpool2_fwd: x_lo and out_lo go together");
  ASIS_REQUIRE(C > 0 && C % 8 == 0 && H >= 2 && W >= 2, "asis_maxpool2_fwd: bad C=%d (multiple of 8) / H=%d W=%d", C, H, W);
  DT_OK(dtype, "asis_maxpool2_fwd");
  const int OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)B * OH * OW * (C / 8);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((maxpool2_fwd_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, reinterpret_cast<const f16*>(x),
                       reinterpret_cast<const f16*>(x_lo), reinterpret_cast<f16*>(out), reinterpret_cast<f16*>(out_lo), idx,
                       B, H, W, OH, OW, C);
  else
    hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s,
                       reinterpret_cast<const bf16*>(x), reinterpret_cast<const bf16*>(x_lo), reinterpret_cast<bf16*>(out),
                       reinterpret_cast<bf16*>(out_lo), idx, B, H, W, OH, OW, C);
  ASIS_CHECK_LAUNCH("asis_maxpool2_fwd");
  return ASIS_OK;
}

extern "C" int asis_maxpool2_bwd(void* stream, const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C) {
  ASIS_REQUIRE(dy && idx && dx, "asis_maxpool2_bwd: null pointer");
  ASIS_REQUIRE(C > 0 && C % 8 == 0 && H >= 2 && W >= 2, "asis_maxpool2_bwd: bad C=%d / H=%d W=%d", C, H, W);
  const int OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, idx,
                     dx, B, H, W, OH, OW, C);
  ASIS_CHECK_LAUNCH("asis_maxpool2_bwd");
  return ASIS_OK;
}

static int convt_geom_ok(const char* name, int B, int H, int W, int Cout, int H2, int W2, int Ctot, int coff, int padT,
                         int padL) {
  ASIS_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0, "%s: Cout=%d must be a positive multiple of 8", name, Cout);
  ASIS_REQUIRE(coff >= 0 && coff % 8 == 0 && Ctot % 8 == 0 && coff + Cout <= Ctot, "%s: bad channel slice %d+%d of %d", name,
               coff, Cout, Ctot);
  ASIS_REQUIRE(padT >= 0 && padL >= 0 && padT + 2 * H <= H2 && padL + 2 * W <= W2,
               "%s: the 2x upsampled map (%dx%d at +%d,+%d) does not fit the %dx%d destination", name, 2 * H, 2 * W, padT,
               padL, H2, W2);
  return ASIS_OK;
}

extern "C" int asis_convt2x2_scatter(void* stream, int dtype, const float* G, void* dst, void* dst_lo, int B, int H, int W,
                                     int Cout, int H2, int W2, int Ctot, int coff, int padT, int padL) {
  ASIS_REQUIRE(G && dst, "asis_convt2x2_scatter: null pointer");
  DT_OK(dtype, "asis_convt2x2_scatter");
  int rc = convt_geom_ok("asis_convt2x2_scatter", B, H, W, Cout, H2, W2, Ctot, coff, padT, padL);
  if (rc != ASIS_OK) return rc;
  const int64_t total = (int64_t)B * H * W * (Cout / 8);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((convt2x2_scatter_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, G, reinterpret_cast<f16*>(dst),
                       reinterpret_cast<f16*>(dst_lo), B, H, W, Cout, H2, W2, Ctot, coff, padT, padL);
  else
    hipLaunchKernelGGL((convt2x2_scatter_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, G,
                       reinterpret_cast<bf16*>(dst), reinterpret_cast<bf16*>(dst_lo), B, H, W, Cout, H2, W2, Ctot, coff, padT,
                       padL);
  ASIS_CHECK_LAUNCH("asis_convt2x2_scatter");
  return ASIS_OK;
}

extern "C" int asis_convt2x2_gather(void* stream, int dtype, const float* dcat, void* dG, void* dG_lo, int B, int H, int W,
                                    int Cout, int H2, int W2, int Ctot, int coff, int padT, int padL) {
  ASIS_REQUIRE(dcat && dG, "asis_convt2x2_gather: null pointer");
  DT_OK(dtype, "asis_convt2x2_gather");
  int rc = convt_geom_ok("asis_convt2x2_gather", B, H, W, Cout, H2, W2, Ctot, coff, padT, padL);
  if (rc != ASIS_OK) return rc;
  const int64_t total = (int64_t)B * H * W * (Cout / 8);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((convt2x2_gather_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, dcat, reinterpret_cast<f16*>(dG),
                       reinterpret_cast<f16*>(dG_lo), B, H, W, Cout, H2, W2, Ctot, coff, padT, padL);
  else
    hipLaunchKernelGGL((convt2x2_gather_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, dcat,
                       reinterpret_cast<bf16*>(dG), reinterpret_cast<bf16*>(dG_lo), B, H, W, Cout, H2, W2, Ctot, coff, padT,
                       padL);
  ASIS_CHECK_LAUNCH("asis_convt2x2_gather");
  return ASIS_OK;
}

extern "C" int asis_conv1x1_dgrad_small(void* stream, int dtype, const void* d_hi, const void* d_lo, int64_t ldd, const float* w,
                                        float* dU, int64_t M, int Cq, int C) {
  ASIS_REQUIRE(d_hi && w && dU, "asis_conv1x1_dgrad_small: null pointer");
  DT_OK(dtype, "asis_conv1x1_dgrad_small");
  ASIS_REQUIRE(M > 0 && C >= 1 && C <= 8 && ldd >= C && Cq > 0 && Cq % 4 == 0 && Cq <= 1024,
               "asis_conv1x1_dgrad_small: C=%d (1..8), Cq=%d (multiple of 4, <= 1024), ldd=%ld", C, Cq, (long)ldd);
  ASIS_REQUIRE(asis_aligned16(w) && asis_aligned16(dU), "asis_conv1x1_dgrad_small: w and dU must be 16-byte aligned");
  const int rpb = 256 / (Cq / 4);
  int64_t nblk = (M + rpb - 1) / rpb;
  if (nblk > 256 * 32) nblk = 256 * 32;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((conv1x1_dgrad_small_kernel<f16>), dim3((unsigned)nblk), dim3(256), 0, s, reinterpret_cast<const f16*>(d_hi),
                       reinterpret_cast<const f16*>(d_lo), ldd, w, dU, M, Cq, C);
  else
    hipLaunchKernelGGL((conv1x1_dgrad_small_kernel<bf16>), dim3((unsigned)nblk), dim3(256), 0, s, reinterpret_cast<const bf16*>(d_hi),
                       reinterpret_cast<const bf16*>(d_lo), ldd, w, dU, M, Cq, C);
  ASIS_CHECK_LAUNCH("asis_conv1x1_dgrad_small");
  return ASIS_OK;
}

extern "C" int asis_convt2x2_bias_nblk(int64_t rows) {
  int64_t n = (rows + 255) / 256;
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int asis_convt2x2_bias_grad(void* stream, const float* dcat, float* partial, int B, int H, int W, int Cout, int H2,
                                       int W2, int Ctot, int coff, int padT, int padL) {
  ASIS_REQUIRE(dcat && partial, "asis_convt2x2_bias_grad: null pointer");
  int rc = convt_geom_ok("asis_convt2x2_bias_grad", B, H, W, Cout, H2, W2, Ctot, coff, padT, padL);
  if (rc != ASIS_OK) return rc;
  const int64_t rows = (int64_t)B * 2 * H * 2 * W;
  const int nblk = asis_convt2x2_bias_nblk(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipLaunchKernelGGL(convt2x2_bias_grad_kernel, dim3(nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dcat, partial,
                     B, 2 * H, 2 * W, Cout, H2, W2, Ctot, coff, padT, padL, rpb);
  ASIS_CHECK_LAUNCH("asis_convt2x2_bias_grad");
  return ASIS_OK;
}

extern "C" int asis_nearest_add_relu(void* stream, int dtype, void* x, void* x_lo, const void* r, const void* r_lo,
                                     const int* ys, const int* xs, int B, int H, int W, int h, int w, int C) {
  ASIS_REQUIRE(x && r && ys && xs, "asis_nearest_add_relu: null pointer");
  ASIS_REQUIRE(C > 0 && C % 8 == 0 && B > 0 && H > 0 && W > 0 && h > 0 && w > 0, "asis_nearest_add_relu: bad shape (C=%d must be a multiple of 8)", C);
  DT_OK(dtype, "asis_nearest_add_relu");
  const int64_t total = (int64_t)B * H * W * (C / 8);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((nearest_add_relu_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, reinterpret_cast<f16*>(x),
                       reinterpret_cast<f16*>(x_lo), reinterpret_cast<const f16*>(r), reinterpret_cast<const f16*>(r_lo), ys, xs,
                       B, H, W, h, w, C);
  else
    hipLaunchKernelGGL((nearest_add_relu_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, reinterpret_cast<bf16*>(x),
                       reinterpret_cast<bf16*>(x_lo), reinterpret_cast<const bf16*>(r), reinterpret_cast<const bf16*>(r_lo), ys,
                       xs, B, H, W, h, w, C);
  ASIS_CHECK_LAUNCH("asis_nearest_add_relu");
  return ASIS_OK;
}

extern "C" int asis_nearest_sum(void* stream, const float* g, float* dr, const int* y0, const int* x0, int B, int H, int W,
                                int h, int w, int C) {
  ASIS_REQUIRE(g && dr && y0 && x0, "asis_nearest_sum: null pointer");
  ASIS_REQUIRE(C > 0 && C % 4 == 0 && B > 0 && H > 0 && W > 0 && h > 0 && w > 0, "asis_nearest_sum: bad shape (C=%d must be a multiple of 4)", C);
  const int64_t total = (int64_t)B * h * w * (C / 4);
  hipLaunchKernelGGL(nearest_sum_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, dr, y0,
                     x0, B, H, W, h, w, C);
  ASIS_CHECK_LAUNCH("asis_nearest_sum");
  return ASIS_OK;
}
