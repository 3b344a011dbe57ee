// Row / elementwise backward kernels of the DINOv2 block (dinov2/layers/block.py:89-114), all HBM-bound streaming:
//   LayerNorm backward            nn.LayerNorm(eps=1e-6), block.py:63,75
//   GELU (erf) forward / backward on 16-bit operands (mlp.py:35) — the training forward keeps the pre-activation
//   column sums of a 16-bit matrix (bias gradients of qkv / fc1)
//   LayerScale + Linear finish    out = x + gamma * (A W^T + b): from G = dout^T A (unscaled wgrad GEMM) and
//                                 cs = colsum(dout):  dW = gamma G,  db = gamma cs,  dgamma = rowsum(W * G) + b cs
// One wave64 per row for the row kernels, 16-byte accesses, fp32 arithmetic.
#include "asis_common.h"

namespace {

constexpr int LN_MAXC = 8;  // float4 chunks per lane -> D <= 2048

inline int grid_for(int64_t total, int cap = 65535 * 4) {
  int64_t g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// dx[row] = (res ? res[row] : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w
// partial[blk][0][c] = sum_rows dy * xhat (d weight),  partial[blk][1][c] = sum_rows dy (d bias)
// Each block owns ROWS_PER_BLOCK consecutive rows, 4 waves striding over them; statistics are recomputed from x.
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ w, float eps,
                                                            const float* __restrict__ res, int64_t ldr,
                                                            float* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ partial, int64_t rows, int D,
                                                            int rows_per_block) {
  __shared__ float red[3][2][MAXC * 64 * 4];  // waves 1..3 park their column sums here
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nchunk = D >> 2;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 gw[MAXC], gb[MAXC], ww[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    gw[i] = gb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lane + 64 * i;
    ww[i] = c < nchunk ? w4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const float invD = 1.0f / (float)D;
  // the next row's loads (x, dy, residual) are issued before this row's three wave reductions: two rows in flight per wave
  constexpr bool PF = MAXC <= 6;   // 6 x MAXC float4 of row data + 3 x MAXC of sums / weight must fit 256 VGPRs
  float4 v[MAXC], g[MAXC], e[MAXC];
  auto load_row = [&](int64_t row, float4* vv, float4* gg, float4* ee) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
    const float4* gr = reinterpret_cast<const float4*>(dy + row * lddy);
    const float4* rr = res ? reinterpret_cast<const float4*>(res + row * ldr) : nullptr;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        vv[i] = xr[c];
        gg[i] = gr[c];
        if (rr) ee[i] = rr[c];
      }
    }
  };
  int64_t row = r0 + wid;
  if constexpr (PF) {
    if (row < r1) load_row(row, v, g, e);
  }
  for (; row < r1; row += 4) {
    float4 vn[PF ? MAXC : 1], gn[PF ? MAXC : 1], en[PF ? MAXC : 1];
    if constexpr (PF) {
      if (row + 4 < r1) load_row(row + 4, vn, gn, en);
    } else {
      load_row(row, v, g, e);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + eps);
    float a = 0.f, bq = 0.f;  // sum g, sum g*xhat
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // xhat
        gw[i].x += g[i].x * v[i].x; gw[i].y += g[i].y * v[i].y; gw[i].z += g[i].z * v[i].z; gw[i].w += g[i].w * v[i].w;
        gb[i].x += g[i].x; gb[i].y += g[i].y; gb[i].z += g[i].z; gb[i].w += g[i].w;
        g[i].x *= ww[i].x; g[i].y *= ww[i].y; g[i].z *= ww[i].z; g[i].w *= ww[i].w;
        a += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        bq += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
      }
    }
    a = wave_sum(a) * invD;
    bq = wave_sum(bq) * invD;
    float4* o = reinterpret_cast<float4*>(dx + row * lddx);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float4 t;
        t.x = rstd * (g[i].x - a - v[i].x * bq);
        t.y = rstd * (g[i].y - a - v[i].y * bq);
        t.z = rstd * (g[i].z - a - v[i].z * bq);
        t.w = rstd * (g[i].w - a - v[i].w * bq);
        if (res) {
          t.x += e[i].x; t.y += e[i].y; t.z += e[i].z; t.w += e[i].w;
        }
        o[c] = t;
      }
    }
    if constexpr (PF) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        v[i] = vn[i];
        g[i] = gn[i];
        e[i] = en[i];
      }
    }
  }
  // column sums over the block's rows: waves 1..3 -> LDS -> wave 0 adds and writes
  if (wid > 0) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      reinterpret_cast<float4*>(red[wid - 1][0])[i * 64 + lane] = gw[i];
      reinterpret_cast<float4*>(red[wid - 1][1])[i * 64 + lane] = gb[i];
    }
  }
  __syncthreads();
  if (wid == 0) {
    float4* pw = reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * D);
    float4* pb = reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * D);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float4 sw = gw[i], sb = gb[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 t = reinterpret_cast<const float4*>(red[k][0])[i * 64 + lane];
          const float4 u = reinterpret_cast<const float4*>(red[k][1])[i * 64 + lane];
          sw.x += t.x; sw.y += t.y; sw.z += t.z; sw.w += t.w;
          sb.x += u.x; sb.y += u.y; sb.z += u.z; sb.w += u.w;
        }
        pw[c] = sw;
        pb[c] = sb;
      }
    }
  }
}

// exact (libm erff) derivative is used in the backward: one pass, not in a GEMM epilogue
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu16_kernel(const T* __restrict__ pre, const T* __restrict__ dpost, T* __restrict__ out,
                                                     int64_t n8) {
  typedef typename T16<T>::v8 v8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const v8 p = reinterpret_cast<const v8*>(pre)[i];
    v8 o;
    if (BWD) {
      const v8 g = reinterpret_cast<const v8*>(dpost)[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (T)((float)g[k] * gelu_erf_grad((float)p[k]));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (T)gelu_erf((float)p[k]);
    }
    reinterpret_cast<v8*>(out)[i] = o;
  }
}

// SwiGLU gate backward (swiglu_ffn.py:30-34): h = silu(x1) * x2 with x12 = [x1 | x2] fp32 [R, 2 Hd], dh 16-bit [R, Hd]
//   d x1 = dh * x2 * sig(x1) * (1 + x1 (1 - sig(x1))),  d x2 = dh * silu(x1)      -> 16-bit [R, 2 Hd]
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const float* __restrict__ x12, const T* __restrict__ dh,
                                                         T* __restrict__ dx12, int64_t R, int Hd) {
  const int cpr = Hd >> 2;
  const int64_t total = R * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr);
    const float4 a = reinterpret_cast<const float4*>(x12 + r * 2 * Hd)[c];
    const float4 b = reinterpret_cast<const float4*>(x12 + r * 2 * Hd + Hd)[c];
    const uint2 gw = reinterpret_cast<const uint2*>(dh + r * Hd)[c];
    float g[4];
    unpack2<T>(gw.x, g[0], g[1]);
    unpack2<T>(gw.y, g[2], g[3]);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    float d1[4], d2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sg = 1.f / (1.f + __expf(-av[k]));
      d1[k] = g[k] * bv[k] * sg * (1.f + av[k] * (1.f - sg));
      d2[k] = g[k] * av[k] * sg;
    }
    uint2 o;
    o.x = pack2<T>(d1[0], d1[1]);
    o.y = pack2<T>(d1[2], d1[3]);
    reinterpret_cast<uint2*>(dx12 + r * 2 * Hd)[c] = o;
    o.x = pack2<T>(d2[0], d2[1]);
    o.y = pack2<T>(d2[2], d2[3]);
    reinterpret_cast<uint2*>(dx12 + r * 2 * Hd + Hd)[c] = o;
  }
}

// partial[blk][c] = sum over the block's rows of x[r, c] (16-bit input, fp32 sums); threads run along columns (8 per thread)
template <typename T>
__global__ __launch_bounds__(256) void colsum16_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ partial,
                                                       int64_t rows, int C, int rows_per_block) {
  typedef typename T16<T>::v8 v8;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int c8 = blockIdx.y * blockDim.x + threadIdx.x; c8 < (C >> 3); c8 += gridDim.y * blockDim.x) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {   // four independent 16-byte loads in flight per thread
      v8 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const v8*>(x + (r + j) * ld + c8 * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += (float)v[j][k];
    }
    for (; r < r1; ++r) {
      const v8 v = *reinterpret_cast<const v8*>(x + r * ld + c8 * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
    }
    float4* o = reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * C + c8 * 8);
    o[0] = make_float4(s[0], s[1], s[2], s[3]);
    o[1] = make_float4(s[4], s[5], s[6], s[7]);
  }
}

// fp32 variant (column sums of the residual-stream gradient)
__global__ __launch_bounds__(256) void colsum32_kernel(const float* __restrict__ x, int64_t ld, float* __restrict__ partial,
                                                       int64_t rows, int C, int rows_per_block) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int c4 = blockIdx.y * blockDim.x + threadIdx.x; c4 < (C >> 2); c4 += gridDim.y * blockDim.x) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = r0; r < r1; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * C + c4 * 4) = s;
  }
}

// out16[r, :] = (T)(scale * x[r, :]) and partial[blk][c] = sum over the block's rows of x[r, c]: the 16-bit GEMM operand of
// a residual-stream gradient and its column sums (LayerScale / bias gradients) in one read of the fp32 rows
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float* __restrict__ x, int64_t ldx, T* __restrict__ out,
                                                          int64_t ldo, float scale, float* __restrict__ partial,
                                                          int64_t rows, int D, int rows_per_block) {
  __shared__ float red[3][MAXC * 64 * 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nchunk = D >> 2;
  float4 acc[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int64_t row = r0 + wid; row < r1; row += 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
    uint2* o = reinterpret_cast<uint2*>(out + row * ldo);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        const float4 v = xr[c];
        acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
        uint2 p;
        p.x = pack2<T>(v.x * scale, v.y * scale);
        p.y = pack2<T>(v.z * scale, v.w * scale);
        o[c] = p;
      }
    }
  }
  if (wid > 0) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) reinterpret_cast<float4*>(red[wid - 1])[i * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wid == 0) {
    float4* pw = reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * D);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float4 s = acc[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 t = reinterpret_cast<const float4*>(red[k])[i * 64 + lane];
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        pw[c] = s;
      }
    }
  }
}

// one wave per output feature n: dW[n,:] = gs * gamma[n] * G[n,:];  dgamma[n] = gs * (sum_k W[n,k] G[n,k] + b[n] cs[n]);
// db[n] = gs * gamma[n] * cs[n]     (gs = 1 / loss scale; gamma NULL = plain Linear: dW = gs G, db = gs cs)
__global__ __launch_bounds__(256) void ls_linear_finish_kernel(const float* __restrict__ G, const float* __restrict__ Wt,
                                                               const float* __restrict__ bias, const float* __restrict__ gamma,
                                                               const float* __restrict__ cs, float gs, float* __restrict__ dW,
                                                               float* __restrict__ db, float* __restrict__ dgamma, int N,
                                                               int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float ga = gamma ? gamma[n] : 1.f;
  const float4* g4 = reinterpret_cast<const float4*>(G + (int64_t)n * K);
  const float4* w4 = reinterpret_cast<const float4*>(Wt + (int64_t)n * K);
  float4* o4 = reinterpret_cast<float4*>(dW + (int64_t)n * K);
  float dot = 0.f;
  const float sc = gs * ga;
  for (int c = lane; c < (K >> 2); c += 64) {
    const float4 g = g4[c];
    if (gamma) {
      const float4 w = w4[c];
      dot += (g.x * w.x + g.y * w.y) + (g.z * w.z + g.w * w.w);
    }
    o4[c] = make_float4(g.x * sc, g.y * sc, g.z * sc, g.w * sc);
  }
  if (gamma) dot = wave_sum(dot);
  if (lane == 0) {
    const float c = cs ? cs[n] : 0.f;
    if (db) db[n] = sc * c;
    if (gamma && dgamma) dgamma[n] = gs * (dot + (bias ? bias[n] : 0.f) * c);
  }
}

}  // namespace

#define DT_OK(dtype, name) ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, name ": bad dtype %d", dtype)

extern "C" int asis_rowblock_nblk(int64_t rows) {
  int64_t n = (rows + 31) / 32;
  if (n > 2048) n = 2048;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int asis_layernorm_bwd(void* stream, const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w,
                                  float eps, const float* res, int64_t ldr, float* dx, int64_t lddx, float* partial,
                                  int64_t rows, int D) {
  ASIS_REQUIRE(dy && x && w && dx && partial, "asis_layernorm_bwd: null pointer");
  ASIS_REQUIRE(D > 0 && D % 4 == 0 && D <= LN_MAXC * 256, "asis_layernorm_bwd: D=%d must be a multiple of 4, <= %d", D, LN_MAXC * 256);
  ASIS_REQUIRE(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!res || ldr % 4 == 0), "asis_layernorm_bwd: row strides must be multiples of 4");
  ASIS_REQUIRE(rows > 0, "asis_layernorm_bwd: no rows");
  const int nblk = asis_rowblock_nblk(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (D <= 1024)
    hipLaunchKernelGGL((layernorm_bwd_kernel<4>), dim3(nblk), dim3(256), 0, s, dy, lddy, x, ldx, w, eps, res, ldr, dx, lddx,
                       partial, rows, D, rpb);
  else if (D <= 1536)
    hipLaunchKernelGGL((layernorm_bwd_kernel<6>), dim3(nblk), dim3(256), 0, s, dy, lddy, x, ldx, w, eps, res, ldr, dx, lddx,
                       partial, rows, D, rpb);
  else
    hipLaunchKernelGGL((layernorm_bwd_kernel<LN_MAXC>), dim3(nblk), dim3(256), 0, s, dy, lddy, x, ldx, w, eps, res, ldr, dx,
                       lddx, partial, rows, D, rpb);
  ASIS_CHECK_LAUNCH("asis_layernorm_bwd");
  return ASIS_OK;
}

namespace {
// erf-GELU of an fp32 pre-activation into a split-precision 16-bit operand pair (hi, lo = rounding residual of hi)
template <typename T>
__global__ __launch_bounds__(256) void gelu_split_kernel(const float* __restrict__ x, T* __restrict__ hi, T* __restrict__ lo, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float g0 = v.x, g1 = v.y, g2 = v.z, g3 = v.w;
    gelu_erf4(g0, g1, g2, g3);
    uint2 h;
    h.x = pack2<T>(g0, g1); h.y = pack2<T>(g2, g3);
    reinterpret_cast<uint2*>(hi)[i] = h;
    if (lo) {
      uint2 l;
      l.x = pack2<T>(lo_part<T>(g0), lo_part<T>(g1)); l.y = pack2<T>(lo_part<T>(g2), lo_part<T>(g3));
      reinterpret_cast<uint2*>(lo)[i] = l;
    }
  }
}
}  // namespace

extern "C" int asis_gelu_split(void* stream, int dtype, const float* x, void* hi, void* lo, int64_t n) {
  ASIS_REQUIRE(x && hi && n > 0 && n % 4 == 0, "asis_gelu_split: bad arguments (n %% 4 == 0)");
  DT_OK(dtype, "asis_gelu_split");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for(n / 4);
  if (dtype == ASIS_F16) hipLaunchKernelGGL((gelu_split_kernel<f16>), dim3(g), dim3(256), 0, s, x, (f16*)hi, (f16*)lo, n / 4);
  else hipLaunchKernelGGL((gelu_split_kernel<bf16>), dim3(g), dim3(256), 0, s, x, (bf16*)hi, (bf16*)lo, n / 4);
  ASIS_CHECK_LAUNCH("asis_gelu_split");
  return ASIS_OK;
}

extern "C" int asis_gelu16(void* stream, int dtype, const void* pre, const void* dpost, void* out, int64_t n) {
  ASIS_REQUIRE(pre && out && n > 0 && n % 8 == 0, "asis_gelu16: bad arguments (n %% 8 == 0)");
  DT_OK(dtype, "asis_gelu16");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t n8 = n / 8;
  const int g = grid_for(n8);
  if (dtype == ASIS_F16) {
    if (dpost) hipLaunchKernelGGL((gelu16_kernel<f16, true>), dim3(g), dim3(256), 0, s, (const f16*)pre, (const f16*)dpost, (f16*)out, n8);
    else hipLaunchKernelGGL((gelu16_kernel<f16, false>), dim3(g), dim3(256), 0, s, (const f16*)pre, (const f16*)nullptr, (f16*)out, n8);
  } else {
    if (dpost) hipLaunchKernelGGL((gelu16_kernel<bf16, true>), dim3(g), dim3(256), 0, s, (const bf16*)pre, (const bf16*)dpost, (bf16*)out, n8);
    else hipLaunchKernelGGL((gelu16_kernel<bf16, false>), dim3(g), dim3(256), 0, s, (const bf16*)pre, (const bf16*)nullptr, (bf16*)out, n8);
  }
  ASIS_CHECK_LAUNCH("asis_gelu16");
  return ASIS_OK;
}

extern "C" int asis_swiglu_bwd(void* stream, int dtype, const float* x12, const void* dh, void* dx12, int64_t R, int Hd) {
  ASIS_REQUIRE(x12 && dh && dx12 && R > 0 && Hd > 0 && Hd % 4 == 0, "asis_swiglu_bwd: bad arguments (Hd %% 4 == 0)");
  DT_OK(dtype, "asis_swiglu_bwd");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for(R * (Hd / 4));
  if (dtype == ASIS_F16) hipLaunchKernelGGL((swiglu_bwd_kernel<f16>), dim3(g), dim3(256), 0, s, x12, (const f16*)dh, (f16*)dx12, R, Hd);
  else hipLaunchKernelGGL((swiglu_bwd_kernel<bf16>), dim3(g), dim3(256), 0, s, x12, (const bf16*)dh, (bf16*)dx12, R, Hd);
  ASIS_CHECK_LAUNCH("asis_swiglu_bwd");
  return ASIS_OK;
}

extern "C" int asis_colsum(void* stream, int dtype, const void* x, int64_t ld, float* partial, int64_t rows, int C) {
  ASIS_REQUIRE(x && partial && rows > 0, "asis_colsum: bad arguments");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16 || dtype == ASIS_F32, "asis_colsum: bad dtype %d", dtype);
  ASIS_REQUIRE(C > 0 && C % 8 == 0 && ld % 8 == 0 && ld >= C, "asis_colsum: C=%d and ld must be multiples of 8", C);
  const int nblk = asis_rowblock_nblk(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // gridDim.y walks the columns (256 threads x 8 / 4 columns each), so few-rows x very-wide inputs (the batch sum of a
  // [B, tokens*D] gradient) still fill the chip
  const int per = dtype == ASIS_F32 ? 4 : 8;
  int gy = (C / per + 255) / 256;
  if (gy > 4096) gy = 4096;
  if (gy < 1) gy = 1;
  if (dtype == ASIS_F16) hipLaunchKernelGGL((colsum16_kernel<f16>), dim3(nblk, gy), dim3(256), 0, s, (const f16*)x, ld, partial, rows, C, rpb);
  else if (dtype == ASIS_BF16) hipLaunchKernelGGL((colsum16_kernel<bf16>), dim3(nblk, gy), dim3(256), 0, s, (const bf16*)x, ld, partial, rows, C, rpb);
  else hipLaunchKernelGGL(colsum32_kernel, dim3(nblk, gy), dim3(256), 0, s, (const float*)x, ld, partial, rows, C, rpb);
  ASIS_CHECK_LAUNCH("asis_colsum");
  return ASIS_OK;
}

extern "C" int asis_cast_colsum(void* stream, int dtype, const float* x, int64_t ldx, void* out, int64_t ldo, float scale,
                                float* partial, int64_t rows, int D) {
  ASIS_REQUIRE(x && out && partial && rows > 0, "asis_cast_colsum: bad arguments");
  DT_OK(dtype, "asis_cast_colsum");
  ASIS_REQUIRE(D > 0 && D % 4 == 0 && D <= LN_MAXC * 256 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= D,
               "asis_cast_colsum: D=%d must be a multiple of 4, <= %d; row strides multiples of 4", D, LN_MAXC * 256);
  const int nblk = asis_rowblock_nblk(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16) {
    if (D <= 1024) hipLaunchKernelGGL((cast_colsum_kernel<f16, 4>), dim3(nblk), dim3(256), 0, s, x, ldx, (f16*)out, ldo, scale, partial, rows, D, rpb);
    else hipLaunchKernelGGL((cast_colsum_kernel<f16, LN_MAXC>), dim3(nblk), dim3(256), 0, s, x, ldx, (f16*)out, ldo, scale, partial, rows, D, rpb);
  } else {
    if (D <= 1024) hipLaunchKernelGGL((cast_colsum_kernel<bf16, 4>), dim3(nblk), dim3(256), 0, s, x, ldx, (bf16*)out, ldo, scale, partial, rows, D, rpb);
    else hipLaunchKernelGGL((cast_colsum_kernel<bf16, LN_MAXC>), dim3(nblk), dim3(256), 0, s, x, ldx, (bf16*)out, ldo, scale, partial, rows, D, rpb);
  }
  ASIS_CHECK_LAUNCH("asis_cast_colsum");
  return ASIS_OK;
}

extern "C" int asis_ls_linear_finish(void* stream, const float* G, const float* W, const float* bias, const float* gamma,
                                     const float* cs, float grad_scale, float* dW, float* db, float* dgamma, int N, int K) {
  ASIS_REQUIRE(G && dW && N > 0 && K > 0 && K % 4 == 0, "asis_ls_linear_finish: bad arguments (K %% 4 == 0)");
  ASIS_REQUIRE(!gamma || W, "asis_ls_linear_finish: gamma needs the weight");
  hipLaunchKernelGGL(ls_linear_finish_kernel, dim3((N + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G, W, bias,
                     gamma, cs, grad_scale, dW, db, dgamma, N, K);
  ASIS_CHECK_LAUNCH("asis_ls_linear_finish");
  return ASIS_OK;
}
