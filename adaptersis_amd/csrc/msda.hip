// Multi-scale deformable attention sampling core + the ConvFFN depthwise conv of the adapters.
//
// msda_fwd replaces MSDeformAttnFunction / ms_deform_attn_core_pytorch
// (backbones/ops/modules/ms_deform_attn.py:33-54) together with the softmax over L*P and the
// sampling-location arithmetic of MSDeformAttn.forward (:155-166):
//     A   = softmax_{l,p}(logits[b,q,m,:])
//     loc = ref[q] + off[b,q,m,l,p] / (W_l, H_l)
//     pix = loc * (W_l, H_l) - 0.5            (grid_sample, align_corners=False)
//     out[b,q,m,:] = sum_{l,p} A * bilinear(value[b, level l, :, m, :], pix)   zero padding
// It is an HBM/L2 gather: one thread owns 8 contiguous channels (one 16-byte load per tap), the
// Dh/8 threads of a head read one contiguous Dh*2-byte row per tap, offsets/logits are fp32.
#include "asis_common.h"

namespace {

constexpr int MAX_LP = 16;

// NC: 16-byte chunks per thread (1 or 2).  The softmax over the L*P points and the tap arithmetic are repeated by every thread
// of a (query, head) group; two chunks per thread halve that share of the work (Dh % 16 == 0).
template <typename T, int NC = 1>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const T* __restrict__ value, const float* __restrict__ offaw,
                                                       int64_t ld_offaw, const float* __restrict__ ref,
                                                       const int* __restrict__ shapes, const int* __restrict__ starts,
                                                       T* __restrict__ out, T* __restrict__ out_lo, int B, int Lq, int Lin, int M, int L, int P,
                                                       int Dh) {
  const int D = M * Dh;
  const int cpq = D / (8 * NC);   // thread slots per query
  const int cph = Dh / (8 * NC);  // thread slots per head
  const int64_t total = (int64_t)B * Lq * cpq;
  const int LP = L * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpq);
    const int64_t bq = i / cpq;
    const int q = (int)(bq % Lq);
    const int b = (int)(bq / Lq);
    const int m = c / cph;
    const float* orow = offaw + bq * ld_offaw;
    const float* lg = orow + (int64_t)M * LP * 2 + m * LP;
    float w[MAX_LP];
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < MAX_LP; ++j)
      if (j < LP) {
        w[j] = lg[j];
        mx = fmaxf(mx, w[j]);
      }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_LP; ++j)
      if (j < LP) {
        w[j] = __expf(w[j] - mx);
        den += w[j];
      }
    const float inv = 1.0f / den;
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    float acc[8 * NC];
#pragma unroll
    for (int e = 0; e < 8 * NC; ++e) acc[e] = 0.f;
    const T* vb = value + (int64_t)b * Lin * D + c * 8 * NC;
    for (int l = 0; l < L; ++l) {
      const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
      const T* vl = vb + (int64_t)starts[l] * D;
#pragma unroll 4
      for (int p = 0; p < P; ++p) {
        const int j = l * P + p;
        const float ox = orow[(m * LP + j) * 2], oy = orow[(m * LP + j) * 2 + 1];
        const float lx = rx + ox / (float)Wl, ly = ry + oy / (float)Hl;
        const float px = lx * (float)Wl - 0.5f, py = ly * (float)Hl - 0.5f;
        const float fx0 = floorf(px), fy0 = floorf(py);
        const float ax = px - fx0, ay = py - fy0;
        // clamp before the int conversion so wild offsets cannot overflow; the bounds test stays exact
        const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)Wl + 1.f);
        const int y0 = (int)fminf(fmaxf(fy0, -2.f), (float)Hl + 1.f);
        const float aw = w[j] * inv;
        // the four corners are fetched unconditionally from coordinates clamped into the level, an out-of-range
        // corner only loses its weight: a load under `if (in range)` is waited for on the spot, one at a time
        uint4 raw[4][NC];
        float wt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
          const bool inb = (unsigned)xx < (unsigned)Wl && (unsigned)yy < (unsigned)Hl;
          const int xc = xx < 0 ? 0 : (xx >= Wl ? Wl - 1 : xx), yc = yy < 0 ? 0 : (yy >= Hl ? Hl - 1 : yy);
          wt[t] = inb ? ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay) * aw : 0.f;
#pragma unroll
          for (int h = 0; h < NC; ++h) raw[t][h] = *reinterpret_cast<const uint4*>(vl + ((int64_t)yc * Wl + xc) * D + 8 * h);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int h = 0; h < NC; ++h) {
            float f0, f1;
            unpack2<T>(raw[t][h].x, f0, f1);
            acc[8 * h + 0] += wt[t] * f0; acc[8 * h + 1] += wt[t] * f1;
            unpack2<T>(raw[t][h].y, f0, f1);
            acc[8 * h + 2] += wt[t] * f0; acc[8 * h + 3] += wt[t] * f1;
            unpack2<T>(raw[t][h].z, f0, f1);
            acc[8 * h + 4] += wt[t] * f0; acc[8 * h + 5] += wt[t] * f1;
            unpack2<T>(raw[t][h].w, f0, f1);
            acc[8 * h + 6] += wt[t] * f0; acc[8 * h + 7] += wt[t] * f1;
          }
      }
    }
#pragma unroll
    for (int h = 0; h < NC; ++h) {
      uint4 o;
      o.x = pack2<T>(acc[8 * h + 0], acc[8 * h + 1]);
      o.y = pack2<T>(acc[8 * h + 2], acc[8 * h + 3]);
      o.z = pack2<T>(acc[8 * h + 4], acc[8 * h + 5]);
      o.w = pack2<T>(acc[8 * h + 6], acc[8 * h + 7]);
      *reinterpret_cast<uint4*>(out + bq * D + c * 8 * NC + 8 * h) = o;
      if (out_lo) {   // rounding residuals: samp ~= out + out_lo feeds output_proj as a split A operand (bf16 at 1e-3: DESIGN.md §3)
        uint4 ol;
        ol.x = pack2<T>(lo_part<T>(acc[8 * h + 0]), lo_part<T>(acc[8 * h + 1]));
        ol.y = pack2<T>(lo_part<T>(acc[8 * h + 2]), lo_part<T>(acc[8 * h + 3]));
        ol.z = pack2<T>(lo_part<T>(acc[8 * h + 4]), lo_part<T>(acc[8 * h + 5]));
        ol.w = pack2<T>(lo_part<T>(acc[8 * h + 6]), lo_part<T>(acc[8 * h + 7]));
        *reinterpret_cast<uint4*>(out_lo + bq * D + c * 8 * NC + 8 * h) = ol;
      }
    }
  }
}

// DWConv 3x3 (pad 1, bias, depthwise) over the token grids of each pyramid level + erf GELU
// (backbones/adapter_blocks.py:67-80,95-97).  x fp32 [B, Ntok, C]; w9 fp32 [9][C]; out 16-bit.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_gelu_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                          const float* __restrict__ bias, const int* __restrict__ shapes,
                                                          const int* __restrict__ starts, int L, T* __restrict__ out,
                                                          int B, int Ntok, int C) {
  const int cpt = C >> 2;  // float4 chunks per token
  const int64_t total = (int64_t)B * Ntok * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt);
    const int64_t bt = i / cpt;
    const int t = (int)(bt % Ntok);
    const int b = (int)(bt / Ntok);
    int l = 0;
    for (int k = 1; k < L; ++k)
      if (t >= starts[k]) l = k;
    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
    const int p = t - starts[l];
    const int y = p / Wl, xx = p - y * Wl;
    float4 acc = reinterpret_cast<const float4*>(bias)[c];
    const float* xb = x + ((int64_t)b * Ntok + starts[l]) * C;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xs = xx + kx - 1;
        if ((unsigned)yy < (unsigned)Hl && (unsigned)xs < (unsigned)Wl) {
          const float4 v = reinterpret_cast<const float4*>(xb + ((int64_t)yy * Wl + xs) * C)[c];
          const float4 ww = reinterpret_cast<const float4*>(w9 + (ky * 3 + kx) * C)[c];
          acc.x += v.x * ww.x;
          acc.y += v.y * ww.y;
          acc.z += v.z * ww.z;
          acc.w += v.w * ww.w;
        }
      }
    uint2 o;
    o.x = pack2<T>(gelu_erf(acc.x), gelu_erf(acc.y));
    o.y = pack2<T>(gelu_erf(acc.z), gelu_erf(acc.w));
    reinterpret_cast<uint2*>(out + bt * C)[c] = o;
  }
}

inline int grid_for(int64_t total, int block = 256, int cap = 256 * 16) {
  int64_t g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int asis_msda_fwd_split(void* stream, int dtype, const void* value, const float* offaw, int64_t ld_offaw,
                                   const float* ref, const int32_t* shapes, const int32_t* starts, void* out, void* out_lo, int B,
                                   int Lq, int Lin, int M, int L, int P, int Dh) {
  ASIS_REQUIRE(value && offaw && ref && shapes && starts && out, "asis_msda_fwd: null pointer");
  ASIS_REQUIRE(!out_lo || asis_aligned16(out_lo), "asis_msda_fwd_split: out_lo must be 16-byte aligned");
  ASIS_REQUIRE(B > 0 && Lq > 0 && Lin > 0 && M > 0 && L > 0 && P > 0, "asis_msda_fwd: bad shape");
  ASIS_REQUIRE(Dh % 8 == 0, "asis_msda_fwd: head dim %d must be a multiple of 8", Dh);
  ASIS_REQUIRE(L * P <= MAX_LP, "asis_msda_fwd: n_levels*n_points=%d exceeds %d", L * P, MAX_LP);
  ASIS_REQUIRE(ld_offaw >= (int64_t)M * L * P * 3, "asis_msda_fwd: ld_offaw too small");
  ASIS_REQUIRE(asis_aligned16(value) && asis_aligned16(out), "asis_msda_fwd: value/out must be 16-byte aligned");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_msda_fwd: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // ASIS_MSDA_NC (default 2): 16-byte chunks per thread when the head dim allows it
  static const int nc_env = [] { const char* e = getenv("ASIS_MSDA_NC"); return e ? atoi(e) : 2; }();
  const int nc = (nc_env >= 2 && Dh % 16 == 0) ? 2 : 1;
  const int64_t total = (int64_t)B * Lq * (M * Dh / (8 * nc));
#define ASIS_MSDA_FWD(TT, NCC)                                                                                              \
  hipLaunchKernelGGL((msda_fwd_kernel<TT, NCC>), dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, s,                      \
                     reinterpret_cast<const TT*>(value), offaw, ld_offaw, ref, shapes, starts, reinterpret_cast<TT*>(out),      \
                     reinterpret_cast<TT*>(out_lo), B, \
                     Lq, Lin, M, L, P, Dh)
  if (dtype == ASIS_F16) { if (nc == 2) ASIS_MSDA_FWD(f16, 2); else ASIS_MSDA_FWD(f16, 1); }
  else { if (nc == 2) ASIS_MSDA_FWD(bf16, 2); else ASIS_MSDA_FWD(bf16, 1); }
#undef ASIS_MSDA_FWD
  ASIS_CHECK_LAUNCH("asis_msda_fwd");
  return ASIS_OK;
}

extern "C" int asis_msda_fwd(void* stream, int dtype, const void* value, const float* offaw, int64_t ld_offaw,
                             const float* ref, const int32_t* shapes, const int32_t* starts, void* out, int B, int Lq,
                             int Lin, int M, int L, int P, int Dh) {
  return asis_msda_fwd_split(stream, dtype, value, offaw, ld_offaw, ref, shapes, starts, out, nullptr, B, Lq, Lin, M, L, P, Dh);
}

extern "C" int asis_dwconv_gelu(void* stream, int dtype, const float* x, const float* w9, const float* bias,
                                const int32_t* shapes, const int32_t* starts, int L, void* out, int B, int Ntok,
                                int C) {
  ASIS_REQUIRE(x && w9 && bias && shapes && starts && out, "asis_dwconv_gelu: null pointer");
  ASIS_REQUIRE(C % 4 == 0 && B > 0 && Ntok > 0 && L > 0, "asis_dwconv_gelu: bad shape");
  ASIS_REQUIRE(asis_aligned16(x) && asis_aligned16(w9) && asis_aligned16(bias) && (((uintptr_t)out) & 7) == 0,
               "asis_dwconv_gelu: pointers must be 16-byte aligned");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_dwconv_gelu: bad dtype %d", dtype);
  const int64_t total = (int64_t)B * Ntok * (C / 4);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((dwconv_gelu_kernel<f16>), dim3(grid_for(total)), dim3(256), 0, s, x, w9, bias, shapes, starts,
                       L, reinterpret_cast<f16*>(out), B, Ntok, C);
  else
    hipLaunchKernelGGL((dwconv_gelu_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, x, w9, bias, shapes, starts,
                       L, reinterpret_cast<bf16*>(out), B, Ntok, C);
  ASIS_CHECK_LAUNCH("asis_dwconv_gelu");
  return ASIS_OK;
}
