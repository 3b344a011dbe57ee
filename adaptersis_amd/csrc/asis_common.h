// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of adaptersis_amd.
// Everything here is written for wave64 / MFMA / 160 KiB LDS; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/asis_hip.h"

// ----- error plumbing (C ABI: return code + thread-local message) -------------------------
extern "C" void asis_set_error_(const char* fmt, ...);
#define ASIS_FAIL(code, ...)          \
  do {                                \
    asis_set_error_(__VA_ARGS__);     \
    return (code);                    \
  } while (0)
#define ASIS_REQUIRE(cond, ...)                        \
  do {                                                 \
    if (!(cond)) ASIS_FAIL(ASIS_EINVAL, __VA_ARGS__);  \
  } while (0)
#define ASIS_CHECK_LAUNCH(name)                                                          \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) ASIS_FAIL(ASIS_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

static inline bool asis_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline long asis_cdiv(long a, long b) { return (a + b - 1) / b; }

// ----- 16-bit operand types ----------------------------------------------------------------
typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct T16;
template <> struct T16<f16> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  typedef f16x2 v2;
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct T16<bf16> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  typedef bf16x2 v2;
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ T to_t16(float x) { return (T)x; }  // RNE (fp16 saturates to inf); T = float passes through
// residual of the 16-bit rounding: x ~= (float)hi + (float)lo with ~22 significant bits (split-precision GEMM operands)
template <typename T> __device__ __forceinline__ float lo_part(float x) { return x - (float)((T)x); }
template <typename T> __device__ __forceinline__ float from_t16(T x) { return (float)x; }

// pack two floats into one 32-bit word of two T (lo = first)
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typename T16<T>::v2 v;
  v[0] = (T)a;
  v[1] = (T)b;
  return __builtin_bit_cast(uint32_t, v);
}
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& a, float& b) {
  typename T16<T>::v2 v = __builtin_bit_cast(typename T16<T>::v2, w);
  a = (float)v[0];
  b = (float)v[1];
}

// ----- MX correction operands (block-scaled fp8 MFMA, v_mfma_scale_f32_16x16x128_f8f6f4; layout facts: scripts/mx_probe.hip) ----
// A split-precision product x w ~= xh wh + (xl wh + xh wl) spends two of its three MFMA passes on correction terms that need ~3
// significant bits.  MX form of the "lo" operand: per element TWO fp8 (e4m3) bytes in the place of the one 16-bit rounding
// residual — activations (A side) as (hi8, lo8), weights (B side) as (lo8, hi8) — so that ONE scaled-fp8 MFMA pass over the
// same [rows][K] x 2-byte layout pairs byte with byte: hi8(x) lo8(w) + lo8(x) hi8(w) = both correction terms, at twice the f16
// rate per byte.  hi8 = fp8(v 2^(7 - e)), lo8 = fp8((v - (T)v) 2^(LO - e)), e = floor(log2(amax of the tensor)), LO = 18 for
// f16 (|residual| <= 2^(e - 11)), 15 for bf16: both scaled maxima stay <= 2^8 (e4m3 saturates to NaN above 448: values are
// clamped as well), and the two terms share ONE dequantisation exponent eA + eB - 7 - LO (per-tensor power-of-two scales).
struct MxScale { float sh, sl; };
template <typename T> struct MxLo;
template <> struct MxLo<f16> { static constexpr int v = 18; };
template <> struct MxLo<bf16> { static constexpr int v = 15; };
template <typename T> __device__ __forceinline__ constexpr int mx_lo_shift() { return MxLo<T>::v; }
__device__ __forceinline__ int mx_exp(float amax) {
  int e = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xFF) - 127;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
template <typename T> __device__ __forceinline__ MxScale mx_scales(float amax) {
  const int e = mx_exp(amax);
  MxScale s;
  s.sh = __builtin_bit_cast(float, (uint32_t)(127 + 7 - e) << 23);
  s.sl = __builtin_bit_cast(float, (uint32_t)(127 + mx_lo_shift<T>() - e) << 23);
  return s;
}
// E8M0 scale byte of the A operand of the correction pass (the B operand carries 127 = 1.0)
template <typename T> __device__ __forceinline__ int mx_code(float amax_a, float amax_b) {
  const int c = 127 + mx_exp(amax_a) + mx_exp(amax_b) - 7 - mx_lo_shift<T>();
  return c < 1 ? 1 : (c > 254 ? 254 : c);
}
// two elements -> one 32-bit word of four fp8: (hi8, lo8) pairs, or (lo8, hi8) for the weight side
template <typename T> __device__ __forceinline__ uint32_t mx_pack2(float x0, float x1, MxScale s, bool wside) {
  const float h0 = (float)((T)x0), h1 = (float)((T)x1);
  const float a0 = __builtin_amdgcn_fmed3f(h0 * s.sh, -448.f, 448.f), b0 = __builtin_amdgcn_fmed3f((x0 - h0) * s.sl, -448.f, 448.f);
  const float a1 = __builtin_amdgcn_fmed3f(h1 * s.sh, -448.f, 448.f), b1 = __builtin_amdgcn_fmed3f((x1 - h1) * s.sl, -448.f, 448.f);
  int w = 0;
  w = wside ? __builtin_amdgcn_cvt_pk_fp8_f32(b0, a0, w, false) : __builtin_amdgcn_cvt_pk_fp8_f32(a0, b0, w, false);
  w = wside ? __builtin_amdgcn_cvt_pk_fp8_f32(b1, a1, w, true) : __builtin_amdgcn_cvt_pk_fp8_f32(a1, b1, w, true);
  return (uint32_t)w;
}
// the same from a stored (hi, lo) pair of 16-bit values (no second rounding of hi + lo: the 16-bit part of the GEMM uses hi as stored)
__device__ __forceinline__ uint32_t mx_pack2_pair(float h0, float l0, float h1, float l1, MxScale s, bool wside) {
  const float a0 = __builtin_amdgcn_fmed3f(h0 * s.sh, -448.f, 448.f), b0 = __builtin_amdgcn_fmed3f(l0 * s.sl, -448.f, 448.f);
  const float a1 = __builtin_amdgcn_fmed3f(h1 * s.sh, -448.f, 448.f), b1 = __builtin_amdgcn_fmed3f(l1 * s.sl, -448.f, 448.f);
  int w = 0;
  w = wside ? __builtin_amdgcn_cvt_pk_fp8_f32(b0, a0, w, false) : __builtin_amdgcn_cvt_pk_fp8_f32(a0, b0, w, false);
  w = wside ? __builtin_amdgcn_cvt_pk_fp8_f32(b1, a1, w, true) : __builtin_amdgcn_cvt_pk_fp8_f32(a1, b1, w, true);
  return (uint32_t)w;
}
// the "lo" word of two elements: the 16-bit rounding residuals, or the MX form when ``amax`` is given
template <typename T> __device__ __forceinline__ uint32_t lo_word2(float x0, float x1, const float* amax, bool wside = false) {
  if (amax) return mx_pack2<T>(x0, x1, mx_scales<T>(*amax), wside);
  return pack2<T>(lo_part<T>(x0), lo_part<T>(x1));
}

// ----- wave64 reductions -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane: two quad permutes, row_half_mirror,
// row_mirror — VALU-only (no LDS crossbar); all 64 lanes must be active
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// sum over 8 consecutive lanes (8 g .. 8 g + 7), result in every lane
__device__ __forceinline__ float row8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

// erf-form GELU (nn.GELU() default; dinov2/layers/mlp.py:35, adapter_blocks.py:87).  erf by Abramowitz-Stegun
// 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-epsilon level and 3 orders below the 16-bit operand rounding): one
// v_exp + one v_rcp + 6 FMAs instead of libm erff's ~40 VALU ops, which cost the fc1 epilogue 25 % of the GEMM.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));  // v_rcp_f32 (1 ulp), not the IEEE division sequence
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float y = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(y, x);
}
// gelu(x) = x Phi(x) (erf form, dinov2/layers/mlp.py:35 nn.GELU()).  With a = |x|:
//     1 - Phi(a) = 2^(P(a) - 1),   P = degree-6 polynomial fit of log2(erfc(a / sqrt(2))) on [0, 7]
//     gelu(x) = max(x, 0) - a * 2^(P(a) - 1)
// ONE transcendental (exp2) and 9 full-rate VALU operations per value, no reciprocal: the Abramowitz-Stegun 7.1.26 form it
// replaces (rcp + exp2 + 11 others) made the fc1 epilogue of the one-workgroup-per-CU GEMM form VALU-bound (117 of 170 us
// at M = 42348).  Fitted in the training container by iteratively re-weighted least squares on the ABSOLUTE gelu error;
// float32 evaluation over [-12, 12] and the fp16 range ends: max |error| 5.1e-7 (A&S: 5.2e-7; both are the rounding of
// x - x q near x = 4), relative error with a 1e-3 floor 8.5e-5 (A&S: 1.7e-4).  The argument is clamped at 7, where 1 - Phi is 1.3e-12.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float a = fminf(ax, 7.0f);
  float p = 3.30870223e-05f;
  p = __builtin_fmaf(p, a, -7.69167002e-04f);
  p = __builtin_fmaf(p, a, 8.08053612e-03f);
  p = __builtin_fmaf(p, a, -5.34118121e-02f);
  p = __builtin_fmaf(p, a, -4.58771202e-01f);
  p = __builtin_fmaf(p, a, -1.15120162e+00f);
  p = __builtin_fmaf(p, a, 6.93082119e-06f - 1.0f);
  return __builtin_fmaf(-ax, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}
// SwiGLU gate `dinov2/layers/swiglu_ffn.py:30-34`: silu(a) * b (the asis_swiglu kernel and the ASIS_ACT_SILU_MUL epilogue: one formula)
__device__ __forceinline__ float silu_mul(float a, float b) { return a / (1.f + __expf(-a)) * b; }
// four values at once on the packed-fp32 pipe (v_pk_fma_f32: the Horner steps of two values per instruction)
__device__ __forceinline__ void gelu_erf4(float& x0, float& x1, float& x2, float& x3) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ xa = {x0, x1}, xb = {x2, x3};
  const f32x2_ aa = {fabsf(x0), fabsf(x1)}, ab = {fabsf(x2), fabsf(x3)};
  const f32x2_ ca = __builtin_elementwise_min(aa, (f32x2_){7.0f, 7.0f}), cb = __builtin_elementwise_min(ab, (f32x2_){7.0f, 7.0f});
  f32x2_ pa = {3.30870223e-05f, 3.30870223e-05f}, pb = pa;
#define ASIS_GELU_STEP(K)                                          \
  pa = __builtin_elementwise_fma(pa, ca, (f32x2_){K, K});          \
  pb = __builtin_elementwise_fma(pb, cb, (f32x2_){K, K});
  ASIS_GELU_STEP(-7.69167002e-04f)
  ASIS_GELU_STEP(8.08053612e-03f)
  ASIS_GELU_STEP(-5.34118121e-02f)
  ASIS_GELU_STEP(-4.58771202e-01f)
  ASIS_GELU_STEP(-1.15120162e+00f)
  ASIS_GELU_STEP(6.93082119e-06f - 1.0f)
#undef ASIS_GELU_STEP
  const f32x2_ qa = {__builtin_amdgcn_exp2f(pa.x), __builtin_amdgcn_exp2f(pa.y)};
  const f32x2_ qb = {__builtin_amdgcn_exp2f(pb.x), __builtin_amdgcn_exp2f(pb.y)};
  const f32x2_ ra = __builtin_elementwise_fma(-aa, qa, __builtin_elementwise_max(xa, (f32x2_){0.f, 0.f}));
  const f32x2_ rb = __builtin_elementwise_fma(-ab, qb, __builtin_elementwise_max(xb, (f32x2_){0.f, 0.f}));
  x0 = ra.x; x1 = ra.y; x2 = rb.x; x3 = rb.y;
}
// d/dx gelu = Phi(x) + x phi(x) from the same rcp / exp2 pair as gelu_erf (epilogue-friendly: no libm erff)
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  const float poly = ((((0.5f * 1.061405429f * t - 0.5f * 1.453152027f) * t + 0.5f * 1.421413741f) * t - 0.5f * 0.284496736f) * t +
                      0.5f * 0.254829592f) * t;
  const float e = __builtin_amdgcn_exp2f(-0.5f * 1.4426950408889634f * (x * x));
  const float q = poly * e;
  return (x >= 0.f ? 1.0f - q : q) + x * e * 0.39894228040143267794f;
}
// d/dx gelu
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): block ids that share
// (id % 8) share an L2, so give each XCD a contiguous chunk of the logical tile order.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
