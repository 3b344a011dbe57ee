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
// gelu(x) = x Phi(x) with Phi from the same A&S erf: for z = x/sqrt(2), q = poly(t) exp(-z^2) / 2 is the smaller tail, so
// gelu = x - x q (x >= 0) or x q (x < 0): no 1 + erf cancellation, 13 VALU ops of which two quarter-rate (rcp, exp2)
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  const float poly = ((((0.5f * 1.061405429f * t - 0.5f * 1.453152027f) * t + 0.5f * 1.421413741f) * t - 0.5f * 0.284496736f) * t +
                      0.5f * 0.254829592f) * t;
  const float r = x * (poly * __builtin_amdgcn_exp2f(-0.5f * 1.4426950408889634f * (x * x)));
  return x >= 0.f ? x - r : r;
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
