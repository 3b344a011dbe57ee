// GPU form of the training-time augmentation pipeline of the reference (`train.py:139-163`, albumentations on uint8
// images inside the DataLoader workers; `tools/dataset.py:150-161`): crop + resize back to S x S, horizontal flip,
// rot90, brightness/contrast and gamma look-up tables, uint8 -> float / 255, mask with nearest sampling.
// One kernel, one pass: every output pixel gathers its four source pixels through the inverse geometric map.
// All arithmetic is integer (OpenCV's 8-bit INTER_LINEAR fixed-point form with 11-bit coefficients), so the result is
// bit-identical to the numpy restatement in oracle/augment_ref.py by construction.
#include "asis_common.h"

namespace {

// per-sample geometry + tables (built on the host by adaptersis_amd/tools/augment.py)
//   geo[b] = {flip, rotk, identity, 0}
//   xofs/yofs [B, S] int32: left / top source index of the bilinear pair (already offset by the crop origin)
//   xa/ya [B, S, 2] int16: the two 11-bit coefficients (alpha0, alpha1), as OpenCV rounds them separately
//   mx/my [B, S] int32: nearest-neighbour source index for the mask
//   lut [B, 256] uint8: brightness/contrast followed by gamma
__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ msk,
                                                      const int4* __restrict__ geo, const int* __restrict__ xofs,
                                                      const int* __restrict__ yofs, const short2* __restrict__ xa,
                                                      const short2* __restrict__ ya, const int* __restrict__ mx,
                                                      const int* __restrict__ my, const uint8_t* __restrict__ lut,
                                                      float* __restrict__ out, int64_t* __restrict__ mout, int S) {
  __shared__ uint8_t s_lut[256];
  const int b = blockIdx.y;
  s_lut[threadIdx.x] = lut[b * 256 + threadIdx.x];
  __syncthreads();
  const int4 g = geo[b];
  const int64_t plane = (int64_t)S * S;
  const uint8_t* im = img + (int64_t)b * plane * 3;
  const uint8_t* mk = msk + (int64_t)b * plane;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (int64_t)gridDim.x * blockDim.x) {
    const int oy = (int)(p / S), ox = (int)(p - (int64_t)oy * S);
    // undo np.rot90(m, k): k = 1: out[i][j] = m[j][S-1-i]; 2: m[S-1-i][S-1-j]; 3: m[S-1-j][i]
    int y = oy, x = ox;
    if (g.y == 1) { y = ox; x = S - 1 - oy; }
    else if (g.y == 2) { y = S - 1 - oy; x = S - 1 - ox; }
    else if (g.y == 3) { y = S - 1 - ox; x = oy; }
    if (g.x) x = S - 1 - x;                       // undo the horizontal flip
    int v0, v1, v2, mv;
    if (g.z) {                                    // no crop: the resize is the identity
      const uint8_t* q = im + ((int64_t)y * S + x) * 3;
      v0 = q[0]; v1 = q[1]; v2 = q[2];
      mv = mk[(int64_t)y * S + x];
    } else {
      const int sx = xofs[b * S + x], sy = yofs[b * S + y];
      const int sx1 = sx + 1 < S ? sx + 1 : S - 1, sy1 = sy + 1 < S ? sy + 1 : S - 1;   // coefficient of the clamped neighbour is 0
      const short2 ax = xa[b * S + x], ay = ya[b * S + y];
      const uint8_t* r0 = im + ((int64_t)sy * S) * 3;
      const uint8_t* r1 = im + ((int64_t)sy1 * S) * 3;
      int res[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = r0[sx * 3 + c] * ax.x + r0[sx1 * 3 + c] * ax.y;        // horizontal pass, 11 fractional bits
        const int h1 = r1[sx * 3 + c] * ax.x + r1[sx1 * 3 + c] * ax.y;
        const int t = (((int)ay.x * (h0 >> 4)) >> 16) + (((int)ay.y * (h1 >> 4)) >> 16);
        int r = (t + 2) >> 2;
        res[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
      }
      v0 = res[0]; v1 = res[1]; v2 = res[2];
      mv = mk[(int64_t)my[b * S + y] * S + mx[b * S + x]];
    }
    float* o = out + (int64_t)b * 3 * plane + p;
    o[0] = (float)s_lut[v0] / 255.0f;
    o[plane] = (float)s_lut[v1] / 255.0f;
    o[2 * plane] = (float)s_lut[v2] / 255.0f;
    mout[(int64_t)b * plane + p] = (int64_t)mv;
  }
}

}  // namespace

extern "C" int asis_augment(void* stream, const uint8_t* img, const uint8_t* mask, const int32_t* geo, const int32_t* xofs,
                            const int32_t* yofs, const int16_t* xa, const int16_t* ya, const int32_t* mx, const int32_t* my,
                            const uint8_t* lut, float* out, int64_t* mask_out, int B, int S) {
  ASIS_REQUIRE(img && mask && geo && xofs && yofs && xa && ya && mx && my && lut && out && mask_out, "asis_augment: null pointer");
  ASIS_REQUIRE(B >= 1 && S >= 2 && S <= 8192, "asis_augment: bad batch / size");
  ASIS_REQUIRE(asis_aligned16(geo) && (reinterpret_cast<uintptr_t>(xa) & 3) == 0 && (reinterpret_cast<uintptr_t>(ya) & 3) == 0,
               "asis_augment: table alignment");
  const int64_t plane = (int64_t)S * S;
  int gx = (int)((plane + 255) / 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(augment_kernel, dim3(gx, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, mask,
                     reinterpret_cast<const int4*>(geo), xofs, yofs, reinterpret_cast<const short2*>(xa),
                     reinterpret_cast<const short2*>(ya), mx, my, lut, out, reinterpret_cast<int64_t*>(mask_out), S);
  ASIS_CHECK_LAUNCH("asis_augment");
  return ASIS_OK;
}

// =====================================================================================================================
// CLAHE stage of the pipeline (`train.py:161`: A.CLAHE(p=0.8) = OpenCV RGB -> Lab (8-bit integer path), tiled
// contrast-limited histogram equalisation of L on an 8 x 8 grid, Lab -> RGB (8-bit integer path)), between the geometric
// stage and the brightness / gamma tables.  It needs whole-image tile histograms of the geometrically transformed image, so
// a batch with at least one CLAHE sample runs three kernels instead of one:
//   augment_geo_u8   the geometric stage of augment_kernel, result kept as uint8 RGB (+ the mask, final)
//   clahe_lut        one workgroup per (sample, tile): L of the tile's pixels (reflect-101 padding when S % 8 != 0) ->
//                    LDS histogram -> clip at the sample's integer limit, redistribute, cumulative sum -> 256-entry tile LUT
//   clahe_apply      per pixel: RGB -> Lab, L through the bilinear blend of the four neighbouring tile LUTs (float, the
//                    products and sums rounded one by one like the scalar C++), Lab -> RGB, brightness/gamma LUT, / 255
// All colour arithmetic is integer on the look-up tables of OpenCV's initLabTabs (built on the host,
// adaptersis_amd/tools/clahe.py) and is bit-identical to oracle/augment_ref.py by construction.
// =====================================================================================================================
namespace {

struct lab_tabs {
  const uint16_t* gamma;     // [256]   sRGBGammaTab_b
  const uint16_t* cbrt;      // [3072]  LabCbrtTab_b
  const uint16_t* l2yf;      // [256][2] LabToYF_b
  const int32_t* ab2xz;      // [36864] abToXZ_b (index v - minABvalue)
  const uint8_t* invgamma;   // [4096]  sRGBInvGammaTab_b
  int32_t fwd[9];            // RGB2Lab_b coefficients (rows X, Y, Z over R, G, B), 12 fractional bits
  int32_t inv[9];            // Lab2RGBinteger coefficients (rows R, G, B over X, Y, Z)
};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ void rgb2lab(const lab_tabs& t, int r, int g, int b, int& L, int& a, int& bb) {
  const int R = t.gamma[r], G = t.gamma[g], B = t.gamma[b];
  const int fX = t.cbrt[descale(R * t.fwd[0] + G * t.fwd[1] + B * t.fwd[2], 12)];
  const int fY = t.cbrt[descale(R * t.fwd[3] + G * t.fwd[4] + B * t.fwd[5], 12)];
  const int fZ = t.cbrt[descale(R * t.fwd[6] + G * t.fwd[7] + B * t.fwd[8], 12)];
  constexpr int Lscale = (116 * 255 + 50) / 100;
  constexpr int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
  L = clamp255(descale(Lscale * fY + Lshift, 15));
  a = clamp255(descale(500 * (fX - fY) + 128 * (1 << 15), 15));
  bb = clamp255(descale(200 * (fY - fZ) + 128 * (1 << 15), 15));
}

__device__ __forceinline__ int rgb2L(const lab_tabs& t, int r, int g, int b) {
  const int R = t.gamma[r], G = t.gamma[g], B = t.gamma[b];
  const int fY = t.cbrt[descale(R * t.fwd[3] + G * t.fwd[4] + B * t.fwd[5], 12)];
  constexpr int Lscale = (116 * 255 + 50) / 100;
  constexpr int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
  return clamp255(descale(Lscale * fY + Lshift, 15));
}

__device__ __forceinline__ void lab2rgb(const lab_tabs& t, int L, int a, int b, int& r, int& g, int& bl) {
  constexpr int BASE = 1 << 14, MIN_AB = -8145;
  const int y = t.l2yf[2 * L], ify = t.l2yf[2 * L + 1];
  const int adiv = ((5 * a * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
  const int bdiv = ((b * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
  const int x = t.ab2xz[ify + adiv - MIN_AB];
  const int z = t.ab2xz[ify - bdiv - MIN_AB];
  int v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int w = descale(t.inv[3 * c] * x + t.inv[3 * c + 1] * y + t.inv[3 * c + 2] * z, 14);
    w = w < 0 ? 0 : (w > 4095 ? 4095 : w);
    v[c] = t.invgamma[w];
  }
  r = v[0]; g = v[1]; bl = v[2];
}

// the geometric stage alone: uint8 RGB [B,S,S,3] out, mask int64 out (final)
__global__ __launch_bounds__(256) void augment_geo_u8_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ msk,
                                                             const int4* __restrict__ geo, const int* __restrict__ xofs,
                                                             const int* __restrict__ yofs, const short2* __restrict__ xa,
                                                             const short2* __restrict__ ya, const int* __restrict__ mx,
                                                             const int* __restrict__ my, uint8_t* __restrict__ out,
                                                             int64_t* __restrict__ mout, int S) {
  const int b = blockIdx.y;
  const int4 g = geo[b];
  const int64_t plane = (int64_t)S * S;
  const uint8_t* im = img + (int64_t)b * plane * 3;
  const uint8_t* mk = msk + (int64_t)b * plane;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (int64_t)gridDim.x * blockDim.x) {
    const int oy = (int)(p / S), ox = (int)(p - (int64_t)oy * S);
    int y = oy, x = ox;
    if (g.y == 1) { y = ox; x = S - 1 - oy; }
    else if (g.y == 2) { y = S - 1 - oy; x = S - 1 - ox; }
    else if (g.y == 3) { y = S - 1 - ox; x = oy; }
    if (g.x) x = S - 1 - x;
    int v0, v1, v2, mv;
    if (g.z) {
      const uint8_t* q = im + ((int64_t)y * S + x) * 3;
      v0 = q[0]; v1 = q[1]; v2 = q[2];
      mv = mk[(int64_t)y * S + x];
    } else {
      const int sx = xofs[b * S + x], sy = yofs[b * S + y];
      const int sx1 = sx + 1 < S ? sx + 1 : S - 1, sy1 = sy + 1 < S ? sy + 1 : S - 1;
      const short2 ax = xa[b * S + x], ay = ya[b * S + y];
      const uint8_t* r0 = im + ((int64_t)sy * S) * 3;
      const uint8_t* r1 = im + ((int64_t)sy1 * S) * 3;
      int res[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = r0[sx * 3 + c] * ax.x + r0[sx1 * 3 + c] * ax.y;
        const int h1 = r1[sx * 3 + c] * ax.x + r1[sx1 * 3 + c] * ax.y;
        const int t = (((int)ay.x * (h0 >> 4)) >> 16) + (((int)ay.y * (h1 >> 4)) >> 16);
        res[c] = clamp255((t + 2) >> 2);
      }
      v0 = res[0]; v1 = res[1]; v2 = res[2];
      mv = mk[(int64_t)my[b * S + y] * S + mx[b * S + x]];
    }
    uint8_t* o = out + ((int64_t)b * plane + p) * 3;
    o[0] = (uint8_t)v0; o[1] = (uint8_t)v1; o[2] = (uint8_t)v2;
    mout[(int64_t)b * plane + p] = (int64_t)mv;
  }
}

// one workgroup per (tile, sample): luts[b][ty][tx][256]
__global__ __launch_bounds__(256) void clahe_lut_kernel(const uint8_t* __restrict__ rgb, const int2* __restrict__ clahe,
                                                        const lab_tabs t, uint8_t* __restrict__ luts, int S, int tiles, int ts) {
#pragma clang fp contract(off)
  __shared__ int hist[256];
  __shared__ int scan[256];
  __shared__ int red[4];
  const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int2 cl = clahe[b];
  if (!cl.x) return;                                   // no CLAHE on this sample (whole workgroup)
  const int ty = tile / tiles, tx = tile - ty * tiles;
  hist[tid] = 0;
  __syncthreads();
  const uint8_t* im = rgb + (int64_t)b * S * S * 3;
  for (int p = tid; p < ts * ts; p += 256) {
    const int py = p / ts, px = p - py * ts;
    int y = ty * ts + py, x = tx * ts + px;
    if (y >= S) y = 2 * S - 2 - y;                     // BORDER_REFLECT_101 of copyMakeBorder (bottom / right only)
    if (x >= S) x = 2 * S - 2 - x;
    const uint8_t* q = im + ((int64_t)y * S + x) * 3;
    atomicAdd(&hist[rgb2L(t, q[0], q[1], q[2])], 1);
  }
  __syncthreads();
  // clip, count the clipped pixels
  int h = hist[tid];
  int over = h > cl.y ? h - cl.y : 0;
  h -= over;
  int s = over;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const int clipped = red[0] + red[1] + red[2] + red[3];
  const int batch = clipped / 256;
  int residual = clipped - batch * 256;
  h += batch;
  if (residual) {                                      // for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++
    const int step = 256 / residual > 1 ? 256 / residual : 1;
    if (tid % step == 0 && tid / step < residual) ++h;
  }
  // inclusive scan over the 256 bins (Hillis-Steele in LDS)
  scan[tid] = h;
  __syncthreads();
#pragma unroll
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  const float lut_scale = __fdiv_rn(255.0f, (float)(ts * ts));   // correctly rounded, like the host's float division
  const int v = __float2int_rn((float)scan[tid] * lut_scale);             // saturate_cast<uchar>(sum * lutScale): cvRound
  luts[(((int64_t)b * tiles + ty) * tiles + tx) * 256 + tid] = (uint8_t)clamp255(v);
}

__global__ __launch_bounds__(256) void clahe_apply_kernel(const uint8_t* __restrict__ rgb, const int2* __restrict__ clahe,
                                                          const lab_tabs t, const uint8_t* __restrict__ luts,
                                                          const uint8_t* __restrict__ lut, float* __restrict__ out, int S,
                                                          int tiles, int ts) {
#pragma clang fp contract(off)   // OpenCV's scalar C++ rounds every product and sum of the blend; hipcc would fuse them into FMAs
  __shared__ uint8_t s_lut[256];
  const int b = blockIdx.y;
  s_lut[threadIdx.x] = lut[b * 256 + threadIdx.x];
  __syncthreads();
  const int2 cl = clahe[b];
  const int64_t plane = (int64_t)S * S;
  const uint8_t* im = rgb + (int64_t)b * plane * 3;
  const uint8_t* lt = luts + (int64_t)b * tiles * tiles * 256;
  const float inv_t = __fdiv_rn(1.0f, (float)ts);
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / S), x = (int)(p - (int64_t)y * S);
    const uint8_t* q = im + p * 3;
    int r = q[0], g = q[1], bl = q[2];
    if (cl.x) {
      int L, a, bb;
      rgb2lab(t, r, g, bl, L, a, bb);
      // plain operators, NOT the __f*_rn wrappers: those are inline functions of the HIP headers and carry the headers' own
      // fp-contract state into the call site — their products and sums were fused into FMAs in spite of the pragma above
      const float tyf = (float)y * inv_t - 0.5f, txf = (float)x * inv_t - 0.5f;
      int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
      const float ya = tyf - (float)ty1, xa = txf - (float)tx1;
      const float ya1 = 1.0f - ya, xa1 = 1.0f - xa;
      int ty2 = ty1 + 1, tx2 = tx1 + 1;
      ty1 = ty1 < 0 ? 0 : ty1; tx1 = tx1 < 0 ? 0 : tx1;
      ty2 = ty2 > tiles - 1 ? tiles - 1 : ty2; tx2 = tx2 > tiles - 1 ? tiles - 1 : tx2;
      const float l11 = (float)lt[(ty1 * tiles + tx1) * 256 + L], l12 = (float)lt[(ty1 * tiles + tx2) * 256 + L];
      const float l21 = (float)lt[(ty2 * tiles + tx1) * 256 + L], l22 = (float)lt[(ty2 * tiles + tx2) * 256 + L];
      const float top = l11 * xa1 + l12 * xa;
      const float bot = l21 * xa1 + l22 * xa;
      const float res = top * ya1 + bot * ya;
      L = clamp255(__float2int_rn(res));
      lab2rgb(t, L, a, bb, r, g, bl);
    }
    float* o = out + (int64_t)b * 3 * plane + p;
    o[0] = (float)s_lut[r] / 255.0f;
    o[plane] = (float)s_lut[g] / 255.0f;
    o[2 * plane] = (float)s_lut[bl] / 255.0f;
  }
}

}  // namespace

extern "C" int asis_augment_geo_u8(void* stream, const uint8_t* img, const uint8_t* mask, const int32_t* geo, const int32_t* xofs,
                                   const int32_t* yofs, const int16_t* xa, const int16_t* ya, const int32_t* mx, const int32_t* my,
                                   uint8_t* out_u8, int64_t* mask_out, int B, int S) {
  ASIS_REQUIRE(img && mask && geo && xofs && yofs && xa && ya && mx && my && out_u8 && mask_out, "asis_augment_geo_u8: null pointer");
  ASIS_REQUIRE(B >= 1 && S >= 2 && S <= 8192, "asis_augment_geo_u8: bad batch / size");
  ASIS_REQUIRE(asis_aligned16(geo) && (reinterpret_cast<uintptr_t>(xa) & 3) == 0 && (reinterpret_cast<uintptr_t>(ya) & 3) == 0,
               "asis_augment_geo_u8: table alignment");
  const int64_t plane = (int64_t)S * S;
  int gx = (int)((plane + 255) / 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(augment_geo_u8_kernel, dim3(gx, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, mask,
                     reinterpret_cast<const int4*>(geo), xofs, yofs, reinterpret_cast<const short2*>(xa),
                     reinterpret_cast<const short2*>(ya), mx, my, out_u8, reinterpret_cast<int64_t*>(mask_out), S);
  ASIS_CHECK_LAUNCH("asis_augment_geo_u8");
  return ASIS_OK;
}

extern "C" int asis_clahe(void* stream, const uint8_t* rgb, const int32_t* clahe, const uint16_t* tab_gamma, const uint16_t* tab_cbrt,
                          const uint16_t* tab_l2yf, const int32_t* tab_ab2xz, const uint8_t* tab_invgamma, const int32_t* coef_fwd,
                          const int32_t* coef_inv, uint8_t* luts, const uint8_t* lut, float* out, int B, int S, int tiles) {
  ASIS_REQUIRE(rgb && clahe && tab_gamma && tab_cbrt && tab_l2yf && tab_ab2xz && tab_invgamma && coef_fwd && coef_inv && luts && lut && out,
               "asis_clahe: null pointer");
  ASIS_REQUIRE(B >= 1 && S >= 16 && S <= 8192 && tiles >= 1 && tiles <= 16, "asis_clahe: bad batch / size / tile grid");
  ASIS_REQUIRE((reinterpret_cast<uintptr_t>(clahe) & 7) == 0, "asis_clahe: clahe table alignment");
  lab_tabs t;
  t.gamma = tab_gamma; t.cbrt = tab_cbrt; t.l2yf = tab_l2yf; t.ab2xz = tab_ab2xz; t.invgamma = tab_invgamma;
  for (int i = 0; i < 9; ++i) { t.fwd[i] = coef_fwd[i]; t.inv[i] = coef_inv[i]; }   // host pointers: 9 + 9 ints
  const int ts = (S % tiles == 0) ? S / tiles : (S + tiles - S % tiles) / tiles;     // tile edge of the (padded) plane
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles * tiles, B), dim3(256), 0, s, rgb, reinterpret_cast<const int2*>(clahe), t, luts,
                     S, tiles, ts);
  const int64_t plane = (int64_t)S * S;
  int gx = (int)((plane + 255) / 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(clahe_apply_kernel, dim3(gx, B), dim3(256), 0, s, rgb, reinterpret_cast<const int2*>(clahe), t, luts, lut, out,
                     S, tiles, ts);
  ASIS_CHECK_LAUNCH("asis_clahe");
  return ASIS_OK;
}
