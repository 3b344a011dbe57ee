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
