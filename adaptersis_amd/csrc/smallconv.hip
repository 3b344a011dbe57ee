// Direct 3x3 convolution (stride 1, pad 1) for layers with a handful of output channels — the decode
// heads' final classifier conv (backbones/decoders.py:135: 64 -> num_classes at 672x672) and its input
// gradient.  As an MFMA implicit GEMM these layers waste a 128-wide tile on 2..11 columns (the forward
// cost 2.3 ms per pass at B=12); they are L2/HBM-bound gathers with ~1 FLOP per byte, so they run on the
// vector ALUs in fp32: x is rebuilt as hi+lo (exact to ~22 bits), weights stay fp32, every lane owns one
// pixel (forward) or one pixel x 8 input channels (dgrad), weights are broadcast from LDS.
#include "asis_common.h"

namespace {

constexpr int MAXCO = 16;
// 16 zero bytes: LDS-DMA source of halo pixels outside the image / of the absent lo half
__device__ __attribute__((aligned(16))) uint4 g_zero_page_sc[1];

template <typename T>
__device__ __forceinline__ void load8(const T* p, float* f) {
  const uint4 raw = *reinterpret_cast<const uint4*>(p);
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&raw);
#pragma unroll
  for (int e = 0; e < 4; ++e) unpack2<T>(w[e], f[2 * e], f[2 * e + 1]);
}

// out[b,y,x,co] = bias[co] + sum_{kh,kw,ci} (xh+xl)[b,y+kh-1,x+kw-1,ci] * w[co,ci,kh,kw]
// thread = (pixel, 8-channel chunk): the Cin/8 lanes of a pixel read its 16-byte chunks side by side (full
// lines), accumulate their channels over the 9 taps, and are summed with xor-shuffles (Cin/8 a power of two).
template <typename T, int CO>
__global__ __launch_bounds__(256) void smallcout_fwd_kernel(const T* __restrict__ xh, const T* __restrict__ xl,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int B, int H, int W, int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [tap][ci][CO]
  for (int i = threadIdx.x; i < 9 * Cin * CO; i += blockDim.x) {
    const int co = i % CO, ci = (i / CO) % Cin, tap = i / (CO * Cin);
    wl[i] = co < Cout ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
  }
  __syncthreads();
  const int cpp = Cin >> 3;  // lanes per pixel (power of two, <= 64)
  const int64_t total = (int64_t)B * H * W * cpp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // every lane of a wave runs the same number of iterations (total and stride are multiples of 64)
  for (int64_t i = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i & (cpp - 1)) * 8;
    const int64_t p = i / cpp;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    // all taps are fetched up front from coordinates clamped into the image and an outside tap is zeroed afterwards:
    // a load under `if (inside)` is waited for on the spot, i.e. up to 18 exposed latencies per pixel
    uint4 rh[9], rl[9];
    float msk[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int t = kh * 3 + kw;
        const int yy = y + kh - 1, xx = x + kw - 1;
        msk[t] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? 1.f : 0.f;
        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const int64_t q = p + (int64_t)(yc - y) * W + (xc - x);
        rh[t] = *reinterpret_cast<const uint4*>(xh + q * Cin + c8);
        rl[t] = xl ? *reinterpret_cast<const uint4*>(xl + q * Cin + c8) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float f[8], g[8];
      unpack2<T>(rh[t].x, f[0], f[1]); unpack2<T>(rh[t].y, f[2], f[3]); unpack2<T>(rh[t].z, f[4], f[5]); unpack2<T>(rh[t].w, f[6], f[7]);
      unpack2<T>(rl[t].x, g[0], g[1]); unpack2<T>(rl[t].y, g[2], g[3]); unpack2<T>(rl[t].z, g[4], g[5]); unpack2<T>(rl[t].w, g[6], g[7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (f[e] + g[e]) * msk[t];
      const float* wt = wl + (t * Cin + c8) * CO;
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] += f[e] * wt[e * CO + c];
    }
    for (int o = 1; o < cpp; o <<= 1)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
    if (c8 == 0)
      for (int c = 0; c < Cout; ++c) out[p * Cout + c] = acc[c] + (bias ? bias[c] : 0.f);
  }
}

// ---- BatchNorm + ReLU + bilinear x2 (align_corners) of the previous stage's raw conv output, evaluated ON LOAD -------------------
// (`decoders.py:131-135`: ... BatchNorm2d, ReLU, Upsample(2, bilinear, align_corners=True), Conv2d(64, classes, 3, padding=1)).  The
// classifier's input is 4x the bytes of the raw map it is made of (12 x 672^2 x 64 channels as hi + lo planes: 1.39 GB written by
// asis_bn_relu_upsample and read back by the forward conv, the hi plane once more by the weight gradient); here the halo tile of
// the upsampled map is computed from the raw fp32 map (0.35 GB, re-read from L2 by neighbouring tiles) while it is staged.
// Same tap order and weights as bn_relu_upsample8_kernel (convmisc.hip).
struct UpSrc {
  const float* raw;      // fp32 NHWC [B, H, W, 64] (low resolution)
  const float* scale;    // BatchNorm scale / shift per channel
  const float* shift;
  int H, W;              // low-resolution size; the conv runs on [2H, 2W]
};
// Two phases per tile, so that no halo value waits on a global load: (A) relu(raw * scale + shift) of the low-resolution region under
// the tile's halo (at most UP_RY x UP_RX source pixels: 10 / 18 high-resolution rows / columns map to <= 6 / 10 source rows / columns
// plus the +1 taps) -> fp32 in LDS; (B) every halo value = the four-tap blend of that region, read from LDS.
constexpr int UP_RY = 8, UP_RX = 12;
__device__ __forceinline__ void up_stage_lowres(const UpSrc& u, int b, int r0, int c0, float* __restrict__ lowr, int tid) {
  constexpr int C = 64;
  for (int it = tid; it < UP_RY * UP_RX * (C / 4); it += 256) {
    const int c4 = it & 15, px = it >> 4;
    const int ry = px / UP_RX, rx = px - ry * UP_RX;
    const int sy = min(r0 + ry, u.H - 1), sx = min(c0 + rx, u.W - 1);
    const float4 v = reinterpret_cast<const float4*>(u.raw + (((int64_t)b * u.H + sy) * u.W + sx) * C)[c4];
    const float4 sc = reinterpret_cast<const float4*>(u.scale)[c4], sh = reinterpret_cast<const float4*>(u.shift)[c4];
    reinterpret_cast<float4*>(lowr + px * C)[c4] = make_float4(fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f),
                                                               fmaxf(v.z * sc.z + sh.z, 0.f), fmaxf(v.w * sc.w + sh.w, 0.f));
  }
}
__device__ __forceinline__ void up8_blend(const UpSrc& u, const float* __restrict__ lowr, int r0, int c0, int oh, int ow, int c8,
                                          float rh, float rw, float acc[8]) {
  constexpr int C = 64;
  const float w1r = rw * ow;
  const int w1 = (int)w1r;
  const int w1p = (w1 < u.W - 1) ? 1 : 0;
  const float wl = w1r - w1;
  const float h1r = rh * oh;
  const int h1 = (int)h1r;
  const int h1p = (h1 < u.H - 1) ? 1 : 0;
  const float hl = h1r - h1;
  const int ly = min(max(h1 - r0, 0), UP_RY - 2), lx = min(max(w1 - c0, 0), UP_RX - 2);   // inside the staged region by construction
  const float* base = lowr + (ly * UP_RX + lx) * C + 8 * c8;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int dh = (t >> 1) ? h1p : 0, dw = (t & 1) ? w1p : 0;
      const float4 v = reinterpret_cast<const float4*>(base + (dh * UP_RX + dw) * C)[h];
      const float wt = ((t >> 1) ? hl : 1.f - hl) * ((t & 1) ? wl : 1.f - wl);
      a.x += wt * v.x; a.y += wt * v.y; a.z += wt * v.z; a.w += wt * v.w;
    }
    acc[4 * h + 0] = a.x; acc[4 * h + 1] = a.y; acc[4 * h + 2] = a.z; acc[4 * h + 3] = a.w;
  }
}

// The halo pixel record is 16 chunks of 16 bytes (8 hi | 8 lo) with the chunk index XORed with the pixel's low bits: the
// 16 lanes of one A-fragment column group read the same chunk of 16 consecutive pixels -> 16 different bank groups.
template <typename T, bool UP = false>
__global__ __launch_bounds__(256, 2) void smallcout_fwd_mfma_kernel(const T* __restrict__ xh, const T* __restrict__ xl,
                                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                                    float* __restrict__ out, int B, int H, int W, int Cout,
                                                                    const UpSrc up = UpSrc{}) {
  constexpr int CIN = 64, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, NPIX = HY * HX;   // 180 halo pixels
  typedef typename T16<T>::v8 v8;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  __shared__ __attribute__((aligned(16))) T tile[NPIX * 2 * CIN];   // [pixel][16 chunks, swizzled][8]
  __shared__ __attribute__((aligned(16))) float lowr[UP ? UP_RY * UP_RX * CIN : 4];   // UP: relu(bn(raw)) of the region under the halo
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fn = lane & 15, fq = lane >> 4;   // MFMA fragment coordinates: A row (pixel) / B, D column (class); K group
  // B fragments: W[class fn][channel 32 ks + 8 fq + j][tap] as hi and rounding residual, zero for classes >= Cout
  v8 wh[9][2], wl[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = fn < Cout ? w[((int64_t)fn * CIN + 32 * ks + 8 * fq + j) * 9 + t] : 0.f;
        wh[t][ks][j] = (T)v;
        wl[t][ks][j] = (T)lo_part<T>(v);
      }
  const float bn = (bias && fn < Cout) ? bias[fn] : 0.f;
  const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
  const int ntiles = B * tiles_y * tiles_x;
  const T* const zp = reinterpret_cast<const T*>(g_zero_page_sc);
  const int dp = lane >> 4, dslot = lane & 15;   // DMA lane map: one instruction = 4 halo pixels x 16 chunk slots
  for (int tl = xcd_remap(blockIdx.x, gridDim.x); tl < ntiles; tl += gridDim.x) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int y0 = ty * TY - 1, x0 = tx * TX - 1;
    if constexpr (UP) {                               // the halo tile of the upsampled map, computed from the raw low-resolution map
      const float rh = H > 1 ? (float)(up.H - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(up.W - 1) / (float)(W - 1) : 0.f;
      const int r0 = (int)(rh * max(y0, 0)), c0 = (int)(rw * max(x0, 0));   // first source row / column under the halo
      up_stage_lowres(up, b, r0, c0, lowr, tid);
      __syncthreads();
      for (int it = tid; it < NPIX * 8; it += 256) {  // (halo pixel, 8-channel group): hi chunk c8 and lo chunk 8 + c8 of the record
        const int hp = it >> 3, c8 = it & 7;
        const int hy = hp / HX, hx = hp - hy * HX;
        const int yy = y0 + hy, xx = x0 + hx;
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) up8_blend(up, lowr, r0, c0, yy, xx, c8, rh, rw, a8);
        uint4 oh, ol;
        oh.x = pack2<T>(a8[0], a8[1]); oh.y = pack2<T>(a8[2], a8[3]); oh.z = pack2<T>(a8[4], a8[5]); oh.w = pack2<T>(a8[6], a8[7]);
        ol.x = pack2<T>(lo_part<T>(a8[0]), lo_part<T>(a8[1])); ol.y = pack2<T>(lo_part<T>(a8[2]), lo_part<T>(a8[3]));
        ol.z = pack2<T>(lo_part<T>(a8[4]), lo_part<T>(a8[5])); ol.w = pack2<T>(lo_part<T>(a8[6]), lo_part<T>(a8[7]));
        T* rec = tile + hp * 2 * CIN;
        *reinterpret_cast<uint4*>(rec + ((c8 ^ (hp & 15)) << 3)) = oh;
        *reinterpret_cast<uint4*>(rec + (((c8 + 8) ^ (hp & 15)) << 3)) = ol;
      }
    } else {
    for (int g = wid; g * 4 < NPIX; g += 4) {       // 45 groups of 4 pixels over the 4 waves
      const int hp = g * 4 + dp;
      const int hy = hp / HX, hx = hp - hy * HX;
      const int yy = y0 + hy, xx = x0 + hx;
      const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const int ch = dslot ^ (hp & 15);             // the chunk this slot holds: 0..7 hi, 8..15 lo
      const bool want_lo = ch >= 8;
      const T* src = zp;
      if (in && (!want_lo || xl)) src = (want_lo ? xl : xh) + (((int64_t)b * H + yy) * W + xx) * CIN + (ch & 7) * 8;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(tile + g * 4 * 2 * CIN), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int py = wid * 2 + rr;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int t = kh * 3 + kw;
          const int hp = (py + kh) * HX + fn + kw;     // halo pixel of this lane's A row
          const T* rec = tile + hp * 2 * CIN;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int ch = 4 * ks + fq;
            const v8 ah = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(rec + ((ch ^ (hp & 15)) << 3)));
            const v8 al = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(rec + (((ch + 8) ^ (hp & 15)) << 3)));
            acc = T16<T>::mfma16(ah, wh[t][ks], acc);
            acc = T16<T>::mfma16(al, wh[t][ks], acc);
            acc = T16<T>::mfma16(ah, wl[t][ks], acc);
          }
          // 144 registers hold the weights: keep the fragment reads of at most one tap in flight ahead of their MFMAs
          __builtin_amdgcn_sched_barrier(0);
        }
      const int oy = ty * TY + py;
      if (fn < Cout && oy < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ox = tx * TX + 4 * fq + j;
          if (ox < W) out[(((int64_t)b * H + oy) * W + ox) * Cout + fn] = acc[j] + bn;
        }
      }
    }
    __syncthreads();   // the tile is overwritten by the next DMA
  }
}

// dx[b,y,x,ci] = sum_{kh,kw,co} (dh+dl)[b, y-(kh-1), x-(kw-1), co] * w[co,ci,kh,kw]; dy has CoP (>= Cout) channels
template <typename T, int CO>
__global__ __launch_bounds__(256) void smallcout_dgrad_kernel(const T* __restrict__ dh, const T* __restrict__ dl, int CoP,
                                                              const float* __restrict__ w, float* __restrict__ dx, int B,
                                                              int H, int W, int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [tap][co][ci]
  for (int i = threadIdx.x; i < 9 * CO * Cin; i += blockDim.x) {
    const int ci = i % Cin, co = (i / Cin) % CO, tap = i / (Cin * CO);
    wl[i] = co < Cout ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
  }
  __syncthreads();
  const int cpp = Cin >> 3;
  const int64_t total = (int64_t)B * H * W * cpp;
  for (int64_t i = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cpp) * 8;
    const int64_t p = i / cpp;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    uint4 rh[9], rl[9];
    float msk[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int t = kh * 3 + kw;
        const int yy = y - (kh - 1), xx = x - (kw - 1);
        msk[t] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? 1.f : 0.f;
        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const int64_t q = p + (int64_t)(yc - y) * W + (xc - x);
        rh[t] = *reinterpret_cast<const uint4*>(dh + q * CoP);  // CO <= 8 channels live in the first 16 bytes
        rl[t] = dl ? *reinterpret_cast<const uint4*>(dl + q * CoP) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float d[8], g[8];
      unpack2<T>(rh[t].x, d[0], d[1]); unpack2<T>(rh[t].y, d[2], d[3]); unpack2<T>(rh[t].z, d[4], d[5]); unpack2<T>(rh[t].w, d[6], d[7]);
      unpack2<T>(rl[t].x, g[0], g[1]); unpack2<T>(rl[t].y, g[2], g[3]); unpack2<T>(rl[t].z, g[4], g[5]); unpack2<T>(rl[t].w, g[6], g[7]);
      const float* wt = wl + t * CO * Cin + c8;
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const float dc = (d[c] + g[c]) * msk[t];
        const float4 w0 = *reinterpret_cast<const float4*>(wt + c * Cin);
        const float4 w1 = *reinterpret_cast<const float4*>(wt + c * Cin + 4);
        acc[0] += dc * w0.x; acc[1] += dc * w0.y; acc[2] += dc * w0.z; acc[3] += dc * w0.w;
        acc[4] += dc * w1.x; acc[5] += dc * w1.y; acc[6] += dc * w1.z; acc[7] += dc * w1.w;
      }
    }
    float4* o = reinterpret_cast<float4*>(dx + p * Cin + c8);
    o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

// ---- input gradient on the matrix pipe --------------------------------------------------------------------------------------
// dx[pixel, ci] = sum over (tap, co) of dy[pixel - tap, co] * w[co, ci, tap]: per 16-pixel output row a GEMM with K = 9 taps x 8
// (padded) classes = 72 -> three 32-wide steps whose K group q IS one tap: the A fragment of lane (pixel, q) is the 16-byte dy
// record of the halo pixel that tap (4 ks + q) points at, the B fragment W[co 0..7][ci][tap].  36 MFMAs per row (3 steps x 4
// ci blocks x the three split passes) against 9 x (unpack + 16 FMA) per pixel and 8 channels on the vector units; the dy halo
// (180 pixels x 32 B) comes in by LDS-DMA, the fp32 result leaves through a per-wave LDS transposition as whole 256-byte
// pixel rows (the accumulator layout would store 64-byte pieces).
template <typename T>
__global__ __launch_bounds__(256, 2) void smallcout_dgrad_mfma_kernel(const T* __restrict__ dh, const T* __restrict__ dl,
                                                                      const float* __restrict__ w, float* __restrict__ dx,
                                                                      int B, int H, int W, int Cout) {
  constexpr int CIN = 64, COP = 8, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, NPIX = HY * HX;   // 180 halo pixels
  constexpr int NG = (NPIX + 31) / 32;                                                          // DMA groups of 32 pixels
  typedef typename T16<T>::v8 v8;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  // two distinct LDS objects, not one [2][..] array: the compiler puts a vmcnt(0) in front of every LDS read that may alias an
  // LDS-DMA in flight, and it tracks aliasing per object -- reads of tile A proceed while the DMA fills tile B
  __shared__ __attribute__((aligned(16))) T tileA[NG * 32 * 2 * COP];   // [halo pixel][hi 8 | lo 8]
  __shared__ __attribute__((aligned(16))) T tileB[NG * 32 * 2 * COP];
  __shared__ __attribute__((aligned(16))) float stage[4][TX * CIN];     // per wave: one output row, [pixel][ci]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fn = lane & 15, fq = lane >> 4;
  // B fragments: K group fq of step ks is tap 4 ks + fq (taps 9..11 do not exist: zero), its 8 K elements the classes
  v8 wh[3][4], wl[3][4];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 4 * ks + fq;
        const float v = (t < 9 && j < Cout) ? w[((int64_t)j * CIN + 16 * nb + fn) * 9 + t] : 0.f;
        wh[ks][nb][j] = (T)v;
        wl[ks][nb][j] = (T)lo_part<T>(v);
      }
  const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
  const int ntiles = B * tiles_y * tiles_x;
  const T* const zp = reinterpret_cast<const T*>(g_zero_page_sc);
  const int dp = lane >> 1, dhalf = lane & 1;   // DMA lane map: one instruction = 32 halo pixels x (hi | lo)
  auto stage_in = [&](int tl, T* tile) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int y0 = ty * TY - 1, x0 = tx * TX - 1;
    for (int g = wid; g < NG; g += 4) {
      const int hp = g * 32 + dp;
      const int hy = hp / HX, hx = hp - hy * HX;
      const int yy = y0 + hy, xx = x0 + hx;
      const bool in = hp < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const T* src = zp;
      if (in && (!dhalf || dl)) src = (dhalf ? dl : dh) + (((int64_t)b * H + yy) * W + xx) * COP;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(tile + g * 32 * 2 * COP), 16, 0, 0);
    }
  };
  auto compute = [&](int tl, const T* tile) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int py = wid * 2 + rr;
      f32x4 acc[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int t = 4 * ks + fq;                  // this lane's tap (per K group)
        const int kh = t / 3, kw = t - 3 * kh;
        // dx[y, x] takes dy[y - (kh - 1), x - (kw - 1)]: halo coordinates (py + 2 - kh, px + 2 - kw); taps >= 9 meet zero weights
        const int hp = t < 9 ? (py + 2 - kh) * HX + fn + 2 - kw : 0;
        const v8 ah = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(tile + hp * 2 * COP));
        const v8 al = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(tile + hp * 2 * COP + COP));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          acc[nb] = T16<T>::mfma16(ah, wh[ks][nb], acc[nb]);
          acc[nb] = T16<T>::mfma16(al, wh[ks][nb], acc[nb]);
          acc[nb] = T16<T>::mfma16(ah, wl[ks][nb], acc[nb]);
        }
      }
      // D[pixel 4 fq + j][ci 16 nb + fn] -> stage[pixel][ci] -> 16-byte stores, contiguous per instruction
      float* st = stage[wid];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) st[(4 * fq + j) * CIN + 16 * nb + fn] = acc[nb][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own writes, then its own reads: no barrier needed
      // one store instruction = 1 KB contiguous (4 pixel rows): lane = (pixel 4 i + lane / 16, 16-byte chunk lane % 16)
      const int oy = ty * TY + py;
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(st + (4 * i + (lane >> 4)) * CIN + (lane & 15) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ox = tx * TX + 4 * i + (lane >> 4);
        if (oy < H && ox < W)
          *reinterpret_cast<float4*>(dx + (((int64_t)b * H + oy) * W + ox) * CIN + (lane & 15) * 4) = v[i];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next row overwrites the stage
    }
  };
  // tile t in A while t + grid lands in B, and the other way round.  The counted wait leaves the result stores of the tile
  // just computed in flight only until the next barrier -- vmcnt counts stores too, so it is a plain vmcnt(0).
  int tl = xcd_remap(blockIdx.x, gridDim.x);
  if (tl < ntiles) stage_in(tl, tileA);
  while (tl < ntiles) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // tile A landed; every wave is done with tile B
    if (tl + (int)gridDim.x < ntiles) stage_in(tl + gridDim.x, tileB);
    compute(tl, tileA);
    tl += gridDim.x;
    if (tl >= ntiles) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tl + (int)gridDim.x < ntiles) stage_in(tl + gridDim.x, tileA);
    compute(tl, tileB);
    tl += gridDim.x;
  }
}

// dW[co][ci][kh][kw] = sum_p dy[p][co] * x[p + (kh-1, kw-1)][ci] for TWO output channels (co0, co0+1) per launch row
// (gridDim.y walks the channel pairs).  Same thread map as the forward: lane = (pixel, 8-channel chunk); the 9 taps'
// chunks are fetched unconditionally from coordinates clamped into the image (an outside tap only loses its dy),
// 9 x 8 x 2 partial sums live in registers over the lane's whole pixel range, and are folded once at the end: lanes of
// equal chunk by xor-shuffles, the four waves through LDS, one fp32 slab row per workgroup (asis_reduce_rows sums the
// rows in a fixed order: deterministic).  As an MFMA wgrad this layer pays a 32-row tile for 2 rows and is bound by
// staging the im2col operand through LDS (1.16 ms at 12 x 672^2 x 64); here x is read from L1/L2 nine times and
// nothing else moves.
template <typename T>
__global__ __launch_bounds__(256) void smallcout_wgrad_kernel(const T* __restrict__ dy, int CoP, const T* __restrict__ x,
                                                              float* __restrict__ slab, int B, int H, int W, int Cin,
                                                              int Cout) {
  __shared__ float red[4][8 * 144];  // [wave][chunk slot][tap*16 + e*2 + c]  (cpp <= 8 chunk slots per wave pass)
  const int cpp = Cin >> 3;          // lanes per pixel: 1, 2, 4 or 8
  const int co0 = blockIdx.y * 2;
  const bool two = co0 + 1 < Cout;
  const int64_t total = (int64_t)B * H * W * cpp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float acc[9][8][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e][0] = acc[t][e][1] = 0.f;
  const int c8 = (threadIdx.x & (cpp - 1)) * 8;  // the chunk is the same in every iteration (stride % cpp == 0)
  for (int64_t i = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t p = i / cpp;
    const int xq = (int)(p % W);
    const int yq = (int)((p / W) % H);
    const uint4 draw = *reinterpret_cast<const uint4*>(dy + p * CoP + (co0 & ~7));
    float d0, d1;
    {
      const uint32_t wsel = ((co0 & 7) >> 1) == 0 ? draw.x : (((co0 & 7) >> 1) == 1 ? draw.y : (((co0 & 7) >> 1) == 2 ? draw.z : draw.w));
      unpack2<T>(wsel, d0, d1);  // co0 is even: the pair sits in one 32-bit word
      if (!two) d1 = 0.f;
    }
    uint4 raw[9];
    float m[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int yy = yq + kh - 1, xx = xq + kw - 1;
        const bool inb = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const int64_t q = p + (int64_t)(yc - yq) * W + (xc - xq);
        raw[kh * 3 + kw] = *reinterpret_cast<const uint4*>(x + q * Cin + c8);
        m[kh * 3 + kw] = inb ? 1.f : 0.f;
      }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float a0 = d0 * m[t], a1 = d1 * m[t];
      float f0, f1;
      unpack2<T>(raw[t].x, f0, f1);
      acc[t][0][0] += a0 * f0; acc[t][0][1] += a1 * f0; acc[t][1][0] += a0 * f1; acc[t][1][1] += a1 * f1;
      unpack2<T>(raw[t].y, f0, f1);
      acc[t][2][0] += a0 * f0; acc[t][2][1] += a1 * f0; acc[t][3][0] += a0 * f1; acc[t][3][1] += a1 * f1;
      unpack2<T>(raw[t].z, f0, f1);
      acc[t][4][0] += a0 * f0; acc[t][4][1] += a1 * f0; acc[t][5][0] += a0 * f1; acc[t][5][1] += a1 * f1;
      unpack2<T>(raw[t].w, f0, f1);
      acc[t][6][0] += a0 * f0; acc[t][6][1] += a1 * f0; acc[t][7][0] += a0 * f1; acc[t][7][1] += a1 * f1;
    }
  }
  // fold lanes that hold the same chunk (lane bits above log2(cpp)), then the four waves
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float v = acc[t][e][c];
        for (int o = cpp; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        acc[t][e][c] = v;
      }
  if (lane < cpp) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[wid][lane * 144 + t * 16 + e * 2 + 0] = acc[t][e][0];
        red[wid][lane * 144 + t * 16 + e * 2 + 1] = acc[t][e][1];
      }
  }
  __syncthreads();
  // slab row layout = the parameter's [Cout][Cin][3][3]
  float* row = slab + (int64_t)blockIdx.x * Cout * Cin * 9;
  for (int i = threadIdx.x; i < cpp * 144; i += blockDim.x) {
    const int ch = i / 144, r = i - ch * 144;
    const int t = r >> 4, e = (r >> 1) & 7, c = r & 1;
    if (c == 1 && !two) continue;
    const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    row[((int64_t)(co0 + c) * Cin + ch * 8 + e) * 9 + t] = v;
  }
}

// ---- weight gradient on the matrix pipe ----------------------------------------------------------------------------------------
// dW[co][ci][tap] = sum over pixels of dy[p][co] * x[p + tap][ci]: the reduction runs over pixels, so both MFMA operands are
// needed K-major (8 consecutive pixels per lane) from tiles that are stored pixel-major -- the transposing LDS read
// ds_read_b64_tr_b16 does that (16 lanes fetch a 4 pixel x 16 column block, every lane ends up with 4 pixels of one column;
// each lane supplies its own row address, so the tap shift of the im2col view is just address arithmetic on the halo tile).
// Per 8 x 16 tile: four K steps of 32 pixels (two tile rows), per step one dy fragment and nine x fragments, 36 MFMAs
// (16x16x32) per wave; wave w owns input channels 16 w .. 16 w + 15 and keeps its nine 16 x 16 accumulators (tap by tap) over
// the workgroup's whole run of tiles.  One fp32 slab row per workgroup, summed by asis_reduce_rows in a fixed order.
template <typename T, bool UP = false>
__global__ __launch_bounds__(256, 2) void smallcout_wgrad_mfma_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                      float* __restrict__ slab, int B, int H, int W, int Cout,
                                                                      const UpSrc up = UpSrc{}) {
  constexpr int CIN = 64, COP = 8, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, NPIX = HY * HX;   // 180 halo pixels
  constexpr int NGX = (NPIX + 7) / 8;                                                           // x DMA groups of 8 pixels
  typedef typename T16<T>::v8 v8;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  __shared__ __attribute__((aligned(16))) T xt[NGX * 8 * CIN];     // [halo pixel][64 ci]
  __shared__ __attribute__((aligned(16))) float lowr[UP ? UP_RY * UP_RX * CIN : 4];   // UP: relu(bn(raw)) of the region under the halo
  __shared__ __attribute__((aligned(16))) T dt[TY * TX * COP];     // [tile pixel][8 co]
  __shared__ __attribute__((aligned(16))) T zt[8];                 // zeros: columns 8..15 of the dy fragment
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 8) zt[tid] = (T)0.f;
  const int fq = lane >> 4, li = lane & 15;
  const int rq = li >> 2, pc = li & 3;    // transposed read: this lane supplies row rq of its group's 4-row block, columns 4 pc ..
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
  const int ntiles = B * tiles_y * tiles_x;
  const T* const zp = reinterpret_cast<const T*>(g_zero_page_sc);
  for (int tl = xcd_remap(blockIdx.x, gridDim.x); tl < ntiles; tl += gridDim.x) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int y0 = ty * TY - 1, x0 = tx * TX - 1;
    if constexpr (UP) {                               // the x halo of the upsampled map, computed from the raw low-resolution map
      const float rh = H > 1 ? (float)(up.H - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(up.W - 1) / (float)(W - 1) : 0.f;
      const int r0 = (int)(rh * max(y0, 0)), c0 = (int)(rw * max(x0, 0));
      up_stage_lowres(up, b, r0, c0, lowr, tid);
      __syncthreads();
      for (int it = tid; it < NGX * 8 * 8; it += 256) {
        const int hp = it >> 3, c8 = it & 7;
        const int hy = hp / HX, hx = hp - hy * HX;
        const int yy = y0 + hy, xx = x0 + hx;
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (hp < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) up8_blend(up, lowr, r0, c0, yy, xx, c8, rh, rw, a8);
        uint4 oh;
        oh.x = pack2<T>(a8[0], a8[1]); oh.y = pack2<T>(a8[2], a8[3]); oh.z = pack2<T>(a8[4], a8[5]); oh.w = pack2<T>(a8[6], a8[7]);
        *reinterpret_cast<uint4*>(xt + hp * CIN + c8 * 8) = oh;
      }
    } else {
    for (int g = wid; g < NGX; g += 4) {             // x halo: one instruction = 8 pixels x 8 chunks
      const int hp = g * 8 + (lane >> 3);
      const int hy = hp / HX, hx = hp - hy * HX;
      const int yy = y0 + hy, xx = x0 + hx;
      const bool in = hp < NPIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const T* src = in ? x + (((int64_t)b * H + yy) * W + xx) * CIN + (lane & 7) * 8 : zp;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(xt + g * 8 * CIN), 16, 0, 0);
    }
    }
    if (wid < 2) {                                    // dy: one instruction = 64 tile pixels x 16 bytes
      const int pt = wid * 64 + lane;
      const int yy = ty * TY + (pt >> 4), xx = tx * TX + (pt & 15);
      const T* src = (yy < H && xx < W) ? dy + (((int64_t)b * H + yy) * W + xx) * COP : zp;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(dt + wid * 64 * COP), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                  // K step = tile rows 2 ks, 2 ks + 1
      auto tr2 = [&](const T* p0, const T* p1) -> v8 {
        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)p0);
        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)p1);
        return __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      // this lane's two source rows of the step: k = 8 fq + 4 half + rq  ->  tile pixel (2 ks + (k >> 4), k & 15)
      const int k0 = 8 * fq + rq, k1 = k0 + 4;
      const v8 bf = tr2(pc < 2 ? dt + (32 * ks + k0) * COP + 4 * pc : zt, pc < 2 ? dt + (32 * ks + k1) * COP + 4 * pc : zt);
      const int h0 = (2 * ks + (k0 >> 4)) * HX + (k0 & 15), h1 = (2 * ks + (k1 >> 4)) * HX + (k1 & 15);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int sh = kh * HX + kw;
          const v8 af = tr2(xt + (h0 + sh) * CIN + 16 * wid + 4 * pc, xt + (h1 + sh) * CIN + 16 * wid + 4 * pc);
          acc[kh * 3 + kw] = T16<T>::mfma16(af, bf, acc[kh * 3 + kw]);
        }
    }
    __syncthreads();   // the tiles are overwritten by the next DMA
  }
  // D[m = ci 16 wid + 4 fq + j][n = co li]  ->  slab row in the parameter's [Cout][Cin][3][3] layout
  float* row = slab + (int64_t)blockIdx.x * Cout * CIN * 9;
  if (li < Cout) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) row[((int64_t)li * CIN + 16 * wid + 4 * fq + j) * 9 + t] = acc[t][j];
  }
}

inline int grid_for(int64_t total, int cap = 256 * 32) {
  int64_t g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T>
int launch_fwd(hipStream_t s, const void* xh, const void* xl, const float* w, const float* bias, float* out, int B, int H,
               int W, int Cin, int Cout) {
  const int64_t total = (int64_t)B * H * W * (Cin / 8);
  const T* a = reinterpret_cast<const T*>(xh);
  const T* b = reinterpret_cast<const T*>(xl);
  // ASIS_SMALLCOUT_TILED (default 1): the LDS-tiled form for the 64 -> (1 | 2) conv on maps of at least one tile
  // ASIS_SMALLCOUT_TILED (default 1): LDS tile + MFMA for the 64 -> (<= 16) conv on maps of at least one tile, 0 = direct form
  static const int tiled = [] { const char* e = getenv("ASIS_SMALLCOUT_TILED"); return e ? atoi(e) : 1; }();
  if (tiled && Cin == 64 && Cout <= 16 && H >= 8 && W >= 16) {
    const int ntiles = B * ((H + 7) / 8) * ((W + 15) / 16);
    hipLaunchKernelGGL((smallcout_fwd_mfma_kernel<T>), dim3(ntiles < 512 ? ntiles : 512), dim3(256), 0, s, a, b, w, bias, out, B, H, W, Cout);
    return 0;
  }
#define FWD(CO)                                                                                                    \
  hipLaunchKernelGGL((smallcout_fwd_kernel<T, CO>), dim3(grid_for(total)), dim3(256), 9 * Cin * CO * sizeof(float), s, \
                     a, b, w, bias, out, B, H, W, Cin, Cout)
  if (Cout <= 2) FWD(2);
  else if (Cout <= 4) FWD(4);
  else if (Cout <= 8) FWD(8);
  else FWD(16);
#undef FWD
  return 0;
}

}  // namespace

extern "C" int asis_conv3x3_smallcout_fwd(void* stream, int dtype, const void* x_hi, const void* x_lo, const float* w,
                                          const float* bias, float* out, int B, int H, int W, int Cin, int Cout) {
  ASIS_REQUIRE(x_hi && w && out, "asis_conv3x3_smallcout_fwd: null pointer");
  ASIS_REQUIRE(Cout >= 1 && Cout <= MAXCO, "asis_conv3x3_smallcout_fwd: Cout=%d must be in 1..%d", Cout, MAXCO);
  ASIS_REQUIRE(Cin >= 8 && Cin <= 64 && (Cin & (Cin - 1)) == 0, "asis_conv3x3_smallcout_fwd: Cin=%d must be 8, 16, 32 or 64", Cin);
  ASIS_REQUIRE(asis_aligned16(x_hi) && (!x_lo || asis_aligned16(x_lo)), "asis_conv3x3_smallcout_fwd: alignment");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_conv3x3_smallcout_fwd: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16) launch_fwd<f16>(s, x_hi, x_lo, w, bias, out, B, H, W, Cin, Cout);
  else launch_fwd<bf16>(s, x_hi, x_lo, w, bias, out, B, H, W, Cin, Cout);
  ASIS_CHECK_LAUNCH("asis_conv3x3_smallcout_fwd");
  return ASIS_OK;
}

extern "C" int asis_conv3x3_smallcout_fwd_up(void* stream, int dtype, const float* raw, const float* scale, const float* shift,
                                             const float* w, const float* bias, float* out, int B, int H, int W, int Cin, int Cout) {
  ASIS_REQUIRE(raw && scale && shift && w && out, "asis_conv3x3_smallcout_fwd_up: null pointer");
  ASIS_REQUIRE(Cin == 64 && Cout >= 1 && Cout <= 16 && B > 0 && H >= 4 && W >= 8,
               "asis_conv3x3_smallcout_fwd_up: Cin=%d must be 64, Cout=%d in 1..16, the low-resolution map at least 4 x 8", Cin, Cout);
  ASIS_REQUIRE(asis_aligned16(raw) && asis_aligned16(scale) && asis_aligned16(shift), "asis_conv3x3_smallcout_fwd_up: alignment");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_conv3x3_smallcout_fwd_up: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int OH = 2 * H, OW = 2 * W;
  const int ntiles = B * ((OH + 7) / 8) * ((OW + 15) / 16);
  const UpSrc up{raw, scale, shift, H, W};
  const dim3 grid(ntiles < 512 ? ntiles : 512);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((smallcout_fwd_mfma_kernel<f16, true>), grid, dim3(256), 0, s, (const f16*)nullptr, (const f16*)nullptr, w, bias, out, B,
                       OH, OW, Cout, up);
  else
    hipLaunchKernelGGL((smallcout_fwd_mfma_kernel<bf16, true>), grid, dim3(256), 0, s, (const bf16*)nullptr, (const bf16*)nullptr, w, bias, out,
                       B, OH, OW, Cout, up);
  ASIS_CHECK_LAUNCH("asis_conv3x3_smallcout_fwd_up");
  return ASIS_OK;
}

extern "C" int asis_conv3x3_smallcout_wgrad_up(void* stream, int dtype, const void* dy, int CoP, const float* raw, const float* scale,
                                               const float* shift, float* slabs, int nblk, int B, int H, int W, int Cin, int Cout) {
  ASIS_REQUIRE(dy && raw && scale && shift && slabs, "asis_conv3x3_smallcout_wgrad_up: null pointer");
  ASIS_REQUIRE(Cin == 64 && CoP == 8 && Cout >= 1 && Cout <= 8 && B > 0 && H >= 4 && W >= 8,
               "asis_conv3x3_smallcout_wgrad_up: Cin=%d must be 64, CoP=%d 8, Cout=%d <= 8, the low-resolution map at least 4 x 8", Cin, CoP, Cout);
  ASIS_REQUIRE(nblk >= 1 && nblk <= 65535, "asis_conv3x3_smallcout_wgrad_up: bad slab count %d", nblk);
  ASIS_REQUIRE(asis_aligned16(dy) && asis_aligned16(raw) && asis_aligned16(scale) && asis_aligned16(shift), "asis_conv3x3_smallcout_wgrad_up: alignment");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_conv3x3_smallcout_wgrad_up: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const UpSrc up{raw, scale, shift, H, W};
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((smallcout_wgrad_mfma_kernel<f16, true>), dim3(nblk), dim3(256), 0, s, reinterpret_cast<const f16*>(dy), (const f16*)nullptr,
                       slabs, B, 2 * H, 2 * W, Cout, up);
  else
    hipLaunchKernelGGL((smallcout_wgrad_mfma_kernel<bf16, true>), dim3(nblk), dim3(256), 0, s, reinterpret_cast<const bf16*>(dy),
                       (const bf16*)nullptr, slabs, B, 2 * H, 2 * W, Cout, up);
  ASIS_CHECK_LAUNCH("asis_conv3x3_smallcout_wgrad_up");
  return ASIS_OK;
}

extern "C" int asis_conv3x3_smallcout_dgrad(void* stream, int dtype, const void* dy_hi, const void* dy_lo, int CoP,
                                            const float* w, float* dx, int B, int H, int W, int Cin, int Cout) {
  ASIS_REQUIRE(dy_hi && w && dx, "asis_conv3x3_smallcout_dgrad: null pointer");
  ASIS_REQUIRE(Cout >= 1 && Cout <= 8 && CoP >= 8 && CoP % 8 == 0, "asis_conv3x3_smallcout_dgrad: Cout=%d (<= 8), CoP=%d", Cout, CoP);
  ASIS_REQUIRE(Cin % 8 == 0 && Cin > 0 && 9 * 8 * Cin * 4 <= 64 * 1024, "asis_conv3x3_smallcout_dgrad: Cin=%d must be a multiple of 8, <= 224", Cin);
  ASIS_REQUIRE(asis_aligned16(dy_hi) && (!dy_lo || asis_aligned16(dy_lo)) && asis_aligned16(dx),
               "asis_conv3x3_smallcout_dgrad: alignment");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_conv3x3_smallcout_dgrad: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)B * H * W * (Cin / 8);
  // ASIS_SMALLCOUT_TILED (default 1): the MFMA form for Cin = 64 with the classes in one 8-channel record
  static const int tiled = [] { const char* e = getenv("ASIS_SMALLCOUT_TILED"); return e ? atoi(e) : 1; }();
  if (tiled && Cin == 64 && CoP == 8 && H >= 8 && W >= 16) {
    const int ntiles = B * ((H + 7) / 8) * ((W + 15) / 16);
    const dim3 grid(ntiles < 512 ? ntiles : 512);
    if (dtype == ASIS_F16)
      hipLaunchKernelGGL((smallcout_dgrad_mfma_kernel<f16>), grid, dim3(256), 0, s, reinterpret_cast<const f16*>(dy_hi),
                         reinterpret_cast<const f16*>(dy_lo), w, dx, B, H, W, Cout);
    else
      hipLaunchKernelGGL((smallcout_dgrad_mfma_kernel<bf16>), grid, dim3(256), 0, s, reinterpret_cast<const bf16*>(dy_hi),
                         reinterpret_cast<const bf16*>(dy_lo), w, dx, B, H, W, Cout);
    ASIS_CHECK_LAUNCH("asis_conv3x3_smallcout_dgrad");
    return ASIS_OK;
  }
  const int CO = Cout <= 2 ? 2 : (Cout <= 4 ? 4 : 8);
  const size_t shm = (size_t)9 * CO * Cin * sizeof(float);
#define DG(T, C)                                                                                                  \
  hipLaunchKernelGGL((smallcout_dgrad_kernel<T, C>), dim3(grid_for(total)), dim3(256), shm, s,                    \
                     reinterpret_cast<const T*>(dy_hi), reinterpret_cast<const T*>(dy_lo), CoP, w, dx, B, H, W, Cin, Cout)
  if (dtype == ASIS_F16) {
    if (CO == 2) DG(f16, 2); else if (CO == 4) DG(f16, 4); else DG(f16, 8);
  } else {
    if (CO == 2) DG(bf16, 2); else if (CO == 4) DG(bf16, 4); else DG(bf16, 8);
  }
#undef DG
  ASIS_CHECK_LAUNCH("asis_conv3x3_smallcout_dgrad");
  return ASIS_OK;
}

extern "C" int asis_conv3x3_smallcout_wgrad(void* stream, int dtype, const void* dy, int CoP, const void* x, float* slabs,
                                            int nblk, int B, int H, int W, int Cin, int Cout) {
  ASIS_REQUIRE(dy && x && slabs, "asis_conv3x3_smallcout_wgrad: null pointer");
  ASIS_REQUIRE(Cout >= 1 && Cout <= CoP && CoP % 8 == 0, "asis_conv3x3_smallcout_wgrad: Cout=%d CoP=%d", Cout, CoP);
  ASIS_REQUIRE(Cin >= 8 && Cin <= 64 && (Cin & (Cin - 1)) == 0, "asis_conv3x3_smallcout_wgrad: Cin=%d must be 8, 16, 32 or 64", Cin);
  ASIS_REQUIRE(nblk >= 1 && nblk <= 65535, "asis_conv3x3_smallcout_wgrad: bad slab count %d", nblk);
  ASIS_REQUIRE(asis_aligned16(dy) && asis_aligned16(x), "asis_conv3x3_smallcout_wgrad: alignment");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_conv3x3_smallcout_wgrad: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(nblk, (Cout + 1) / 2), block(256);
  // ASIS_SMALLCOUT_TILED (default 1): the MFMA form for Cin = 64 with the classes in one 8-channel record
  static const int tiled = [] { const char* e = getenv("ASIS_SMALLCOUT_TILED"); return e ? atoi(e) : 1; }();
  if (tiled && Cin == 64 && CoP == 8 && Cout <= 8 && H >= 8 && W >= 16) {
    if (dtype == ASIS_F16)
      hipLaunchKernelGGL((smallcout_wgrad_mfma_kernel<f16>), dim3(nblk), block, 0, s, reinterpret_cast<const f16*>(dy),
                         reinterpret_cast<const f16*>(x), slabs, B, H, W, Cout);
    else
      hipLaunchKernelGGL((smallcout_wgrad_mfma_kernel<bf16>), dim3(nblk), block, 0, s, reinterpret_cast<const bf16*>(dy),
                         reinterpret_cast<const bf16*>(x), slabs, B, H, W, Cout);
    ASIS_CHECK_LAUNCH("asis_conv3x3_smallcout_wgrad");
    return ASIS_OK;
  }
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((smallcout_wgrad_kernel<f16>), grid, block, 0, s, reinterpret_cast<const f16*>(dy), CoP,
                       reinterpret_cast<const f16*>(x), slabs, B, H, W, Cin, Cout);
  else
    hipLaunchKernelGGL((smallcout_wgrad_kernel<bf16>), grid, block, 0, s, reinterpret_cast<const bf16*>(dy), CoP,
                       reinterpret_cast<const bf16*>(x), slabs, B, H, W, Cin, Cout);
  ASIS_CHECK_LAUNCH("asis_conv3x3_smallcout_wgrad");
  return ASIS_OK;
}
