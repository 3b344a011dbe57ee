// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M,N] = epilogue( A[M,K] * B[N,K]^T )      16-bit operands, fp32 accumulate
//
// Tiling (wave64, v_mfma_f32_32x32x16_{f16,bf16}):
//   workgroup 256 threads = 4 waves (2x2), tile 128x128x64, each wave a 64x64 sub-tile
//   = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  A and B tiles are staged
//   global -> VGPR -> LDS (register staging lets the A loader be an arbitrary per-lane
//   gather with zero fill: that is what turns the same kernel into a 3x3 implicit-GEMM
//   convolution on NHWC activations).  Two LDS buffers (64 KiB) -> 2 workgroups / CU;
//   the global loads of tile t+1 are issued before the MFMAs of tile t and written to the
//   other buffer after them (one barrier per K tile).
//   LDS image: [row][64 k] (128-B rows); the 16-B chunk index is XOR-swizzled with
//   (row>>1)&7, which makes the ds_read_b128 fragment reads (16-lane groups
//   {0-3,12-15,20-27}...) conflict-free (MI355X_MICROARCH.md §LDS).
//   Block ids are remapped XCD-aware so the N tiles of one M tile share an L2.
//
// Reference sites replaced: see include/asis_hip.h (asis_gemm).
#include <stdlib.h>

#include "asis_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;

template <typename T, bool CONV>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const asis_gemm_desc d) {
  typedef typename T16<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T lds[2 * (BM + BN) * BK];  // 64 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;

  const int tiles_n = (d.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n;
  const int tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int bz = blockIdx.y;

  const T* __restrict__ A = reinterpret_cast<const T*>(d.A) + (int64_t)bz * d.strideA;
  const T* __restrict__ B = reinterpret_cast<const T*>(d.B) + (int64_t)bz * d.strideB;

  // ---- loader geometry: thread -> (row lrow + 32*i, 16-byte chunk lchk) -----------------
  const int lrow = tid >> 3;
  const int lchk = tid & 7;
  int64_t a_off[4];  // dense: element offset of row start; conv: pixel index base of image b
  int a_ih0[4], a_iw0[4];
  bool a_ok[4], b_ok[4];
  int64_t b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lrow + 32 * i;
    a_ok[i] = m < d.M;
    if (CONV) {
      const int mm = a_ok[i] ? m : 0;
      const int ohw = d.OH * d.OW;
      const int b = mm / ohw;
      const int rem = mm - b * ohw;
      const int oh = rem / d.OW;
      const int ow = rem - oh * d.OW;
      a_ih0[i] = oh * d.stride - d.pad;
      a_iw0[i] = ow * d.stride - d.pad;
      a_off[i] = (int64_t)b * d.H * d.W;
    } else {
      a_off[i] = (int64_t)m * d.lda;
      a_ih0[i] = a_iw0[i] = 0;
    }
    const int n = n0 + lrow + 32 * i;
    b_ok[i] = n < d.N;
    b_off[i] = (int64_t)n * d.ldb;
  }

  uint4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
    const int kk = k0 + lchk * 8;
    const bool kok = kk < d.K;
    int kh = 0, kw = 0, ci = kk;
    if (CONV) {
      const int tap = kk / d.Cin;
      ci = kk - tap * d.Cin;
      kh = tap / d.KW;
      kw = tap - kh * d.KW;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (CONV) {
        const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
        if (a_ok[i] && kok && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
          v = *reinterpret_cast<const uint4*>(A + (a_off[i] + (int64_t)ih * d.W + iw) * d.Cin + ci);
      } else {
        if (a_ok[i] && kok) v = *reinterpret_cast<const uint4*>(A + a_off[i] + kk);
      }
      ra[i] = v;
      uint4 w = make_uint4(0, 0, 0, 0);
      if (b_ok[i] && kok) w = *reinterpret_cast<const uint4*>(B + b_off[i] + kk);
      rb[i] = w;
    }
  };
  auto store_tile = [&](int buf) {
    T* As = lds + buf * ((BM + BN) * BK);
    T* Bs = As + BM * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 32 * i;
      const int sw = (lchk ^ ((row >> 1) & 7)) << 3;
      *reinterpret_cast<uint4*>(As + row * BK + sw) = ra[i];
      *reinterpret_cast<uint4*>(Bs + row * BK + sw) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = (d.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int fr = lane & 31, fh = lane >> 5;
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load_tile((t + 1) * BK);
    const T* As = lds + buf * ((BM + BN) * BK);
    const T* Bs = As + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v8 af[2], bf[2];
      const int chunk = 2 * ks + fh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wr * 64 + i * 32 + fr;
        af[i] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(As + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3)));
        const int col = wc * 64 + i * 32 + fr;
        bf[i] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Bs + col * BK + ((chunk ^ ((col >> 1) & 7)) << 3)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = T16<T>::mfma32(af[i], bf[j], acc[i][j]);
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------
  const int64_t cbase = (int64_t)bz * d.strideC;
  const float* __restrict__ res = d.res ? d.res + (int64_t)bz * d.strideR : nullptr;
  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wc * 64 + j * 32 + fr;
    const bool cok = col < d.N;
    const float bn = (d.bias_n && cok) ? d.bias_n[col] : 0.f;
    const float sc = (d.scale_n && cok) ? d.scale_n[col] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < d.M && cok) {
          float v = acc[i][j][r] + bn;
          if (d.bias_m) v += d.bias_m[row];
          if (d.act == ASIS_ACT_GELU) v = gelu_erf(v);
          else if (d.act == ASIS_ACT_RELU) v = fmaxf(v, 0.f);
          v *= sc;
          if (res) v += res[(int64_t)row * d.ldr + col];
          if (d.out_f32) reinterpret_cast<float*>(d.C)[cbase + (int64_t)row * d.ldc + col] = v;
          else reinterpret_cast<T*>(d.C)[cbase + (int64_t)row * d.ldc + col] = to_t16<T>(v);
          csum[j] += v;
          csq[j] += v * v;
        }
      }
    }
  }
  if (d.stats) {
    // deterministic per-(tile_m, column) partial sums: lanes l / l+32 hold the same column,
    // waves wr=0/1 the two row halves; combine through LDS (free after the last barrier).
    float* red = reinterpret_cast<float*>(lds);  // [2 wr][2 kind][128 col]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = csum[j] + __shfl_xor(csum[j], 32, 64);
      float q = csq[j] + __shfl_xor(csq[j], 32, 64);
      if (fh == 0) {
        red[(wr * 2 + 0) * BN + wc * 64 + j * 32 + fr] = s;
        red[(wr * 2 + 1) * BN + wc * 64 + j * 32 + fr] = q;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int col = n0 + tid;
      if (col < d.N) {
        d.stats[((int64_t)tile_m * 2 + 0) * d.N + col] = red[0 * BN + tid] + red[2 * BN + tid];
        d.stats[((int64_t)tile_m * 2 + 1) * d.N + col] = red[1 * BN + tid] + red[3 * BN + tid];
      }
    }
  }
}

#include "gemm_big.h"
#include "gemm_p8.h"

// ASIS_GEMM_P8 / asis_gemm_set_option("p8", v): -1 = not read yet
static int g_gemm_p8 = -1;
static bool ph8_m16_on() { static const int v = [] { const char* e = getenv("ASIS_GEMM_8P_M16"); return e ? atoi(e) : 1; }(); return v != 0; }

template <typename T>
int launch(hipStream_t s, const asis_gemm_desc& d) {
  // large-tile LDS-DMA kernel (gemm_big.h); ASIS_GEMM_BIG=0 forces the 128x128 register-staged kernel
  static const int group_m = [] { const char* e = getenv("ASIS_GEMM_GROUPM"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
  static const int big_mode = [] { const char* e = getenv("ASIS_GEMM_BIG"); return e ? atoi(e) : 1; }();
  const bool vec_ok = (d.N % 4 == 0) && (d.ldc % 4 == 0);
  const bool split = d.A_lo != nullptr || d.B_lo != nullptr;
  if (d.act == ASIS_ACT_GELU_GRAD) {  // only the vector epilogue of the large-tile kernels implements it
    const bool ok = big_mode && !d.conv && !split && !d.stats && d.aux && d.K % 32 == 0 && d.M >= 256 && d.N >= 128 && vec_ok &&
                    d.ld_aux % 4 == 0 && d.batch == 1 && (reinterpret_cast<uintptr_t>(d.aux) & 7) == 0 &&
                    (reinterpret_cast<uintptr_t>(d.C) & 15) == 0 && !d.bias_m;
    if (!ok) return ASIS_EINVAL;
  }
  if (d.act == ASIS_ACT_SILU_MUL) {
    // SwiGLU epilogue: lives in the 16-bit fast path of the 8-phase 16x16-MFMA instances (gemm_big.h) — the launch goes there
    // directly or is refused (a fall-through to another epilogue would write x1 | x2 as if they were outputs)
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool mxs = d.mx_amax_a && d.mx_amax_b && d.A_lo && d.B_lo;
    const bool ok = big_mode && ph8_m16_on() && !d.conv && (mxs || !split) && !d.out_f32 && !d.res && !d.res16 && !d.C_lo && !d.rowstats &&
                    !d.stats && !d.scale_n && !d.bias_m && !d.ln_mr && !d.aux && d.batch == 1 && d.ksplit <= 1 && d.K % 64 == 0 && d.M >= 256 &&
                    d.N >= 256 && d.N % 32 == 0 && d.ldc % 8 == 0 && d.ldc >= d.N / 2 && al16(d.C) && (!d.bias_n || al16(d.bias_n)) &&
                    (int64_t)d.N * d.ldb < (1ll << 31);
    if (!ok) return ASIS_EINVAL;
    dim3 g8(((d.M + 255) / 256) * ((d.N + 255) / 256), 1), block(512);
    if (mxs) hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, false, true, 64, 1, true, true, 0, true>), g8, block, 0, s, d, group_m);
    else hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, false, false, 64, 1, true, true>), g8, block, 0, s, d, group_m);
    return 0;
  }
  if (d.ksplit > 1 && !(big_mode && d.conv)) return ASIS_EINVAL;
  // ASIS_GEMM_P8 (default 1): dense launches with at least one 256x256 tile per CU run on the PERSISTENT form of the 8-phase
  // kernel (gemm_p8.h: one workgroup per CU walks its tiles, the next tile's first K tile is staged during the last K tile of
  // the current one, the epilogue's stores drain under the next tile) when K <= 2048 (at K = 4096 a tile is 64 K tiles long
  // and the per-tile savings no longer pay for the static tile lists: 347 vs 341 us); split-precision halves (A_lo / B_lo)
  // are further K parts of the same stream; 2 = any K, from 16 tiles on (tests); 3 = any K; 0 = never
  {
    if (g_gemm_p8 < 0) { const char* e = getenv("ASIS_GEMM_P8"); g_gemm_p8 = e ? atoi(e) : 1; }
    const int p8 = g_gemm_p8;
    const int64_t p8_tiles = (int64_t)((d.M + 255) / 256) * ((d.N + 255) / 256);
    const int kparts = 1 + (d.A_lo ? 1 : 0) + (d.B_lo ? 1 : 0);
    auto al = [](const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    if (p8 && big_mode && ph8_m16_on() && !d.conv && !d.stats && d.batch == 1 && !d.bias_m && !d.ln_cols && !d.C_lo && !d.rowstats && !d.res16 &&
        !d.mx_amax_a && !d.mx_amax_b &&
        d.ksplit <= 1 && d.K % 64 == 0 &&
        d.M >= 256 && d.N >= 256 && p8_tiles >= (p8 == 2 ? 16 : 256) && d.K >= 128 && (p8 >= 2 || d.K * kparts <= 2048) &&
        (int64_t)d.M * d.lda * 2 < (1ll << 32) && (int64_t)d.N * d.ldb * 2 < (1ll << 32) && d.N % 8 == 0 && d.ldc % 8 == 0 &&
        al(d.C, 16) && (!d.res || (al(d.res, 16) && d.ldr % 4 == 0)) && (!d.bias_n || al(d.bias_n, 16)) && (!d.scale_n || al(d.scale_n, 16)) &&
        (d.act != ASIS_ACT_GELU_GRAD || (d.aux && al(d.aux, 8) && d.ld_aux % 4 == 0))) {
      const int nwg = p8_tiles >= 256 ? 256 : (int)(p8_tiles / 8) * 8;
      if (d.ln_mr) hipLaunchKernelGGL((gemm_p8_kernel<T, 1>), dim3(nwg), dim3(512), 0, s, d, group_m);   // LayerNorm-fold consumer
      else hipLaunchKernelGGL((gemm_p8_kernel<T, 0>), dim3(nwg), dim3(512), 0, s, d, group_m);
      return 0;
    }
  }
  // LayerNorm-fold fields (C_lo / rowstats / res16 / ln_mr): implemented by the persistent kernel above and by the dense 8-phase
  // one-tile-per-workgroup form below; anything else is refused (the caller runs asis_layernorm and plain launches)
  const bool lnx = d.C_lo || d.rowstats || d.res16 || d.ln_mr;
  if (lnx) {
    static const int ph8x = [] { const char* e = getenv("ASIS_GEMM_8P"); return e ? atoi(e) : 1; }();
    const bool ph8_ok = ph8x && ph8_m16_on() && big_mode && !split && !d.conv && !d.stats && d.K % 64 == 0 && d.M >= 256 && d.N >= 256 &&
                        (ph8x == 2 || d.K >= 2048 || (ph8x == 1 && d.K >= 1024 && (int64_t)((d.M + 255) / 256) * ((d.N + 255) / 256) * d.batch >= 128)) &&
                        d.N % 8 == 0 && d.ldc % 8 == 0 && (!d.res16 || (d.res16_lo && d.ldr16 % 8 == 0 && d.ldr16 >= d.N && asis_aligned16(d.res16) && asis_aligned16(d.res16_lo))) &&
                        (!d.C_lo || !d.out_f32) && (!d.ln_mr || d.ln_cs) &&
                        // producer launches (dedicated 8-columns-per-lane epilogue of gemm_big.h): plain v = res + scale * (acc + bias)
                        (!(d.C_lo || d.rowstats || d.res16) ||
                         ((!d.res || (d.ldr % 8 == 0 && asis_aligned16(d.res))) && (!d.C_lo || asis_aligned16(d.C_lo)) && asis_aligned16(d.C) &&
                          (!d.bias_n || asis_aligned16(d.bias_n)) && (!d.scale_n || asis_aligned16(d.scale_n)) && d.batch == 1 && !d.bias_m &&
                          d.act == ASIS_ACT_NONE && !d.ln_mr));
    if (!ph8_ok) return ASIS_EINVAL;
  }
  if (split) {  // one pass over the virtual 3K reduction; only on the large-tile kernel
    const bool ok = big_mode && d.K % BK == 0 && d.M >= 256 && d.N >= 32 && vec_ok && (d.out_f32 || !d.conv) &&
                    (!d.conv || (d.Cin % BK == 0 && d.A_lo && d.B_lo));
    if (!ok) return ASIS_EINVAL;
    const int bm = 256, bn = d.N > 64 ? 128 : 64;
    if (d.ksplit > 1 && (d.batch != 1 || d.stats || (d.KH * d.KW) % d.ksplit != 0 || !d.out_f32 || d.res)) return ASIS_EINVAL;
    dim3 grid(((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn), d.ksplit > 1 ? d.ksplit : d.batch), block(512);
    if (d.mx_amax_a || d.mx_amax_b) {
      // MX form of the lo operands (include/asis_hip.h): two K parts, the second on the block-scaled fp8 MFMA; 16x16 MFMA
      // instances of the three convolution tile forms
      if (!(d.mx_amax_a && d.mx_amax_b && d.A_lo && d.B_lo)) return ASIS_EINVAL;
      if (!d.conv) {   // dense: the 8-phase one-tile-per-workgroup form, K tiles alternating 16-bit / MX
        if (!(d.K % 64 == 0 && d.N >= 256 && d.batch == 1 && d.ksplit <= 1 && !d.stats && (int64_t)d.N * d.ldb < (1ll << 31))) return ASIS_EINVAL;
        dim3 g8(((d.M + 255) / 256) * ((d.N + 255) / 256), 1);
        hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, false, true, 64, 1, true, true, 0, true>), g8, block, 0, s, d, group_m);
        return 0;
      }
      if ((int64_t)d.B_ * d.H * d.W * d.Cin >= (1ll << 31) || (int64_t)d.N * d.ldb >= (1ll << 31) || d.H >= 65536 || d.W >= 65536) return ASIS_EINVAL;
      static const int mx_ph8 = [] { const char* e = getenv("ASIS_MX_PH8"); return e ? atoi(e) : 1; }();   // lab: 0 = generic forms only
      if (mx_ph8 && d.N >= 256 && (d.N % 256 == 0 || d.N >= 1024) && d.batch == 1) {
        dim3 g8(((d.M + 255) / 256) * ((d.N + 255) / 256), d.ksplit > 1 ? d.ksplit : 1);
        hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, true, true, 64, 1, true, true, 0, true>), g8, block, 0, s, d, group_m);
      } else if (bn == 64 && d.ksplit <= 1 && d.M >= 512) {
        dim3 g512(((d.M + 511) / 512) * ((d.N + 63) / 64), 1);
        hipLaunchKernelGGL((gemm_big_kernel<T, 8, 1, 2, 2, 2, 0, true, true, 64, 1, false, true, 0, true>), g512, block, 0, s, d, group_m);
      } else if (bn == 128) {
        hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, true, true, 64, 1, false, true, 0, true>), grid, block, 0, s, d, group_m);
      } else {
        hipLaunchKernelGGL((gemm_big_kernel<T, 8, 1, 1, 2, 3, 0, true, true, 64, 1, false, true, 0, true>), grid, block, 0, s, d, group_m);
      }
      return 0;
    }
    static const int conv32 = [] { const char* e = getenv("ASIS_CONV_BK32"); return e ? atoi(e) : 0; }();
    // <= 64 output channels: 512x64 tile, 8 waves of 64x64 (1 KB of LDS fragment reads per MFMA; the 256x64 form's 32x64
    // wave tiles need 1.5 KB and are LDS-bound), one workgroup per CU: +0.6 % on the step (ASIS_CONV_T512=0: old form)
    static const int t512 = [] { const char* e = getenv("ASIS_CONV_T512"); return e ? atoi(e) : 1; }();
    static const int cm16 = [] { const char* e = getenv("ASIS_CONV_M16"); return e ? atoi(e) : 0; }();  // 16x16x32 MFMAs in the split convs
    // ASIS_CONV_8P (default 1): convolutions with whole 256-column tiles on the 8-phase 256x256x64 main loop (gemm_big.h):
    // the long reductions of the decoder convs are where that loop is at its best (K = 9 * Cin * 3 parts)
    static const int conv8p = [] { const char* e = getenv("ASIS_CONV_8P"); return e ? atoi(e) : 1; }();
    if (d.conv && conv8p && d.N >= 256 && (d.N % 256 == 0 || d.N >= 1024) && d.batch == 1) {
      dim3 g8(((d.M + 255) / 256) * ((d.N + 255) / 256), d.ksplit > 1 ? d.ksplit : 1);
      hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, true, true, 64, 1, true, true>), g8, block, 0, s, d, group_m);
      return 0;
    }
    if (d.conv && bn == 64 && t512 && d.ksplit <= 1 && d.M >= 512) {
      dim3 g512(((d.M + 511) / 512) * ((d.N + 63) / 64), 1);
      if (cm16) hipLaunchKernelGGL((gemm_big_kernel<T, 8, 1, 2, 2, 2, 0, true, true, 64, 1, false, true>), g512, block, 0, s, d, group_m);
      else hipLaunchKernelGGL((gemm_big_kernel<T, 8, 1, 2, 2, 2, 0, true, true, 64, 1>), g512, block, 0, s, d, group_m);
      return 0;
    }
    if (d.conv) {
      if (conv32 && bn == 128) hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, true, true, 32, 4>), grid, block, 0, s, d, group_m);
      else if (bn == 128 && cm16) hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, true, true, 64, 2, false, true>), grid, block, 0, s, d, group_m);
      else if (bn == 128) hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, true, true>), grid, block, 0, s, d, group_m);
      else hipLaunchKernelGGL((gemm_big_kernel<T, 8, 1, 1, 2, 3, 0, true, true>), grid, block, 0, s, d, group_m);
    } else {
      if (bn == 128) hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, false, true>), grid, block, 0, s, d, group_m);
      else hipLaunchKernelGGL((gemm_big_kernel<T, 8, 1, 1, 2, 3, 0, false, true>), grid, block, 0, s, d, group_m);
    }
    return 0;
  }
  static const int m16 = [] { const char* e = getenv("ASIS_GEMM_M16"); return e ? atoi(e) : 1; }();  // 16x16x32 MFMAs in the default dense form
  // ASIS_GEMM_8P: 0 = never; 1 (default) = K >= 2048 (fc2 / its input gradient: 343 vs 476 us at 42348x1024x4096) and,
  // since the phases run on 16x16x32 MFMAs (ASIS_GEMM_8P_M16: fc2 391 -> 350 us in isolation), the unbatched K >= 1024
  // GEMMs too (in the step: qk 222 -> 214 us, proj 160 -> 139, fc1 457 -> 441, adapter projections 242 -> 216; +3.5 % on the
  // step) — the batched ragged V^T GEMM lost there at first (66 vs 72 us: coarser tile quantisation) and wins since the form's
  // 16-bit outputs are converted before the LDS transposition (+0.9 % on the step);
  // (only when there are at least 128 of its 256x256 tiles: a launch that cannot fill the chip is better off with twice as
  // many 256x128 tiles); 3 = K >= 2048 only (the round-1 rule); 2 = wherever the shape allows
  static const int ph8 = [] { const char* e = getenv("ASIS_GEMM_8P"); return e ? atoi(e) : 1; }();
  static const int ph8_mink = [] { const char* e = getenv("ASIS_GEMM_8P_MINK"); return e ? atoi(e) : 1024; }();  // lab: see above
  if (ph8 && (ph8 == 2 || d.K >= 2048 || (ph8 == 1 && d.K >= ph8_mink && (int64_t)((d.M + 255) / 256) * ((d.N + 255) / 256) * d.batch >= 128)) && big_mode && !d.conv && !d.stats && d.K % 64 == 0 && d.M >= 256 && d.N >= 256) {
    // 256x256x64 tile, 8-phase main loop (one workgroup per CU: 128 KB of LDS)
    dim3 grid(((d.M + 255) / 256) * ((d.N + 255) / 256), d.batch), block(512);
    // ASIS_GEMM_8P_M16 (default 1): the phases issue 16 v_mfma_f32_16x16x32 instead of 8 32x32x16 (same FLOP, higher clock)
    static const int ph8_m16 = [] { const char* e = getenv("ASIS_GEMM_8P_M16"); return e ? atoi(e) : 1; }();
    // ASIS_GEMM_8P_SLAB32=1 (lab): 16-bit outputs through the fp32 slab epilogue like the others
    static const int ph8_slab32 = [] { const char* e = getenv("ASIS_GEMM_8P_SLAB32"); return e ? atoi(e) : 0; }();
    const int group_m_f = group_m | (ph8_slab32 ? 0x10000 : 0);
    if (ph8_m16) {
      if (lnx) hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, false, false, 64, 1, true, true, 1>), grid, block, 0, s, d, group_m);
      else hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, false, false, 64, 1, true, true>), grid, block, 0, s, d, group_m_f);
      return 0;
    }
    hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, false, false, 64, 1, true>), grid, block, 0, s, d, group_m);
    return 0;
  }
  if (big_mode && !d.conv && !d.stats && d.K % 32 == 0 && d.M >= 256 && d.N >= 128) {
    // 256x128x32 tile, 3 LDS stages (72 KB) and <= 128 VGPRs: TWO workgroups per CU, so one workgroup's epilogue
    // (HBM-write bound) overlaps the other's K loop: +12-15 % over the 1-workgroup 256x256x64 / 256x128x64 forms
    // on the fc1 shape (scripts/gemm_lab.hip).  big_mode 2/3 keep the older forms selectable for A/B runs.
    const int bm = 256, bn = (big_mode == 2 && d.N >= 2048) ? 256 : 128;
    dim3 grid(((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn), d.batch), block(512);
    if (big_mode == 2 && bn == 256) hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2>), grid, block, 0, s, d, group_m);
    else if (big_mode >= 2) hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3>), grid, block, 0, s, d, group_m);
    else if (m16) hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, false, false, 32, 4, false, true>), grid, block, 0, s, d, group_m);
    else hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, false, false, 32, 4>), grid, block, 0, s, d, group_m);
    return 0;
  }
  // implicit-GEMM convolution on the same kernel: K tiles must lie inside one tap, and BatchNorm statistics
  // come from the vectorised epilogue
  if (big_mode && d.conv && d.Cin % BK == 0 && d.M >= 256 && d.N >= 32 && vec_ok && d.out_f32) {
    const int bm = 256, bn = d.N > 64 ? 128 : 64;
    if (d.ksplit > 1 && (d.stats || (d.KH * d.KW) % d.ksplit != 0 || d.res)) return ASIS_EINVAL;
    dim3 grid(((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn), d.ksplit > 1 ? d.ksplit : 1), block(512);
    static const int conv8p = [] { const char* e = getenv("ASIS_CONV_8P"); return e ? atoi(e) : 1; }();
    if (conv8p && d.N >= 256 && (d.N % 256 == 0 || d.N >= 1024) && d.batch == 1) {
      dim3 g8(((d.M + 255) / 256) * ((d.N + 255) / 256), d.ksplit > 1 ? d.ksplit : 1);
      hipLaunchKernelGGL((gemm_big_kernel<T, 2, 4, 4, 2, 2, 0, true, false, 64, 1, true, true>), g8, block, 0, s, d, group_m);
      return 0;
    }
    if (bn == 128) hipLaunchKernelGGL((gemm_big_kernel<T, 4, 2, 2, 2, 3, 0, true>), grid, block, 0, s, d, group_m);
    else hipLaunchKernelGGL((gemm_big_kernel<T, 8, 1, 1, 2, 3, 0, true>), grid, block, 0, s, d, group_m);
    return 0;
  }
  if (d.ksplit > 1) return ASIS_EINVAL;  // K parts exist on the large-tile conv forms only
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n, d.batch), block(NTHREADS);
  if (d.conv)
    hipLaunchKernelGGL((gemm_kernel<T, true>), grid, block, 0, s, d);
  else
    hipLaunchKernelGGL((gemm_kernel<T, false>), grid, block, 0, s, d);
  return 0;
}

}  // namespace

extern "C" int asis_gemm_tiles_m(int M) { return (M + BM - 1) / BM; }

extern "C" int asis_gemm_set_option(const char* name, int value) {
  ASIS_REQUIRE(name != nullptr, "asis_gemm_set_option: null name");
  if (strcmp(name, "p8") == 0) { g_gemm_p8 = value; return ASIS_OK; }
  ASIS_FAIL(ASIS_EINVAL, "asis_gemm_set_option: unknown option '%s'", name);
}


extern "C" int asis_gemm(void* stream, const asis_gemm_desc* dp) {
  ASIS_REQUIRE(dp != nullptr, "asis_gemm: null descriptor");
  asis_gemm_desc d = *dp;
  ASIS_REQUIRE(d.A && d.B && d.C, "asis_gemm: null operand pointer");
  ASIS_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "asis_gemm: M,N,K must be positive (got %d,%d,%d)", d.M, d.N, d.K);
  ASIS_REQUIRE(d.dtype == ASIS_F16 || d.dtype == ASIS_BF16, "asis_gemm: bad dtype %d", d.dtype);
  ASIS_REQUIRE(d.K % 8 == 0, "asis_gemm: K=%d must be a multiple of 8", d.K);
  ASIS_REQUIRE(d.ldb % 8 == 0 && d.ldb >= d.K, "asis_gemm: ldb=%ld must be a multiple of 8 and >= K", (long)d.ldb);
  ASIS_REQUIRE(asis_aligned16(d.A) && asis_aligned16(d.B), "asis_gemm: A and B must be 16-byte aligned");
  ASIS_REQUIRE(d.strideA % 8 == 0 && d.strideB % 8 == 0, "asis_gemm: batch strides must be multiples of 8");
  ASIS_REQUIRE(d.ldc >= (d.act == ASIS_ACT_SILU_MUL ? d.N / 2 : d.N), "asis_gemm: ldc=%ld < N=%d", (long)d.ldc, d.N);
  ASIS_REQUIRE(d.act >= 0 && d.act <= ASIS_ACT_GELU_GRAD, "asis_gemm: bad act %d", d.act);
  if (d.batch <= 0) d.batch = 1;
  ASIS_REQUIRE(d.batch <= 65535, "asis_gemm: batch %d too large", d.batch);
  if (d.res) ASIS_REQUIRE(d.ldr >= d.N, "asis_gemm: ldr=%ld < N", (long)d.ldr);
  if (d.conv) {
    ASIS_REQUIRE(d.batch == 1, "asis_gemm: conv mode needs batch == 1 (batch is folded into M)");
    ASIS_REQUIRE(d.Cin % 8 == 0 && d.Cin > 0, "asis_gemm: conv Cin=%d must be a multiple of 8", d.Cin);
    ASIS_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.pad >= 0, "asis_gemm: bad conv geometry");
    ASIS_REQUIRE(d.K == d.KH * d.KW * d.Cin, "asis_gemm: conv K=%d != KH*KW*Cin=%d", d.K, d.KH * d.KW * d.Cin);
    ASIS_REQUIRE(d.OH == (d.H + 2 * d.pad - d.KH) / d.stride + 1 && d.OW == (d.W + 2 * d.pad - d.KW) / d.stride + 1,
                 "asis_gemm: conv output size mismatch");
    ASIS_REQUIRE((int64_t)d.B_ * d.OH * d.OW == d.M, "asis_gemm: conv M=%d != B*OH*OW", d.M);
  } else {
    ASIS_REQUIRE(d.lda % 8 == 0 && d.lda >= d.K, "asis_gemm: lda=%ld must be a multiple of 8 and >= K", (long)d.lda);
  }
  if (d.stats) ASIS_REQUIRE(d.batch == 1, "asis_gemm: stats need batch == 1");
  if (d.res16) ASIS_REQUIRE(d.res16_lo && !d.res && d.ldr16 % 4 == 0 && d.ldr16 >= d.N && (reinterpret_cast<uintptr_t>(d.res16) & 7) == 0 &&
                            (reinterpret_cast<uintptr_t>(d.res16_lo) & 7) == 0, "asis_gemm: res16 needs res16_lo, no fp32 res, ldr16 %% 4 == 0, 8-byte aligned planes");
  if (d.C_lo) ASIS_REQUIRE(!d.out_f32 && (reinterpret_cast<uintptr_t>(d.C_lo) & 7) == 0, "asis_gemm: C_lo is the second plane of a 16-bit output");
  if (d.ln_mr) ASIS_REQUIRE(d.ln_cs && (reinterpret_cast<uintptr_t>(d.ln_mr) & 7) == 0 && asis_aligned16(d.ln_cs), "asis_gemm: ln_mr needs ln_cs (16-byte aligned)");
  if (d.rowstats) ASIS_REQUIRE(d.batch == 1 && (reinterpret_cast<uintptr_t>(d.rowstats) & 7) == 0, "asis_gemm: rowstats need batch == 1");
  // the row statistics are per 64-column group ([M, N / 64, 2]): a ragged last group would need its own count in the finalizer
  if (d.rowstats) ASIS_REQUIRE(d.N % 64 == 0, "asis_gemm: rowstats need N %% 64 == 0 (N = %d)", (int)d.N);
  const int64_t tiles = (int64_t)asis_cdiv(d.M, BM) * asis_cdiv(d.N, BN);
  ASIS_REQUIRE(tiles < (1ll << 31), "asis_gemm: too many tiles");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ASIS_REQUIRE(!d.conv || (d.A_lo == nullptr) == (d.B_lo == nullptr), "asis_gemm: a split convolution needs both A_lo and B_lo");
  ASIS_REQUIRE((!d.A_lo || asis_aligned16(d.A_lo)) && (!d.B_lo || asis_aligned16(d.B_lo)), "asis_gemm: split halves must be 16-byte aligned");
  const int rc = (d.dtype == ASIS_F16) ? launch<f16>(s, d) : launch<bf16>(s, d);
  if (rc != 0 && d.act == ASIS_ACT_SILU_MUL)
    ASIS_FAIL(ASIS_EINVAL, "asis_gemm: ASIS_ACT_SILU_MUL needs a dense 16-bit-output launch on the 8-phase form (include/asis_hip.h: K %% 64 == 0, "
                           "M >= 256, N >= 256, N %% 32 == 0, ldc %% 8 == 0, plain or MX split operands, bias only)");
  if (rc != 0) ASIS_FAIL(ASIS_EINVAL, "asis_gemm: split-precision operands / ASIS_ACT_GELU_GRAD / ksplit need the large-tile path (K %% 64 "
                                        "== 0 (GELU_GRAD: 32), M >= 256, N >= 32 (128), N and ldc multiples of 4, fp32 output for "
                                        "split; conv: Cin %% 64 == 0)");
  ASIS_CHECK_LAUNCH("asis_gemm");
  return ASIS_OK;
}
