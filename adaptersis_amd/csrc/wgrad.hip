// Weight-gradient GEMM (both operands K-strided: "TN"), conv-aware, split-K, deterministic.
//
//   dW[co][ci][kh][kw] = sum_p dy[p][co] * x[b, oh*s + kh - pad, ow*s + kw - pad, ci]
//                        p = (b, oh, ow) over all output pixels          (conv2d weight grad)
//   with KH = KW = 1 it is the nn.Linear weight grad dW[n_out][n_in] = dy^T x.
//
// As a GEMM: C[M = Cout][N = KH*KW*Cin] = A^T B with A = dy [P, Cout] and B = implicit
// im2col(x) [P, KH*KW*Cin]; the reduction index (pixels) is the slow axis of both operands, so
// tiles are staged in LDS as [k][m] / [k][n] (coalesced 16-byte rows) and the MFMA fragments
// (8 consecutive k per lane) are assembled with ds_read_b64_tr_b16, the gfx950 transposing LDS
// read (cdna_hip_programming.md T10).  16-byte chunks of each 256-B LDS row are XOR-swizzled
// with (k & 3) << 2 so the four k-rows of one transposed read land in different banks.
// The pixel range is split over gridDim.y; every split writes its own fp32 slab (already in the
// parameter's [Cout, Cin, KH, KW] layout) and asis_reduce_rows sums the slabs in a fixed order.
#include "asis_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;

// DENSE: 1x1 / stride 1 / no padding (nn.Linear and 1x1 conv weight gradients): x row p is pixel p, no index arithmetic
template <typename T, bool DENSE>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const asis_wgrad_desc d) {
  typedef typename T16<T>::v8 v8;
  typedef s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
  __shared__ __attribute__((aligned(16))) T lds[2 * 2 * BK * BM];  // [buf][A|B][64 k][128] = 64 KiB

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int Ntot = d.KH * d.KW * d.Cin;
  const int tiles_n = (Ntot + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int64_t k_begin = (int64_t)blockIdx.y * d.k_per_split;
  int64_t k_end = k_begin + d.k_per_split;
  if (k_end > d.P) k_end = d.P;

  const T* __restrict__ A = reinterpret_cast<const T*>(d.dy);
  const T* __restrict__ X = reinterpret_cast<const T*>(d.x);

  // loader: thread -> k rows (tid>>4) + 16*i, 16-byte chunk (tid & 15)
  const int lk = tid >> 4, lch = tid & 15;
  const int am = m0 + lch * 8;
  const bool a_col_ok = am < d.CoP;
  const int bn = n0 + lch * 8;
  const bool b_col_ok = bn < Ntot;
  const int tap = b_col_ok ? bn / d.Cin : 0;
  const int ci = bn - tap * d.Cin;
  const int kh = tap / d.KW, kw = tap - kh * d.KW;
  const int ohw = d.OH * d.OW;
  const bool small_p = d.P < (1 << 24);
  const float inv_ohw = 1.0f / (float)ohw, inv_ow = 1.0f / (float)d.OW;

  uint4 ra[4], rb[4];
  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t p = k0 + lk + 16 * i;
      uint4 va = make_uint4(0, 0, 0, 0), vb = va;
      if (p < k_end) {
        if (a_col_ok) va = *reinterpret_cast<const uint4*>(A + p * d.ld_dy + am);
        if (DENSE) {
          if (b_col_ok) vb = *reinterpret_cast<const uint4*>(X + p * d.Cin + bn);
        } else if (b_col_ok) {
          int b, rem, oh, ow;
          if (small_p) {  // P < 2^24: quotients from an fp32 reciprocal estimate + one correction step (exact), ~8 VALU
            b = (int)((float)(int)p * inv_ohw);   // instead of two ~30-instruction integer divisions per staged row
            rem = (int)p - b * ohw;
            if (rem < 0) { --b; rem += ohw; } else if (rem >= ohw) { ++b; rem -= ohw; }
            oh = (int)((float)rem * inv_ow);
            ow = rem - oh * d.OW;
            if (ow < 0) { --oh; ow += d.OW; } else if (ow >= d.OW) { ++oh; ow -= d.OW; }
          } else {
            b = (int)(p / ohw);
            rem = (int)(p - (int64_t)b * ohw);
            oh = rem / d.OW;
            ow = rem - oh * d.OW;
          }
          const int ih = oh * d.stride + kh - d.pad, iw = ow * d.stride + kw - d.pad;
          if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
            vb = *reinterpret_cast<const uint4*>(X + (((int64_t)b * d.H + ih) * d.W + iw) * d.Cin + ci);
        }
      }
      ra[i] = va;
      rb[i] = vb;
    }
  };
  auto store_tile = [&](int buf) {
    T* As = lds + buf * (2 * BK * BM);
    T* Bs = As + BK * BM;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = lk + 16 * i;
      const int sw = (lch ^ ((k & 3) << 2)) << 3;
      *reinterpret_cast<uint4*>(As + k * BM + sw) = ra[i];
      *reinterpret_cast<uint4*>(Bs + k * BN + sw) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = (int)((k_end - k_begin + BK - 1) / BK);
  if (nt > 0) {
    load_tile(k_begin);
    store_tile(0);
  }
  __syncthreads();

  // transposed-read geometry: 16-lane group g reads a 4(k) x 16(m) block
  const int g = lane >> 4, li = lane & 15;
  const int fh = g >> 1;              // k half of the MFMA fragment
  const int msub = 16 * (g & 1);      // column sub-block inside the 32-wide fragment
  const int q = li >> 2, pc = li & 3; // this lane supplies row q, columns 4*pc..4*pc+3
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load_tile(k_begin + (int64_t)(t + 1) * BK);
    const T* As = lds + buf * (2 * BK * BM);
    const T* Bs = As + BK * BM;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ca = wr * 64 + i * 32 + msub + 4 * pc;  // first column supplied by this lane
        const int cb = wc * 64 + i * 32 + msub + 4 * pc;
        s16x4 lo[2], hi[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int k = ks * 16 + 8 * fh + 4 * half + q;
          const int swz = (k & 3) << 2;
          const T* pa = As + k * BM + ((((ca >> 3) ^ swz) << 3) | (ca & 7));
          const T* pb = Bs + k * BN + ((((cb >> 3) ^ swz) << 3) | (cb & 7));
          const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(pa));
          const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(pb));
          if (half == 0) { lo[0] = va; lo[1] = vb; } else { hi[0] = va; hi[1] = vb; }
        }
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 a8 = __builtin_shufflevector(lo[0], hi[0], 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8 b8 = __builtin_shufflevector(lo[1], hi[1], 0, 1, 2, 3, 4, 5, 6, 7);
        af[i] = __builtin_bit_cast(v8, a8);
        bf[i] = __builtin_bit_cast(v8, b8);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = T16<T>::mfma32(af[i], bf[j], acc[i][j]);
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: slab[split][(co*Cin + ci)*taps + tap]
  float* slab = d.out + (int64_t)blockIdx.y * d.Cout * Ntot;
  const int taps = d.KH * d.KW;
  const int fr = lane & 31, fq = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + fr;
    if (n >= Ntot) continue;
    const int tp = n / d.Cin, cc = n - tp * d.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
        if (m < d.Cout) slab[((int64_t)m * d.Cin + cc) * taps + tp] = acc[i][j][r];
      }
  }
}

}  // namespace

extern "C" int asis_wgrad_splits(int64_t P, int Cout, int Ntot) {
  const int64_t tiles = asis_cdiv(Cout, BM) * asis_cdiv(Ntot, BN);
  int64_t want = asis_cdiv(1024, tiles);          // ~4 workgroups per CU overall
  const int64_t max_by_k = asis_cdiv(P, 4 * BK);  // at least 4 K tiles per split
  if (want > max_by_k) want = max_by_k;
  if (want > 512) want = 512;
  if (want < 1) want = 1;
  return (int)want;
}

extern "C" int asis_wgrad(void* stream, const asis_wgrad_desc* dp) {
  ASIS_REQUIRE(dp != nullptr, "asis_wgrad: null descriptor");
  asis_wgrad_desc d = *dp;
  ASIS_REQUIRE(d.dy && d.x && d.out, "asis_wgrad: null pointer");
  ASIS_REQUIRE(d.dtype == ASIS_F16 || d.dtype == ASIS_BF16, "asis_wgrad: bad dtype %d", d.dtype);
  ASIS_REQUIRE(d.Cin % 8 == 0 && d.Cin > 0 && d.Cout > 0, "asis_wgrad: Cin=%d must be a multiple of 8", d.Cin);
  ASIS_REQUIRE(d.CoP % 8 == 0 && d.CoP >= d.Cout && d.ld_dy >= d.CoP && d.ld_dy % 8 == 0,
               "asis_wgrad: dy channels must be padded to a multiple of 8 (CoP=%d, ld=%ld)", d.CoP, (long)d.ld_dy);
  ASIS_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.pad >= 0, "asis_wgrad: bad conv geometry");
  ASIS_REQUIRE(d.OH == (d.H + 2 * d.pad - d.KH) / d.stride + 1 && d.OW == (d.W + 2 * d.pad - d.KW) / d.stride + 1,
               "asis_wgrad: output size mismatch");
  ASIS_REQUIRE(d.P == (int64_t)d.B_ * d.OH * d.OW, "asis_wgrad: P != B*OH*OW");
  ASIS_REQUIRE(asis_aligned16(d.dy) && asis_aligned16(d.x), "asis_wgrad: operands must be 16-byte aligned");
  ASIS_REQUIRE(d.splits >= 1 && d.splits <= 65535, "asis_wgrad: bad splits %d", d.splits);
  const int Ntot = d.KH * d.KW * d.Cin;
  d.k_per_split = asis_cdiv(asis_cdiv(d.P, d.splits), BK) * BK;
  ASIS_REQUIRE((int64_t)d.k_per_split * d.splits >= d.P, "asis_wgrad: internal split error");
  dim3 grid((unsigned)(asis_cdiv(d.Cout, BM) * asis_cdiv(Ntot, BN)), d.splits), block(NTHREADS);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool dense = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0;
  if (d.dtype == ASIS_F16) {
    if (dense) hipLaunchKernelGGL((wgrad_kernel<f16, true>), grid, block, 0, s, d);
    else hipLaunchKernelGGL((wgrad_kernel<f16, false>), grid, block, 0, s, d);
  } else {
    if (dense) hipLaunchKernelGGL((wgrad_kernel<bf16, true>), grid, block, 0, s, d);
    else hipLaunchKernelGGL((wgrad_kernel<bf16, false>), grid, block, 0, s, d);
  }
  ASIS_CHECK_LAUNCH("asis_wgrad");
  return ASIS_OK;
}
