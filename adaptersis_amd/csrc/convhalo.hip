// Halo-tile 3x3 convolution for NARROW outputs (Cout = 64 / 128) on split-precision operands with the MX correction pass.
//
//   out[b, y, x, n] = bias[n] + sum_{tap, c} (x_hi w_hi)[...] + MX pass (x_hi8 w_lo8 + x_lo8 w_hi8)      fp32 NHWC
//
// Replaces, for the last two FeatureDecoder stages (`backbones/decoders.py:109-135`: 256 -> 128 at 168^2, 128 -> 64 at 336^2 for a
// 588^2 input), the implicit-GEMM form of gemm_big.h.  There every K tile of a tap re-stages its 256..512 pixels x 64 channels
// from L2, i.e. each input element crosses L2 -> LDS nine times (once per tap) and twice more for the second plane; with only
// 64 / 128 output columns to amortise a staged row over, those launches are bound by LDS fill, not by the matrix pipe
// (6.2 GB of fills for 0.69 GB of input at 128 -> 64).  Here one workgroup owns a 16 x 16 pixel tile and ALL output
// channels; per 64-channel chunk and plane the 18 x 18 halo is staged ONCE (register-staged 16-byte loads, zero outside the image)
// and the nine taps read their A fragments from it at shifted pixel addresses; the weights of one kernel row (three taps,
// Cout x 64 channels each) are double-buffered per tap; four waves of four tile rows each, two workgroups per CU.  Same operand planes, same packed weights (asis_pack_conv_weight(_mx) mode 0: k = tap * Cin + c), same scales
// as the gemm_big.h MX instances; the summation order differs (chunk -> plane -> tap instead of chunk -> tap -> plane), so results
// agree to fp32 rounding, not bit for bit.
#include <type_traits>
#include "asis_common.h"

namespace {

typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef int v8i_ __attribute__((ext_vector_type(8)));

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv_halo_mx_kernel(const f16* __restrict__ x_hi, const f16* __restrict__ x_mx,
                                                              const f16* __restrict__ w_hi, const f16* __restrict__ w_mx,
                                                              const float* __restrict__ bias, const float* __restrict__ amax_a,
                                                              const float* __restrict__ amax_b, float* __restrict__ out,
                                                              float* __restrict__ stats, int B, int H, int W, int Cin, int TH, int TW, int CT) {
  // COUT channels of the layer's CT per workgroup (blockIdx.y picks the 64-channel group: the 128-channel form of this kernel
  // spills).  Four waves, each FOUR tile rows (four 16-pixel blocks) x all output channels: a B fragment read feeds 8 MFMAs; TWO workgroups per
  // CU (<= 74 KB of LDS, 256 registers at two waves per SIMD) so that one's barriers and staging hide under the other's MFMAs.
  constexpr int NB = COUT / 16;                 // 16-column blocks of the output channels
  constexpr int MB = 4;                         // tile rows per wave
  constexpr int HP = 18 * 18;                   // halo pixels
  constexpr int HALO = HP * 64;                 // f16 elements of one halo chunk (64 channels)
  constexpr int WT = COUT * 64;                 // f16 elements of one tap's weights
  constexpr int NT = 256;
  __shared__ __attribute__((aligned(16))) f16 lds[HALO + 3 * WT];   // one halo + a ring of three tap-weight buffers
  f16* const halo = lds;
  f16* const wbuf0 = lds + HALO;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q16 = lane >> 4;
  int bid = blockIdx.x;
  const int tx_ = bid % TW; bid /= TW;
  const int ty_ = bid % TH;
  const int b = bid / TH;
  const int y0 = ty_ * 16, x0 = tx_ * 16;
  const int64_t img = (int64_t)b * H * W;
  const int K9 = 9 * Cin;
  const int mx_sc = mx_code<f16>(*amax_a, *amax_b);
  const int mx_one = 127;
  const int n0 = blockIdx.y * COUT;              // first output channel of this workgroup
  w_hi += (int64_t)n0 * K9;
  w_mx += (int64_t)n0 * K9;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: groups g = chunk * 2 + plane (plane 0 = 16-bit hi, 1 = MX).  One halo buffer (the next group's halo waits in
  // registers while this group's nine taps run), the weights of one tap double-buffered ------------------------------------------
  const int ngroups = (Cin / 64) * 2;
  constexpr int HLOADS = (HP * 8 + NT - 1) / NT;   // 16-byte pieces per thread and halo (2592 pieces)
  uint4 hreg[HLOADS];
  auto halo_load = [&](int g) {   // global -> registers (unconditional loads from clamped addresses, zeroed after)
    const f16* src = (g & 1) ? x_mx : x_hi;
    const int c0 = (g >> 1) * 64;
#pragma unroll
    for (int it = 0; it < HLOADS; ++it) {
      int idx = it * NT + tid;
      idx = idx < HP * 8 ? idx : HP * 8 - 1;
      const int p = idx >> 3, ch = idx & 7;
      const int hy = p / 18, hx = p - hy * 18;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const int cy = ok ? gy : 0, cx = ok ? gx : 0;
      const uint4 v = *reinterpret_cast<const uint4*>(src + ((img + (int64_t)cy * W + cx) * Cin + c0 + ch * 8));
      hreg[it] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto halo_store = [&]() {   // registers -> LDS, chunk c of pixel p at chunk c ^ ((p >> 1) & 7): 128-byte rows, conflict-free b128 reads
#pragma unroll
    for (int it = 0; it < HLOADS; ++it) {
      const int idx = it * NT + tid;
      if (idx < HP * 8) {
        const int p = idx >> 3, ch = idx & 7;
        *reinterpret_cast<uint4*>(halo + p * 64 + ((ch ^ ((p >> 1) & 7)) << 3)) = hreg[it];
      }
    }
  };
  // weights: LDS-DMA, no staging registers — tap-step st = group * 9 + tap goes into ring buffer st % 3, issued TWO steps ahead of
  // its use (a global load outlasts one tap of MFMAs).  One wave-instruction = 8 rows (output channels) x 128 bytes; the lane that
  // fills slot s of row n fetches chunk s ^ ((n >> 1) & 7) (the image is swizzled on the source side, as in gemm_big.h).
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  static_assert(COUT == 64, "two DMA pieces per wave and tap: 4 waves x 2 x 8 rows");
  auto w_dma = [&](int st) {
    const int g = st / 9, tap = st - g * 9;
    const f16* src = (g & 1) ? w_mx : w_hi;
    const int c0 = (g >> 1) * 64;
    f16* dst = wbuf0 + (st % 3) * WT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = wid * 2 + i;
      const int n = q * 8 + (lane >> 3), ch = (lane & 7) ^ ((n >> 1) & 7);
      __builtin_amdgcn_global_load_lds((glb_ptr)(src + (int64_t)n * K9 + tap * Cin + c0 + ch * 8), (lds_ptr)(dst + q * 8 * 64), 16, 0, 0);
    }
  };
  auto mfma_mx = [](f32x4& a, v8i_ x, v8i_ y, int sx, int sy) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(a) : "v"(x), "v"(y), "v"(sx), "v"(sy));
  };

  const int nsteps = ngroups * 9;
  w_dma(0);
  w_dma(1);
  halo_load(0);
  halo_store();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int step = 0;
  for (int g = 0; g < ngroups; ++g) {
    const bool next_group = g + 1 < ngroups;
#pragma unroll 1
   for (int tap = 0; tap < 9; ++tap, ++step) {
    if (step + 2 < nsteps) w_dma(step + 2);      // into ring buffer (step + 2) % 3, last read in step - 1
    if (tap == 0 && next_group) halo_load(g + 1); // registers; lands under the nine taps of this group
    const f16* wt = wbuf0 + (step % 3) * WT;
    const int ty = tap / 3, tx = tap - ty * 3;
    // A fragments: the wave's four tile rows (16 pixels each), tap-shifted inside the halo; a lane's two 16-byte chunks (k groups
    // q16 and 4 + q16) as ONE 8-register vector: the scaled fp8 MFMA takes it whole, the 16-bit MFMAs take its halves
    v8i_ aw[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int p = (wid * MB + mb + ty) * 18 + r16 + tx;
      aw[mb] = __builtin_shufflevector(*reinterpret_cast<const v4i_*>(halo + p * 64 + (((q16) ^ ((p >> 1) & 7)) << 3)),
                                       *reinterpret_cast<const v4i_*>(halo + p * 64 + (((4 + q16) ^ ((p >> 1) & 7)) << 3)), 0, 1, 2, 3, 4, 5, 6, 7);
    }
    auto lo4 = [](v8i_ x) { return __builtin_bit_cast(f16x8, __builtin_shufflevector(x, x, 0, 1, 2, 3)); };
    auto hi4 = [](v8i_ x) { return __builtin_bit_cast(f16x8, __builtin_shufflevector(x, x, 4, 5, 6, 7)); };
    constexpr int NH = 2;                        // B fragments two blocks at a time (register budget)
#pragma unroll
    for (int h = 0; h < NB / NH; ++h) {
      v8i_ bw[NH];
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int n = (h * NH + k) * 16 + r16;
        bw[k] = __builtin_shufflevector(*reinterpret_cast<const v4i_*>(wt + n * 64 + (((q16) ^ ((n >> 1) & 7)) << 3)),
                                        *reinterpret_cast<const v4i_*>(wt + n * 64 + (((4 + q16) ^ ((n >> 1) & 7)) << 3)), 0, 1, 2, 3, 4, 5, 6, 7);
      }
      if (g & 1) {                               // MX plane: one block-scaled fp8 MFMA per 16 x 16 block (K = 128 bytes)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm MFMAs below are opaque to the compiler's wait insertion
#pragma unroll
        for (int k = 0; k < NH; ++k)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) mfma_mx(acc[mb][h * NH + k], bw[k], aw[mb], mx_sc, mx_one);
      } else {
#pragma unroll
        for (int k = 0; k < NH; ++k)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            acc[mb][h * NH + k] = T16<f16>::mfma16(lo4(bw[k]), lo4(aw[mb]), acc[mb][h * NH + k]);
            acc[mb][h * NH + k] = T16<f16>::mfma16(hi4(bw[k]), hi4(aw[mb]), acc[mb][h * NH + k]);
          }
      }
    }
    // the next step's weights (issued one step ago) must have landed: behind them this wave has issued this step's two pieces and,
    // in tap 0 of a group that prefetches a halo, the halo's 11 register loads (tap 1 waits for those too: whatever order the
    // compiler gave the two kinds of loads in tap 0, vmcnt(2) is then enough)
    if (step + 1 < nsteps) {
      if (step + 2 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (tap == 0 && next_group) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __syncthreads();
    if (tap == 8 && next_group) {                // every wave is done with this group's halo: overwrite it, then release the readers
      halo_store();
      __syncthreads();
    }
   }
  }

  // ---- epilogue: bias, fp32 NHWC stores (lane: pixel r16 of its tile row, channels nb * 16 + 4 q16 ..), BatchNorm partial sums ----
  float4 s_sum[NB], s_sq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    s_sum[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_sq[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int gy = y0 + wid * MB + mb, gx = x0 + r16;
    const bool ok = gy < H && gx < W;
    float* op = out + (img + (int64_t)(ok ? gy : 0) * W + (ok ? gx : 0)) * CT + n0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = nb * 16 + 4 * q16;
      const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n0 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 v = make_float4(acc[mb][nb][0] + bv.x, acc[mb][nb][1] + bv.y, acc[mb][nb][2] + bv.z, acc[mb][nb][3] + bv.w);
      if (ok) {
        *reinterpret_cast<float4*>(op + n) = v;
        s_sum[nb].x += v.x; s_sum[nb].y += v.y; s_sum[nb].z += v.z; s_sum[nb].w += v.w;
        s_sq[nb].x += v.x * v.x; s_sq[nb].y += v.y * v.y; s_sq[nb].z += v.z * v.z; s_sq[nb].w += v.w * v.w;
      }
    }
  }
  if (stats) {
    // fold the 16 pixels of a lane group (r16) with xor-shuffles, then the 4 waves through LDS (the staging buffers are dead)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s_sum[nb].x += __shfl_xor(s_sum[nb].x, o, 64); s_sum[nb].y += __shfl_xor(s_sum[nb].y, o, 64);
        s_sum[nb].z += __shfl_xor(s_sum[nb].z, o, 64); s_sum[nb].w += __shfl_xor(s_sum[nb].w, o, 64);
        s_sq[nb].x += __shfl_xor(s_sq[nb].x, o, 64); s_sq[nb].y += __shfl_xor(s_sq[nb].y, o, 64);
        s_sq[nb].z += __shfl_xor(s_sq[nb].z, o, 64); s_sq[nb].w += __shfl_xor(s_sq[nb].w, o, 64);
      }
    float* red = reinterpret_cast<float*>(lds);   // [4 waves][2][COUT]
    if (r16 == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        *reinterpret_cast<float4*>(red + (wid * 2 + 0) * COUT + nb * 16 + 4 * q16) = s_sum[nb];
        *reinterpret_cast<float4*>(red + (wid * 2 + 1) * COUT + nb * 16 + 4 * q16) = s_sq[nb];
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const int which = tid / COUT, n = tid - which * COUT;
      float s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s += red[(w4 * 2 + which) * COUT + n];
      stats[((int64_t)blockIdx.x * 2 + which) * CT + n0 + n] = s;
    }
  }
}

}  // namespace

extern "C" int asis_conv3x3_halo_mx_tiles(int B, int H, int W) { return B * ((H + 15) / 16) * ((W + 15) / 16); }

extern "C" int asis_conv3x3_halo_mx(void* stream, int dtype, const void* x_hi, const void* x_mx, const void* w_hi, const void* w_mx,
                                    const float* bias, const float* amax_a, const float* amax_b, float* out, float* stats, int B,
                                    int H, int W, int Cin, int Cout) {
  ASIS_REQUIRE(x_hi && x_mx && w_hi && w_mx && amax_a && amax_b && out, "asis_conv3x3_halo_mx: null pointer");
  ASIS_REQUIRE(dtype == ASIS_F16, "asis_conv3x3_halo_mx: float16 operands only (the MX correction pass is built for them)");
  ASIS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 64 == 0 && (Cout == 64 || Cout == 128),
               "asis_conv3x3_halo_mx: Cin=%d must be a multiple of 64, Cout=%d one of 64 / 128", Cin, Cout);
  ASIS_REQUIRE(asis_aligned16(x_hi) && asis_aligned16(x_mx) && asis_aligned16(w_hi) && asis_aligned16(w_mx) && asis_aligned16(out) &&
                   (!bias || asis_aligned16(bias)), "asis_conv3x3_halo_mx: pointers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int TH = (H + 15) / 16, TW = (W + 15) / 16;
  dim3 grid((unsigned)(B * TH * TW), (unsigned)(Cout / 64)), block(256);
  const f16* xh = reinterpret_cast<const f16*>(x_hi);
  const f16* xm = reinterpret_cast<const f16*>(x_mx);
  const f16* wh = reinterpret_cast<const f16*>(w_hi);
  const f16* wm = reinterpret_cast<const f16*>(w_mx);
  hipLaunchKernelGGL((conv_halo_mx_kernel<64>), grid, block, 0, s, xh, xm, wh, wm, bias, amax_a, amax_b, out, stats, B, H, W, Cin, TH, TW, Cout);
  ASIS_CHECK_LAUNCH("asis_conv3x3_halo_mx");
  return ASIS_OK;
}
