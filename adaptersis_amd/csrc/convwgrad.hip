// Weight gradient of the narrow 3x3 decoder convolutions (Cout 64 / 128, stride 1, pad 1) on halo tiles (round 5).
//
//   dW[co][ci][tap] = sum over pixels p of dy[p][co] * x[p + tap][ci]          (autograd of backbones/decoders.py:109-135)
//
// The implicit-GEMM form of wgrad.hip (128 x 128 x 64 register-staged tiles, split-K slabs) runs these layers at 0.12 / 0.19
// matrix-pipe busy: with 64 output channels its M tile is half padding and every tap re-reads x through the im2col index
// arithmetic.  Here, as in smallconv.hip's classifier kernel, the reduction index (pixels) is served from tiles staged ONCE:
//   * work item = an 8 x 16 pixel tile of one image; its dy tile [128 px][64 co] and the 10 x 18 halo of x [180 px][128 ci]
//     land in LDS by LDS-DMA (asm: attn_tiles.h), double-buffered, one barrier per tile;
//   * both MFMA operands are needed K-major (8 consecutive pixels per lane) from pixel-major tiles: ds_read_b64_tr_b16; the
//     nine taps are nine constant address offsets into the halo tile — no im2col arithmetic, x is fetched once per tile
//     instead of once per tap;
//   * v_mfma_f32_16x16x32, K step = two tile rows x 16 columns: k = 8 fq + j -> pixel (row 2 ks + (fq & 1), column 8 (fq >> 1)
//     + j), so that the two 16-lane groups of a half wave read two ROWS of the halo at the same columns: with the chunk swizzle
//     key(row, col) = (col & 3) | (row & 1) << 2 the eight pixels of a transposing read fall into eight different bank octets,
//     and the key is a per-lane constant per (tap column, tap row parity): every LDS address is one of 6 + 4 per-lane
//     registers plus an immediate;
//   * workgroup = 8 waves = 64 output channels x 128 input channels x 9 taps: wave w owns input channels 16 w .. 16 w + 15 and
//     keeps its 4 x 9 accumulator tiles (144 registers) over the workgroup's whole run of pixel tiles; wider layers split into
//     (co block, ci block) combinations over blockIdx.y.  One fp32 slab row per blockIdx.x in the parameter's
//     [Cout][Cin][3][3] layout, summed by asis_reduce_rows in a fixed order (deterministic).
#include "asis_common.h"
#include "attn_tiles.h"

namespace {

using attn_tiles::glds16;
using attn_tiles::lds_tr_ptr;
using attn_tiles::s16x8;

constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;   // pixel tile and its halo
constexpr int NHALO = HY * HX;                             // 180 halo pixels
constexpr int NSLOT = 184;                                 // rounded up to DMA groups of 4 pixels
constexpr int CIB = 128, COB = 64;                         // channels per workgroup
constexpr int XT = NSLOT * CIB, DT = TY * TX * COB;        // elements per buffer

__device__ __attribute__((aligned(16))) uint4 g_zero_page_cw[1];

template <typename T>
__global__ __launch_bounds__(512, 1) void conv_wgrad_halo_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                float* __restrict__ slab, int B, int H, int W, int Cin,
                                                                int Cout, int ld_dy, int ncib) {
  typedef typename T16<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T lds[2 * (XT + DT)];   // [buf][x halo 184 x 128 | dy 128 x 64] = 2 x 62.0 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fq = lane >> 4, li = lane & 15, rq = li >> 2, pc = li & 3;
  const int cib = blockIdx.y % ncib, cob = blockIdx.y / ncib;
  const int ci0 = cib * CIB, co0 = cob * COB;

  // ---- per-lane LDS read bases (elements) -------------------------------------------------------------------------------
  // x halo: slot (r, c) = 18 r + c, 256-byte rows, 32-byte piece p of the row at position p ^ key, key = (c & 3) | (r & 1) << 2.
  // This lane's pixel of a K step / tap: r = 2 ks + kh + (fq & 1), c = kw + 4 half + 8 (fq >> 1) + rq.
  int xb[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int key = ((rq + kw) & 3) | ((((fq & 1) + par) & 1) << 2);
      xb[kw][par] = (HX * (fq & 1) + 8 * (fq >> 1) + rq) * CIB + ((wid ^ key) << 4) + 4 * pc;
    }
  // dy tile: pixel pt = 16 row + col, 128-byte rows, 32-byte piece cb at position cb ^ kd, kd = ((pt >> 1) & 1) | ((pt >> 4) & 1) << 1
  int db[4];
  {
    const int kd = (rq >> 1) | ((fq & 1) << 1);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) db[cb] = (16 * (fq & 1) + 8 * (fq >> 1) + rq) * COB + ((cb ^ kd) << 4) + 4 * pc;
  }

  // ---- per-lane DMA assignments (tile-invariant) ----------------------------------------------------------------------------
  // x halo: 46 wave-instructions of 4 slots x 16 chunks; wave w issues groups w, w + 8, ...: lane = (slot in group, 16-byte chunk)
  const int xs_sl = lane >> 4, xs_ch = lane & 15;
  int x_off[6];        // source offset (elements) relative to the tile's (y0 - 1, x0 - 1) pixel, or -1: slot past the halo
  int x_r[6], x_c[6];  // halo row / column of the slot (for the image-border test)
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int slot = (wid + 8 * i) * 4 + xs_sl;
    const int r = slot / HX, c = slot - r * HX;
    const int key = (c & 3) | ((r & 1) << 2);
    const int piece = (xs_ch >> 1) ^ key;
    x_r[i] = r;
    x_c[i] = c;
    x_off[i] = (wid + 8 * i) * 4 < NSLOT && slot < NHALO ? ci0 + piece * 16 + (xs_ch & 1) * 8 : -1;
  }
  // dy: 16 wave-instructions of 8 pixels x 8 chunks; wave w issues groups w and w + 8
  const int ds_px = lane >> 3, ds_ch = lane & 7;
  int d_pt[2], d_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pt = (wid + 8 * i) * 8 + ds_px;
    const int kd = ((pt >> 1) & 1) | (((pt >> 4) & 1) << 1);
    d_pt[i] = pt;
    d_off[i] = co0 + (((ds_ch >> 1) ^ kd) << 4) + (ds_ch & 1) * 8;
  }

  const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
  const int ntiles = B * tiles_y * tiles_x;
  const char* const zp = reinterpret_cast<const char*>(g_zero_page_cw);
  auto issue = [&](int tl, int buf) {
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, b = tl / (tiles_x * tiles_y);
    const int y0 = ty * TY - 1, x0 = tx * TX - 1;
    T* xt = lds + buf * (XT + DT);
    T* dt = xt + XT;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if ((wid + 8 * i) * 4 < NSLOT) {   // wave-uniform
        const int yy = y0 + x_r[i], xx = x0 + x_c[i];
        const bool in = x_off[i] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const char* src = in ? reinterpret_cast<const char*>(x + (((int64_t)b * H + yy) * W + xx) * Cin + x_off[i]) : zp;
        glds16(src, xt + (wid + 8 * i) * 4 * CIB);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int yy = ty * TY + (d_pt[i] >> 4), xx = tx * TX + (d_pt[i] & 15);
      const char* src = (yy < H && xx < W) ? reinterpret_cast<const char*>(dy + (((int64_t)b * H + yy) * W + xx) * ld_dy + d_off[i]) : zp;
      glds16(src, dt + (wid + 8 * i) * 8 * COB);
    }
  };

  f32x4 acc[4][9];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tr2 = [&](const T* p0, const T* p1) -> v8 {
    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)p0);
    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)p1);
    return __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto compute = [&](int buf) {
    const T* xt = lds + buf * (XT + DT);
    const T* dt = xt + XT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {   // K step = tile rows 2 ks, 2 ks + 1
      v8 af[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) af[cb] = tr2(dt + 32 * ks * COB + db[cb], dt + (32 * ks + 4) * COB + db[cb]);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const T* p = xt + (HX * (2 * ks + kh) + kw) * CIB + xb[kw][kh & 1];
          const v8 bf = tr2(p, p + 4 * CIB);
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) acc[cb][kh * 3 + kw] = T16<T>::mfma16(af[cb], bf, acc[cb][kh * 3 + kw]);
        }
    }
  };

  // ---- the run of tiles of this workgroup: DMA of tile i + 1 under the MFMAs of tile i ------------------------------------
  int tl = blockIdx.x;
  int buf = 0;
  if (tl < ntiles) issue(tl, 0);
  for (; tl < ntiles; tl += gridDim.x, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // tile tl has landed for every wave; everyone is done with the other buffer
    if (tl + (int)gridDim.x < ntiles) issue(tl + gridDim.x, buf ^ 1);
    compute(buf);
  }

  // D[m = co 16 cb + 4 fq + j][n = ci 16 wid + li] -> slab row in the parameter's [Cout][Cin][3][3] layout
  float* row = slab + (int64_t)blockIdx.x * Cout * Cin * 9;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        row[((int64_t)(co0 + 16 * cb + 4 * fq + j) * Cin + ci0 + 16 * wid + li) * 9 + t] = acc[cb][t][j];
}

}  // namespace

extern "C" int asis_conv3x3_wgrad_halo_nblk(int B, int H, int W, int Cin, int Cout) {
  const int combos = (Cin / CIB) * (Cout / COB);
  const int ntiles = B * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
  int n = 256 / (combos > 0 ? combos : 1);
  if (n > ntiles) n = ntiles;
  return n < 1 ? 1 : n;
}

extern "C" int asis_conv3x3_wgrad_halo(void* stream, int dtype, const void* dy, int64_t ld_dy, const void* x, float* slabs, int nblk,
                                       int B, int H, int W, int Cin, int Cout) {
  ASIS_REQUIRE(dy && x && slabs, "asis_conv3x3_wgrad_halo: null pointer");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_conv3x3_wgrad_halo: bad dtype %d", dtype);
  ASIS_REQUIRE(B > 0 && H >= TY && W >= TX && Cin % CIB == 0 && Cin > 0 && Cout % COB == 0 && Cout > 0,
               "asis_conv3x3_wgrad_halo: needs Cin %% 128 == 0, Cout %% 64 == 0 and a map of at least 8 x 16 (B=%d H=%d W=%d Cin=%d Cout=%d)",
               B, H, W, Cin, Cout);
  ASIS_REQUIRE(ld_dy % 8 == 0 && ld_dy >= Cout, "asis_conv3x3_wgrad_halo: ld_dy=%ld must be a multiple of 8 and >= Cout", (long)ld_dy);
  ASIS_REQUIRE(asis_aligned16(dy) && asis_aligned16(x), "asis_conv3x3_wgrad_halo: operands must be 16-byte aligned");
  const int combos = (Cin / CIB) * (Cout / COB);
  ASIS_REQUIRE(nblk >= 1 && nblk <= 65535 && combos <= 65535, "asis_conv3x3_wgrad_halo: bad slab count %d", nblk);
  ASIS_REQUIRE((int64_t)B * H * W * (Cin > ld_dy ? Cin : ld_dy) < (1LL << 40), "asis_conv3x3_wgrad_halo: tensors too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(nblk, combos), block(512);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((conv_wgrad_halo_kernel<f16>), grid, block, 0, s, (const f16*)dy, (const f16*)x, slabs, B, H, W, Cin, Cout, (int)ld_dy, Cin / CIB);
  else
    hipLaunchKernelGGL((conv_wgrad_halo_kernel<bf16>), grid, block, 0, s, (const bf16*)dy, (const bf16*)x, slabs, B, H, W, Cin, Cout, (int)ld_dy, Cin / CIB);
  ASIS_CHECK_LAUNCH("asis_conv3x3_wgrad_halo");
  return ASIS_OK;
}
