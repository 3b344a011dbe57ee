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
// BME: rows (output channels) of the tile that are computed: 128, 64 or 32.  The LDS image keeps 128 columns; a
// narrower tile only changes which fragments the four waves own (2x2 of 64x64 | 1x4 of 64x32 | 1x4 of 32x32), so a
// conv with 64 (or 2) output channels does not pay MFMAs for 128.
template <typename T, bool DENSE, int BME = 128>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const asis_wgrad_desc d) {
  constexpr int WR = BME == 128 ? 2 : 1, WC = 4 / WR;
  constexpr int FI = BME / (32 * WR), FJ = BN / (32 * WC);
  typedef typename T16<T>::v8 v8;
  typedef s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
  __shared__ __attribute__((aligned(16))) T lds[2 * 2 * BK * BM];  // [buf][A|B][64 k][128] = 64 KiB

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = WR == 2 ? wid >> 1 : 0, wc = WR == 2 ? (wid & 1) : wid;
  const int m_off = wr * (BME / WR), n_off = wc * (BN / WC);
  const int Ntot = d.KH * d.KW * d.Cin;
  const int tiles_n = (Ntot + BN - 1) / BN;
  // XCD-aware order: all tiles of one pixel split read the same x and dy pixels (each column tile a different tap /
  // channel slice of them); dealt round-robin they sit on 8 different L2s and every one fetches the operands from HBM
  // (measured for the 64-channel decoder conv: 4.8 GB of L2 fills per launch for 0.5 GB of tensors, at 6.3 TB/s the
  // launch was memory-bound).  xcd_remap keeps a split's tiles on one XCD.
  const int lin = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bx = lin % gridDim.x, by = lin / gridDim.x;
  const int tile_m = bx / tiles_n, tile_n = bx - tile_m * tiles_n;
  const int m0 = tile_m * BME, n0 = tile_n * BN;
  const int64_t k_begin = (int64_t)by * d.k_per_split;
  int64_t k_end = k_begin + d.k_per_split;
  if (k_end > d.P) k_end = d.P;

  const T* __restrict__ A = reinterpret_cast<const T*>(d.dy);
  const T* __restrict__ X = reinterpret_cast<const T*>(d.x);

  // loader: thread -> k rows (tid>>4) + 16*i, 16-byte chunk (tid & 15)
  const int lk = tid >> 4, lch = tid & 15;
  const int am = m0 + lch * 8;
  const bool a_col_ok = am < d.CoP && lch * 8 < BME;
  const int bn = n0 + lch * 8;
  const bool b_col_ok = bn < Ntot;
  const int tap = b_col_ok ? bn / d.Cin : 0;
  const int ci = bn - tap * d.Cin;
  const int kh = tap / d.KW, kw = tap - kh * d.KW;
  const int ohw = d.OH * d.OW;

  // Loader state, advanced incrementally: a K step moves every staged row BK pixels on, i.e. (sb images, soh rows,
  // sow columns) with at most one carry per digit, so no division happens inside the loop (the divisions used to be
  // ~300 VALU instructions per staged row against 16 MFMAs per tile).  Loads are unconditional from addresses clamped
  // into the tensors; what must read as zero (rows past the split, padding taps, columns past the tile) is masked
  // when the registers go to LDS, after the wait that is needed there anyway (a predicated `v = load` would put that
  // wait right behind the issue).
  const int sb = BK / ohw, srem = BK - sb * ohw;
  const int soh = srem / d.OW, sow = srem - soh * d.OW;
  const int64_t img = (int64_t)d.H * d.W * d.Cin;
  struct Row {  // one staged pixel row; four named instances (arrays of these ended up in scratch memory)
    const T* pa;  // dy row (+ column chunk)
    const T* px;  // DENSE: x row (+ column chunk); conv: base of the row's image (+ ci)
    int64_t p;
    int oh, ow;
  };
  Row w0, w1, w2, w3;
  uint32_t okm = 0;  // bit i: row i of the tile in flight is inside the split; bit 4+i: its tap is inside the image
  auto init_row = [&](Row& r, int i) {
    r.p = k_begin + lk + 16 * i;
    const int64_t pc = r.p < k_end ? r.p : (k_end > k_begin ? k_end - 1 : k_begin);
    if (r.p < k_end) okm |= 1u << i;
    r.pa = A + pc * d.ld_dy + (a_col_ok ? am : 0);
    if (DENSE) {
      r.px = X + pc * d.Cin + (b_col_ok ? bn : 0);
      r.oh = r.ow = 0;
    } else {
      const int64_t bimg = pc / ohw;
      const int rem = (int)(pc - bimg * ohw);
      r.oh = rem / d.OW;
      r.ow = rem - r.oh * d.OW;
      r.px = X + bimg * img + (b_col_ok ? ci : 0);
    }
  };
  init_row(w0, 0);
  init_row(w1, 1);
  init_row(w2, 2);
  init_row(w3, 3);
  const int64_t a_step = (int64_t)BK * d.ld_dy, x_step = (int64_t)BK * d.Cin;
  auto adv_row = [&](Row& r, int i) {
    r.p += BK;
    const bool in = r.p < k_end;  // rows past the split keep their last valid addresses and are masked
    okm = in ? okm : (okm & ~(1u << i));
    r.pa += in ? a_step : 0;
    if (DENSE) {
      r.px += in ? x_step : 0;
    } else {
      int w2_ = r.ow + sow;
      const int c1 = w2_ >= d.OW;
      w2_ -= c1 ? d.OW : 0;
      int h2 = r.oh + soh + c1;
      const int c2 = h2 >= d.OH;
      h2 -= c2 ? d.OH : 0;
      r.ow = in ? w2_ : r.ow;
      r.oh = in ? h2 : r.oh;
      r.px += in ? (int64_t)(sb + c2) * img : 0;
    }
  };
  auto x_addr = [&](const Row& r, int i) -> const T* {  // the tap's address, clamped into the image
    if (DENSE) return r.px;
    const int ih = r.oh * d.stride + kh - d.pad, iw = r.ow * d.stride + kw - d.pad;
    const bool inb = (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
    okm = inb ? (okm | (16u << i)) : (okm & ~(16u << i));
    const int ihc = ih < 0 ? 0 : (ih >= d.H ? d.H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= d.W ? d.W - 1 : iw);
    return r.px + (ihc * d.W + iwc) * d.Cin;
  };
  auto advance = [&]() {
    adv_row(w0, 0);
    adv_row(w1, 1);
    adv_row(w2, 2);
    adv_row(w3, 3);
  };
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  auto load_tile = [&]() {
    ra0 = *reinterpret_cast<const uint4*>(w0.pa);
    rb0 = *reinterpret_cast<const uint4*>(x_addr(w0, 0));
    ra1 = *reinterpret_cast<const uint4*>(w1.pa);
    rb1 = *reinterpret_cast<const uint4*>(x_addr(w1, 1));
    ra2 = *reinterpret_cast<const uint4*>(w2.pa);
    rb2 = *reinterpret_cast<const uint4*>(x_addr(w2, 2));
    ra3 = *reinterpret_cast<const uint4*>(w3.pa);
    rb3 = *reinterpret_cast<const uint4*>(x_addr(w3, 3));
  };
  auto store_row = [&](T* As, T* Bs, int i, const uint4& va, const uint4& vb) {
    const int k = lk + 16 * i;
    const int sw = (lch ^ ((k & 3) << 2)) << 3;
    const bool row_ok = (okm >> i) & 1u;
    const bool tap_ok = DENSE || ((okm >> (4 + i)) & 1u);
    // component-wise: `c ? va : z` on two lvalues selects an ADDRESS and sends both through scratch memory
    const uint32_t ma = (row_ok && a_col_ok) ? 0xFFFFFFFFu : 0u, mb = (row_ok && b_col_ok && tap_ok) ? 0xFFFFFFFFu : 0u;
    *reinterpret_cast<uint4*>(As + k * BM + sw) = make_uint4(va.x & ma, va.y & ma, va.z & ma, va.w & ma);
    *reinterpret_cast<uint4*>(Bs + k * BN + sw) = make_uint4(vb.x & mb, vb.y & mb, vb.z & mb, vb.w & mb);
  };
  auto store_tile = [&](int buf) {
    T* As = lds + buf * (2 * BK * BM);
    T* Bs = As + BK * BM;
    store_row(As, Bs, 0, ra0, rb0);
    store_row(As, Bs, 1, ra1, rb1);
    store_row(As, Bs, 2, ra2, rb2);
    store_row(As, Bs, 3, ra3, rb3);
  };

  f32x16 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = (int)((k_end - k_begin + BK - 1) / BK);
  if (nt > 0) {
    load_tile();
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
    if (t + 1 < nt) {
      advance();
      load_tile();
    }
    const T* As = lds + buf * (2 * BK * BM);
    const T* Bs = As + BK * BM;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v8 af[FI], bf[FJ];
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      auto frag = [&](const T* S, int c0) -> v8 {  // 8 consecutive k of column block c0 (this lane: 4 columns at c0)
        s16x4 h[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int k = ks * 16 + 8 * fh + 4 * half + q;
          const int swz = (k & 3) << 2;
          h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(S + k * BM + ((((c0 >> 3) ^ swz) << 3) | (c0 & 7))));
        }
        return __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7));
      };
#pragma unroll
      for (int i = 0; i < FI; ++i) af[i] = frag(As, m_off + i * 32 + msub + 4 * pc);
#pragma unroll
      for (int j = 0; j < FJ; ++j) bf[j] = frag(Bs, n_off + j * 32 + msub + 4 * pc);
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = T16<T>::mfma32(af[i], bf[j], acc[i][j]);
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: slab[split][(co*Cin + ci)*taps + tap]
  float* slab = d.out + (int64_t)by * d.Cout * Ntot;
  const int taps = d.KH * d.KW;
  const int fr = lane & 31, fq = lane >> 5;
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    const int n = n0 + n_off + j * 32 + fr;
    if (n >= Ntot) continue;
    const int tp = n / d.Cin, cc = n - tp * d.Cin;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + m_off + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
        if (m < d.Cout) slab[((int64_t)m * d.Cin + cc) * taps + tp] = acc[i][j][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Large-tile form for nn.Linear weight gradients (dense, Cout % 256 == 0, Cin % 128 == 0): 256 x 128 x 32 tile, 8 waves of
// 64 x 64, both operand tiles staged [k][m] / [k][n] by LDS-DMA (global_load_lds_dwordx4: no staging registers, no LDS
// store pass) into a 3-stage ring with a counted vmcnt (one K tile stays in flight across the barrier), fragments by the
// same transposing reads as above — the step that took the dense GEMM from the 128^2 register-staged kernel to the
// 256-row LDS-DMA kernel (gemm_big.h, two workgroups per CU).  The weight gradients of the unfrozen ViT are the largest
// kernel of BASELINE config 4 (16 % of its step at ~410 TFLOP/s on the form above).
// A wave-instruction lands 1 KB lane-linearly: 2 k-rows of the A image (512 B each) or 4 of the B image (256 B each); the
// 16-byte chunk swizzle (k & 3) << 2 is applied to the per-lane SOURCE address.  Rows past the split end read a zero page.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint4 g_wgrad_zero[1];

// CONV (round 3): the same kernel for stride-1 convolutions with Cin % 128 == 0 -- a 128-column tile of the implicit im2col
// operand then lies inside ONE tap, so the only difference is the per-lane SOURCE address of the x rows: pixel (b, oh, ow) of
// the K tile, shifted by the tile's tap, or the zero page outside the image (tracked incrementally, 32 pixels per K tile).
template <typename T, bool CONV = false>
__global__ __launch_bounds__(512, 2) void wgrad_dense_big_kernel(const asis_wgrad_desc d) {
  constexpr int TM = 256, TN = 128, TK = 32, NS = 3;
  constexpr int STAGE = TK * (TM + TN);               // 12288 elements = 24 KB
  constexpr int G = 3;                                // LDS-DMA instructions per wave and K tile (2 for A, 1 for B)
  typedef typename T16<T>::v8 v8;
  typedef s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  __shared__ __attribute__((aligned(16))) T lds[NS * STAGE];   // 72 KB: two workgroups per CU

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;                 // 4 x 2 waves of 64 x 64
  const int tiles_n = (CONV ? d.KH * d.KW * d.Cin : d.Cin) / TN;
  const int lin = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int bx = lin % gridDim.x, by = lin / gridDim.x;
  const int tile_m = bx / tiles_n, tile_n = bx - tile_m * tiles_n;
  const int m0 = tile_m * TM, n0 = tile_n * TN;
  const int64_t k_begin = (int64_t)by * d.k_per_split;
  int64_t k_end = k_begin + d.k_per_split;
  if (k_end > d.P) k_end = d.P;
  const int nt = k_end > k_begin ? (int)((k_end - k_begin + TK - 1) / TK) : 0;

  // per-lane sources: A instruction j of this wave = k rows 2 (2 wid + j) + (lane >> 5), chunk lane & 31;
  //                   B instruction     = k rows 4 wid + (lane >> 4), chunk lane & 15
  const T* zp = reinterpret_cast<const T*>(g_wgrad_zero);
  const T* A = reinterpret_cast<const T*>(d.dy);
  const T* X = reinterpret_cast<const T*>(d.x);
  int ka[2], kb;
  const T* pa[2];
  const T* pb;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    ka[j] = 2 * (2 * wid + j) + (lane >> 5);
    const int ch = (lane & 31) ^ ((ka[j] & 3) << 2);
    pa[j] = A + (k_begin + ka[j]) * d.ld_dy + m0 + ch * 8;
  }
  kb = 4 * wid + (lane >> 4);
  const int tap = CONV ? n0 / d.Cin : 0, ci0 = n0 - tap * d.Cin;       // a column tile sits inside one tap
  const int tkh = CONV ? tap / d.KW : 0, tkw = tap - tkh * d.KW;
  const int bchunk = ((lane & 15) ^ ((kb & 3) << 2)) * 8;
  pb = X + (k_begin + kb) * (int64_t)d.Cin + n0 + bchunk;
  // CONV: pixel (cb, coh, cow) of this lane's x row in the NEXT tile to be issued (issue() is called with t = 0, 1, 2, ...)
  int cb = 0, coh = 0, cow = 0;
  if (CONV) {
    const int64_t p0 = k_begin + kb;
    const int ohw = d.OH * d.OW;
    cb = (int)(p0 / ohw);
    const int rem = (int)(p0 - (int64_t)cb * ohw);
    coh = rem / d.OW;
    cow = rem - coh * d.OW;
  }
  const int64_t a_step = (int64_t)TK * d.ld_dy, x_step = (int64_t)TK * d.Cin;
  auto issue = [&](int t) {
    T* st = lds + (t % NS) * STAGE;
    const int64_t kt = k_begin + (int64_t)t * TK;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const T* src = (kt + ka[j] < k_end) ? pa[j] + (int64_t)t * a_step : zp;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(st + (2 * (2 * wid + j)) * TM), 16, 0, 0);
    }
    const T* srcb = zp;
    if (CONV) {
      const int ih = coh + tkh - d.pad, iw = cow + tkw - d.pad;
      if (kt + kb < k_end && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
        srcb = X + (((int64_t)cb * d.H + ih) * d.W + iw) * d.Cin + ci0 + bchunk;
      cow += TK;                                  // the next tile's pixel: TK further in (b, oh, ow) order
      while (cow >= d.OW) { cow -= d.OW; ++coh; }
      while (coh >= d.OH) { coh -= d.OH; ++cb; }
    } else if (kt + kb < k_end) {
      srcb = pb + (int64_t)t * x_step;
    }
    __builtin_amdgcn_global_load_lds((glb_ptr)srcb, (lds_ptr)(st + TK * TM + (4 * wid) * TN), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nt) issue(s0);

  // transposed-read geometry (see wgrad_kernel): 16-lane group g reads a 4 (k) x 16 (m) block
  const int g = lane >> 4, li = lane & 15;
  const int fh = g >> 1, msub = 16 * (g & 1);
  const int q = li >> 2, pc = li & 3;
  for (int t = 0; t < nt; ++t) {
    if (nt - t - 1 >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + NS - 1 < nt) issue(t + NS - 1);
    const T* As = lds + (t % NS) * STAGE;
    const T* Bs = As + TK * TM;
#pragma unroll
    for (int ks = 0; ks < TK / 16; ++ks) {
      // The transposing reads are inline asm, waited for inside the statement: behind an LDS-DMA issue hipcc puts an
      // `s_waitcnt vmcnt(0)` in front of every LDS read IT sees (it cannot prove the read does not touch the bytes in flight),
      // which drained the K tile that had just been issued and serialised the ring (found in the ISA of the first build).
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      uint32_t ad[8];
#pragma unroll
      for (int f = 0; f < 4; ++f) {                      // fragments: A rows i = 0, 1 ; B columns j = 0, 1
        const T* S = f < 2 ? As : Bs;
        const int RS = f < 2 ? TM : TN;
        const int c0 = (f < 2 ? wm * 64 + f * 32 : wn * 64 + (f - 2) * 32) + msub + 4 * pc;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int k = ks * 16 + 8 * fh + 4 * half + q;
          const int swz = (k & 3) << 2;
          ad[2 * f + half] = (uint32_t)(uintptr_t)(lds_tr_ptr)(S + k * RS + ((((c0 >> 3) ^ swz) << 3) | (c0 & 7)));
        }
      }
      s16x4 h[8];
      asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
                   "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5]), "=&v"(h[6]), "=&v"(h[7])
                   : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                   : "memory");
      v8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(h[2 * i], h[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7));
        bf[i] = __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(h[4 + 2 * i], h[5 + 2 * i], 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = T16<T>::mfma32(af[i], bf[j], acc[i][j]);
    }
  }

  const int taps = CONV ? d.KH * d.KW : 1;
  float* slab = d.out + (int64_t)by * d.Cout * d.Cin * taps;
  const int fr = lane & 31, fq = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = (CONV ? ci0 : n0) + wn * 64 + j * 32 + fr;   // input channel
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fq;
        slab[((int64_t)m * d.Cin + n) * taps + tap] = acc[i][j][r];   // the parameter's [Cout][Cin][KH][KW] layout
      }
  }
}

}  // namespace

static inline int wgrad_bme(int Cout) {
  static const int force = [] { const char* e = getenv("ASIS_WGRAD_BME"); return e ? atoi(e) : 0; }();  // lab: 128 = always the full tile
  if (force == 128) return BM;
  return Cout <= 32 ? 32 : (Cout <= 64 ? 64 : BM);
}

extern "C" int asis_wgrad_splits(int64_t P, int Cout, int Ntot) {
  const int64_t tiles = asis_cdiv(Cout, wgrad_bme(Cout)) * asis_cdiv(Ntot, BN);
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
  const int bme = wgrad_bme(d.Cout);
  dim3 grid((unsigned)(asis_cdiv(d.Cout, bme) * asis_cdiv(Ntot, BN)), d.splits), block(NTHREADS);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool dense = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0;
  // ASIS_WGRAD_BIG (default 1): nn.Linear weight gradients with whole 256 x 128 tiles on the LDS-DMA form
  static const int big = [] { const char* e = getenv("ASIS_WGRAD_BIG"); return e ? atoi(e) : 1; }();
  if (big && dense && d.Cout % 256 == 0 && d.Cin % 128 == 0 && d.CoP == d.Cout && d.P >= 1024 && d.k_per_split % 32 == 0) {
    dim3 gb((unsigned)((d.Cout / 256) * (d.Cin / 128)), d.splits);
    if (d.dtype == ASIS_F16) hipLaunchKernelGGL((wgrad_dense_big_kernel<f16>), gb, dim3(512), 0, s, d);
    else hipLaunchKernelGGL((wgrad_dense_big_kernel<bf16>), gb, dim3(512), 0, s, d);
    ASIS_CHECK_LAUNCH("asis_wgrad");
    return ASIS_OK;
  }
  // the same tile form for stride-1 convolutions whose 128-column tiles lie inside one tap (decoder_1 .. 3)
  if (big && !dense && d.stride == 1 && d.Cout % 256 == 0 && d.Cin % 128 == 0 && d.CoP == d.Cout && d.P >= 1024 &&
      d.k_per_split % 32 == 0 && (int64_t)d.B_ * d.H * d.W * d.Cin < (1LL << 40)) {
    dim3 gb((unsigned)((d.Cout / 256) * (Ntot / 128)), d.splits);
    if (d.dtype == ASIS_F16) hipLaunchKernelGGL((wgrad_dense_big_kernel<f16, true>), gb, dim3(512), 0, s, d);
    else hipLaunchKernelGGL((wgrad_dense_big_kernel<bf16, true>), gb, dim3(512), 0, s, d);
    ASIS_CHECK_LAUNCH("asis_wgrad");
    return ASIS_OK;
  }
#define ASIS_WGRAD_LAUNCH(TT, DN)                                                                         \
  do {                                                                                                    \
    if (bme == 32) hipLaunchKernelGGL((wgrad_kernel<TT, DN, 32>), grid, block, 0, s, d);                  \
    else if (bme == 64) hipLaunchKernelGGL((wgrad_kernel<TT, DN, 64>), grid, block, 0, s, d);             \
    else hipLaunchKernelGGL((wgrad_kernel<TT, DN, 128>), grid, block, 0, s, d);                           \
  } while (0)
  if (d.dtype == ASIS_F16) {
    if (dense) ASIS_WGRAD_LAUNCH(f16, true); else ASIS_WGRAD_LAUNCH(f16, false);
  } else {
    if (dense) ASIS_WGRAD_LAUNCH(bf16, true); else ASIS_WGRAD_LAUNCH(bf16, false);
  }
#undef ASIS_WGRAD_LAUNCH
  ASIS_CHECK_LAUNCH("asis_wgrad");
  return ASIS_OK;
}
