// Persistent form of the 8-phase dense GEMM (included by gemm.hip).
//
//   C[M,N] = epilogue( A[M,K] * B[N,K]^T ),  256x256x64 tiles, 8 waves (2x4) of 128x64 on v_mfma_f32_16x16x32
//
// gemm_big.h's 8-phase kernel runs ONE tile per workgroup at one workgroup per CU (128 KB of LDS, 248 VGPRs), so per tile it
// pays, fully exposed: the workgroup launch, the cold fill of the first K tile (every CU bursting 64 KB at once: ~3 us,
// MI355X_MICROARCH.md "prologue HBM burst"), the epilogue and the drain of its stores before the CU is handed on.  At
// K = 1024 a tile is only 16 K tiles (~27 us) long.  Here ONE workgroup per CU walks a static list of tiles and the
// operand stream never stops:
//   * the LDS-DMA of the NEXT tile's first K tile is issued during the phases of the current tile's LAST K tile (the
//     per-lane source pointers are switched two at a time, right before the phase that uses them), so a tile starts with
//     its operands already in LDS;
//   * the epilogue transposes through the LDS stage the last K tile has just vacated (the other stage holds the next tile's
//     K tile 0) and its global stores are not waited for: they drain under the next tile's first K tile, whose phase-0/1
//     waits are dropped (everything they would wait for has landed during the epilogue) — the first counted vmcnt that
//     covers the stores is phase 3's;
//   * the two wave rows are brought level for the epilogue (all 8 waves transpose and store concurrently) and staggered
//     again for the next tile: 3 barrier intervals per tile.
// Tile order: the logical order of gemm_big.h (bands of GROUP_M row tiles, column-major inside a band, a contiguous range per
// XCD); inside an XCD's range workgroup l takes positions l, l + 32, ...: at any time the 32 CUs of an XCD work on 32
// consecutive positions = 8 x 4 / 4 x 8 tiles, the shape that minimises (rows + columns) of operand panels per L2.
//
// Split-precision operands (asis_gemm_desc.A_lo / B_lo, include/asis_hip.h): the K stream of a tile runs over 2 or 3
// K-long parts (A, B), (A_lo, B), (A, B_lo) — only the wave-uniform operand BASE changes between parts.
// Host contract (gemm.hip checks it; anything else runs on gemm_big.h): dense, batch 1, no stats / bias_m,
// K % 64 == 0, N % 8 == 0, every pointer 16-byte aligned, ldc / ldr / ld_aux multiples of 8 (no scalar epilogue here).
#pragma once
#include <type_traits>
#include "asis_common.h"

namespace {

// LNF: which LayerNorm-fold epilogue fields this instance implements (include/asis_hip.h; a separate instance: the plain kernel
// has no register to spare): 0 none, 1 = ln_mr / ln_cs in the 16-bit epilogue (q|k, fc1).  The producer fields (res16 / C_lo /
// rowstats) are NOT built here: with them this kernel spilled ~200 VGPRs into its main loop in every arrangement tried, and the
// one-tile-per-workgroup form (gemm_big.h, LNF) runs the projection within 4 % of this one — asis_gemm sends those launches there.
template <typename T, int LNF = 0>
__global__ __launch_bounds__(512, 1) void gemm_p8_kernel(const asis_gemm_desc d, const int GROUP_M) {
  // (Tried in round 3 and removed: a non-temporal cache policy (LDS-DMA aux = nt) on the operand whose panels an XCD's chunks
  // do not share.  L2 fills went UP 5-30 % with the hint on either operand, and merely having the two aux variants of the
  // builtin behind a wave-uniform branch in the loop cost 40-60 % of the kernel's speed: profiles/r03_gemm_p8_nt_policy.txt.)
  typedef typename T16<T>::v8 v8;
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int STAGE = (BM + BN) * BK;  // elements per LDS stage (64 KB)
  // LNF == 1: + 2 KB behind the stages for the (mean, rstd) pairs of the tile's 256 rows (LDS-DMA at the tile's start: a
  // global load issued in the epilogue itself costs a full loaded-memory round trip, ~1.5 us per tile, in the exposed epilogue)
  __shared__ __attribute__((aligned(16))) T lds[2 * STAGE + 1024];   // + (mean, rstd) block (LNF == 1)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int lr = lane >> 3, lc = lane & 7;      // LDS-DMA: 8 rows x 8 chunks of 16 B per wave-instruction
  const int r16 = lane & 15, q16 = lane >> 4;   // 16x16x32 fragments / accumulators

  // ---- this workgroup's tiles ------------------------------------------------------------------------------------
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nl = gridDim.x >> 3;
  const int xq = ntiles >> 3, xr = ntiles & 7;
  const int xbase = (xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xcnt = xq + (xcd < xr ? 1 : 0);
  if (wl >= xcnt) return;
  auto decode = [&](int p, int& m0, int& n0) {
    const int band = p / (GROUP_M * tiles_n);
    const int first_m = band * GROUP_M;
    const int band_m = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_band = p - band * GROUP_M * tiles_n;
    const int tn = in_band / band_m;
    m0 = (first_m + (in_band - tn * band_m)) * BM;
    n0 = tn * BN;
  };

  const char* A = reinterpret_cast<const char*>(d.A);     // operand bases of the K part being STAGED (wave-uniform)
  const char* B = reinterpret_cast<const char*>(d.B);
  const int nparts = 1 + (d.A_lo ? 1 : 0) + (d.B_lo ? 1 : 0);
  int spart = 0;
  // staging order of the 8-phase loop (gemm_big.h): instructions 0,1 of a wave stage the row halves 0 of both wave rows,
  // 2,3 the halves 1; for B the column halves 0 / 1 of the four wave columns
  auto grp_a = [&](int j) -> int { const int g = wid * 2 + (j & 1); return (g < 8 ? 0 : 128) + (j >> 1) * 64 + (g & 7) * 8; };
  auto grp_b = [&](int j) -> int { const int g = wid * 2 + (j & 1); return (g >> 2) * 64 + (j >> 1) * 32 + (g & 3) * 8; };
  // per-lane source of a DMA instruction as a 32-bit BYTE offset from the (wave-uniform) operand base: the saddr form of
  // global_load_lds, half the address registers of full pointers (the host checks that both operands are < 4 GB)
  // The image swizzle of row (grp + lr), grp a multiple of 8: ((row >> 1) & 7) = (lr >> 1) | ((j & 1) << 2) for both operands.
  // `ln` is the lane id (an OPAQUE copy at the tile switch, so that nothing of this is precomputed and kept live across the
  // main loop: the kernel has no register to spare).
  const uint32_t lda_b = (uint32_t)d.lda * 2u, ldb_b = (uint32_t)d.ldb * 2u;
  auto a_ptr = [&](int j, int m0, int ln) -> uint32_t {
    int gr = m0 + grp_a(j) + (ln >> 3);
    gr = gr < d.M ? gr : d.M - 1;
    return (uint32_t)gr * lda_b + ((((ln & 7) ^ (ln >> 4)) << 4) ^ ((j & 1) << 6));
  };
  auto b_ptr = [&](int j, int n0, int ln) -> uint32_t {
    int gr = n0 + grp_b(j) + (ln >> 3);
    gr = gr < d.N ? gr : d.N - 1;
    return (uint32_t)gr * ldb_b + ((((ln & 7) ^ (ln >> 4)) << 4) ^ ((j & 1) << 6));
  };

  int m0, n0, m0n = 0, n0n = 0;
  int pos = wl;                                  // position inside the XCD's range
  decode(xbase + pos, m0, n0);
  bool has_next = pos + nl < xcnt;
  if (has_next) decode(xbase + pos + nl, m0n, n0n);

  uint32_t asrc[4], bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    asrc[j] = a_ptr(j, m0, lane);
    bsrc[j] = b_ptr(j, n0, lane);
  }
  uint32_t sk0 = 0;     // K offset (bytes) of the K tile being staged
  int g = 0;            // running K tile count: K tile g lives in stage g & 1
  auto dma_a = [&](int stage, int j) {
    __builtin_amdgcn_global_load_lds((glb_ptr)(A + (asrc[j] + sk0)), (lds_ptr)(lds + stage * STAGE + grp_a(j) * BK), 16, 0, 0);
  };
  auto dma_b = [&](int stage, int j) {
    __builtin_amdgcn_global_load_lds((glb_ptr)(B + (bsrc[j] + sk0)), (lds_ptr)(lds + stage * STAGE + BM * BK + grp_b(j) * BK), 16, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dma_a(0, j);
    dma_b(0, j);
  }
  // source of the next K tile to stage: K offset, then the next part (its operand bases), wrapping to part 0 = the next tile
  auto advance = [&]() {
    sk0 += BK * 2;
    if (sk0 == (uint32_t)d.K * 2u) {
      sk0 = 0;
      if (nparts > 1) {
        spart = spart + 1 == nparts ? 0 : spart + 1;
        const bool alo = spart == 1 && d.A_lo, blo = spart != 0 && !alo;
        A = reinterpret_cast<const char*>(alo ? d.A_lo : d.A);
        B = reinterpret_cast<const char*>(blo ? d.B_lo : d.B);
      }
    }
  };
  advance();
  float* const lds_mr = reinterpret_cast<float*>(lds + 2 * STAGE);
  auto dma_mr = [&](int m0t) {   // LNF == 1: thread t fetches float t of the tile's [256][2] (mean, rstd) block
    if constexpr (LNF == 1) {
      if (d.ln_mr) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int t = wid * 64 + ln;
        int row = m0t + (t >> 1);
        row = row < d.M ? row : d.M - 1;
        __builtin_amdgcn_global_load_lds((glb_ptr)(d.ln_mr + 2 * (int64_t)row + (t & 1)), (lds_ptr)(lds_mr + wid * 64), 4, 0, 0);
      }
    }
  };
  dma_mr(m0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int nt = nparts * (d.K / BK);
  f32x4 acc[8][4];      // 16x16 C^T tiles: lane (r16, q16) owns row r16, columns 4 q16 .. 4 q16 + 3
  v8 af[2][4], b0f[4], b1f[4];
  auto rd_a = [&](const T* As, int rh) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int row = (wm * 4 + rh * 2) * 32 + t4 * 16 + r16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[t4 >> 1][(t4 & 1) * 2 + ks] =
            __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(As + row * BK + (((4 * ks + q16) ^ ((row >> 1) & 7)) << 3)));
    }
  };
  auto rd_b = [&](const T* Bs, int ch, v8* bf) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int col = (wn * 2 + ch) * 32 + jj * 16 + r16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        bf[jj * 2 + ks] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Bs + col * BK + (((4 * ks + q16) ^ ((col >> 1) & 7)) << 3)));
    }
  };
  auto mma = [&](int rh, int ch, const v8* bf) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[rh * 4 + t4][ch * 2 + jj] = T16<T>::mfma16(bf[jj * 2 + ks], af[t4 >> 1][(t4 & 1) * 2 + ks], acc[rh * 4 + t4][ch * 2 + jj]);
    __builtin_amdgcn_s_setprio(0);
  };

  const float* const zp = reinterpret_cast<const float*>(g_zero_page);
  for (;;) {
    if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave row by one barrier interval
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: the 8-phase schedule of gemm_big.h over one continuous stream of K tiles -----------------------
    for (int t = 0; t < nt; ++t, ++g) {
      const T* As = lds + (g & 1) * STAGE;
      const T* Bs = As + BM * BK;
      const int ns = (g + 1) & 1;                // stage of the K tile being staged
      const bool last = t + 1 == nt;
      const bool more = !last || has_next;
      const bool sw = last && has_next;          // the K tile being staged is the NEXT tile's first: switch the sources
      const bool first = t == 0;                 // nothing of this K tile is in flight any more (prologue / epilogue waited)
      // phase 0
      rd_a(As, 0);
      rd_b(Bs, 0, b0f);
      if (more) {
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); asrc[0] = a_ptr(0, m0n, ln); asrc[1] = a_ptr(1, m0n, ln); }
        dma_a(ns, 0); dma_a(ns, 1);
        if (!first) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      mma(0, 0, b0f);
      __builtin_amdgcn_s_barrier();
      // phase 1
      rd_b(Bs, 1, b1f);
      if (more) {
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); bsrc[0] = b_ptr(0, n0n, ln); bsrc[1] = b_ptr(1, n0n, ln); }
        dma_b(ns, 0); dma_b(ns, 1);
        if (!first) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      mma(0, 1, b1f);
      __builtin_amdgcn_s_barrier();
      // phase 2
      rd_a(As, 1);
      if (more) {
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); bsrc[2] = b_ptr(2, n0n, ln); bsrc[3] = b_ptr(3, n0n, ln); }
        dma_b(ns, 2); dma_b(ns, 3);
      }
      __builtin_amdgcn_s_barrier();
      mma(1, 1, b1f);
      __builtin_amdgcn_s_barrier();
      // phase 3
      if (more) {
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); asrc[2] = a_ptr(2, m0n, ln); asrc[3] = a_ptr(3, m0n, ln); }
        dma_a(ns, 2); dma_a(ns, 3);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      advance();
      __builtin_amdgcn_s_barrier();
      mma(1, 0, b0f);
      __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // level the two wave rows: all 8 waves run the epilogue together

    // ---- epilogue: through the stage the last K tile has vacated (the other one holds the next tile's K tile 0) -------
    // (the 4 LDS-DMA instructions of that K tile still in flight are waited for here; the barrier behind the epilogue
    // makes them visible to the other waves before anyone reads them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T* const stage_free = lds + ((g - 1) & 1) * STAGE;
    // the epilogue's per-lane indices hang off an opaque copy of the lane id, so that none of its (tile-invariant) address
    // arithmetic is hoisted out of the tile loop and kept in registers across the main loop (29 spilled VGPRs otherwise)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int r16 = lane_e & 15, q16 = lane_e >> 4;
    if (!d.out_f32 && !d.res && !d.res16 && !d.C_lo && !d.rowstats && !d.scale_n && d.act != ASIS_ACT_GELU_GRAD) {
      // 16-bit outputs (q|k, fc1 + GELU): bias and activation in the accumulator layout, converted to 16 bits BEFORE the LDS
      // transposition (8-byte writes, 16-byte reads), 16-byte stores of 8 rows x 128 B
      constexpr int SW16 = 72;                          // slab row in 16-bit elements (144 B: conflict-free 8-byte writes)
      T* slab16 = stage_free + wid * 4096;              // 8 KB per wave
      const int rr8 = lane_e >> 3, c8 = lane_e & 7;
      // LayerNorm folded into the weight (asis_gemm_desc.ln_mr / ln_cs): v = rstd * (acc - mean * cs[n]) + b'[n].  Without it
      // mean = 0, cs = 0, rstd = 1: fma(-0, 0, acc) = acc and fma(acc, 1, b) = acc + b exactly, the same bits as before.
      float4 bj[4], cj[4];
      const bool mrp = LNF == 1 && d.ln_mr != nullptr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int colj = n0 + wn * 64 + j * 16 + 4 * q16;
        bj[j] = (d.bias_n && colj < d.N) ? *reinterpret_cast<const float4*>(d.bias_n + colj) : make_float4(0.f, 0.f, 0.f, 0.f);
        cj[j] = (mrp && colj < d.N) ? *reinterpret_cast<const float4*>(d.ln_cs + colj) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int colr = n0 + wn * 64 + c8 * 8;
      const bool cokr = colr < d.N;
      T* const Cw = reinterpret_cast<T*>(d.C) + colr;
      float2 mrv[8];     // LNF == 1: (mean, rstd) of this lane's eight rows, fetched before the first slab pass
      if constexpr (LNF == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int rowl = (wm * 4 + (k >> 1)) * 32 + (k & 1) * 16 + r16;   // staged by dma_mr at this tile's start
          mrv[k] = mrp ? *reinterpret_cast<const float2*>(lds_mr + 2 * rowl) : make_float2(0.f, 1.f);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          float nmean = -0.f, rstd = 1.f;
          if constexpr (LNF == 1) {
            nmean = -mrv[2 * i + ii].x;
            rstd = mrv[2 * i + ii].y;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 v;
            if constexpr (LNF == 1) {
              v = make_float4(__builtin_fmaf(__builtin_fmaf(nmean, cj[j].x, acc[2 * i + ii][j][0]), rstd, bj[j].x),
                              __builtin_fmaf(__builtin_fmaf(nmean, cj[j].y, acc[2 * i + ii][j][1]), rstd, bj[j].y),
                              __builtin_fmaf(__builtin_fmaf(nmean, cj[j].z, acc[2 * i + ii][j][2]), rstd, bj[j].z),
                              __builtin_fmaf(__builtin_fmaf(nmean, cj[j].w, acc[2 * i + ii][j][3]), rstd, bj[j].w));
            } else {
              v = make_float4(acc[2 * i + ii][j][0] + bj[j].x, acc[2 * i + ii][j][1] + bj[j].y,
                              acc[2 * i + ii][j][2] + bj[j].z, acc[2 * i + ii][j][3] + bj[j].w);
            }
            if (d.act == ASIS_ACT_GELU) gelu_erf4(v.x, v.y, v.z, v.w);
            else if (d.act == ASIS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            uint2 pk;
            pk.x = pack2<T>(v.x, v.y);
            pk.y = pack2<T>(v.z, v.w);
            *reinterpret_cast<uint2*>(slab16 + (ii * 16 + r16) * SW16 + 16 * j + 4 * q16) = pk;
          }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int lrow = p * 8 + rr8;
          const int row = m0 + (wm * 4 + i) * 32 + lrow;
          const uint4 w = *reinterpret_cast<const uint4*>(slab16 + lrow * SW16 + c8 * 8);
          if (row < d.M && cokr) *reinterpret_cast<uint4*>(Cw + (int64_t)row * d.ldc) = w;
        }
      }
    } else {
      // fp32 slab (LayerScale + fp32 residual outputs, GELU' fused input gradients, fp32 outputs): 32 rows x 64 columns per
      // pass group in an UNPADDED slab (8 KB per wave: the vacated stage has room for exactly that), 16-byte chunk index
      // XOR (row & 7): the b128 writes of the accumulator layout (8 consecutive rows, one chunk) and the b128 reads of the
      // row layout (16 chunks of one row per 16 lanes) are both conflict-free
      float* slab = reinterpret_cast<float*>(stage_free) + wid * 2048;
      const int rr = lane_e >> 4, ch = lane_e & 15;
      const int col = n0 + wn * 64 + ch * 4;
      const bool cok = col < d.N;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (cok && d.bias_n) b4 = *reinterpret_cast<const float4*>(d.bias_n + col);
      if (cok && d.scale_n) s4 = *reinterpret_cast<const float4*>(d.scale_n + col);
      const bool has_aux = d.act == ASIS_ACT_GELU_GRAD;
      const int colc = cok ? col : 0;
      const float* const resp = d.res ? d.res + colc : zp;
      const int64_t ldr_e = d.res ? d.ldr : 0;
      // LayerNorm-fold chain (include/asis_hip.h): residual as two 16-bit planes, output as two planes + per-row partial sums
      const T* const r16h = reinterpret_cast<const T*>(d.res16) + colc;
      const T* const r16l = reinterpret_cast<const T*>(d.res16_lo) + colc;
      const bool res_planes = LNF == 2 && d.res16 != nullptr;
      T* const Clo = LNF == 2 ? reinterpret_cast<T*>(d.C_lo) : nullptr;
      float* const rstats = LNF == 2 ? d.rowstats : nullptr;
      const int sgroups = (d.N + 63) >> 6, sgrp = (n0 + wn * 64) >> 6;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int lrow = ii * 16 + r16;
            *reinterpret_cast<float4*>(slab + lrow * 64 + (((4 * j + q16) ^ (lrow & 7)) << 2)) =
                make_float4(acc[2 * i + ii][j][0], acc[2 * i + ii][j][1], acc[2 * i + ii][j][2], acc[2 * i + ii][j][3]);
          }
        float4 r4[8];
        uint2 pw[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int row = m0 + (wm * 4 + i) * 32 + p * 4 + rr;
          const int rowc = row < d.M ? row : d.M - 1;
          if (LNF == 2 && res_planes) {   // (hi.xy | lo.xy) bit patterns ride in the float4's registers until the pass that uses them
            const uint2 h = *reinterpret_cast<const uint2*>(r16h + (int64_t)rowc * d.ldr16);
            const uint2 l = *reinterpret_cast<const uint2*>(r16l + (int64_t)rowc * d.ldr16);
            r4[p] = make_float4(__builtin_bit_cast(float, h.x), __builtin_bit_cast(float, h.y), __builtin_bit_cast(float, l.x),
                                __builtin_bit_cast(float, l.y));
          } else {
            r4[p] = *reinterpret_cast<const float4*>(resp + (int64_t)rowc * ldr_e);
          }
          pw[p] = make_uint2(0u, 0u);
        }
        if (has_aux) {
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int row = m0 + (wm * 4 + i) * 32 + p * 4 + rr;
            const int rowc = row < d.M ? row : d.M - 1;
            pw[p] = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(d.aux) + (int64_t)rowc * d.ld_aux + colc);
          }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int lrow = p * 4 + rr;
          const int row = m0 + (wm * 4 + i) * 32 + lrow;
          float4 v = *reinterpret_cast<const float4*>(slab + lrow * 64 + ((ch ^ (lrow & 7)) << 2));
          if (LNF == 2 && res_planes) {   // hi + lo is exact in fp32 (two non-overlapping 11-bit pieces)
            float h0, h1, h2, h3, l0, l1, l2, l3;
            unpack2<T>(__builtin_bit_cast(uint32_t, r4[p].x), h0, h1);
            unpack2<T>(__builtin_bit_cast(uint32_t, r4[p].y), h2, h3);
            unpack2<T>(__builtin_bit_cast(uint32_t, r4[p].z), l0, l1);
            unpack2<T>(__builtin_bit_cast(uint32_t, r4[p].w), l2, l3);
            r4[p] = make_float4(h0 + l0, h1 + l1, h2 + l2, h3 + l3);
          }
          float st_s = 0.f, st_q = 0.f;
          if (row < d.M && cok) {
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            if (d.act == ASIS_ACT_GELU) gelu_erf4(v.x, v.y, v.z, v.w);
            else if (has_aux) {  // input-gradient GEMM of fc2 fused with GELU's backward
              float g0, g1, g2, g3;
              unpack2<T>(pw[p].x, g0, g1);
              unpack2<T>(pw[p].y, g2, g3);
              v.x *= gelu_erf_grad_fast(g0); v.y *= gelu_erf_grad_fast(g1); v.z *= gelu_erf_grad_fast(g2); v.w *= gelu_erf_grad_fast(g3);
            }
            else if (d.act == ASIS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w;
            v.x += r4[p].x; v.y += r4[p].y; v.z += r4[p].z; v.w += r4[p].w;
            if (d.out_f32) {
              *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.C) + (int64_t)row * d.ldc + col) = v;
            } else {
              uint2 pk;
              pk.x = pack2<T>(v.x, v.y);
              pk.y = pack2<T>(v.z, v.w);
              *reinterpret_cast<uint2*>(reinterpret_cast<T*>(d.C) + (int64_t)row * d.ldc + col) = pk;
              if (LNF == 2 && Clo) {
                uint2 pl;
                pl.x = pack2<T>(lo_part<T>(v.x), lo_part<T>(v.y));
                pl.y = pack2<T>(lo_part<T>(v.z), lo_part<T>(v.w));
                *reinterpret_cast<uint2*>(Clo + (int64_t)row * d.ldc + col) = pl;
              }
            }
            if constexpr (LNF == 2) {
              st_s = (v.x + v.y) + (v.z + v.w);
              st_q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
          }
          if constexpr (LNF == 2) {
            if (rstats) {   // wave-uniform; the 16 lanes of a DPP row hold the 64 columns of one output row
              st_s = row16_sum(st_s);
              st_q = row16_sum(st_q);
              if (ch == 0 && row < d.M && sgrp < sgroups)   // tiles that overhang N: no group past the last one
                *reinterpret_cast<float2*>(rstats + ((int64_t)row * sgroups + sgrp) * 2) = make_float2(st_s, st_q);
            }
          }
        }
      }
    }
    if (!has_next) break;
    // next tile: its K tile 0 is in stage g & 1; the slabs are dead once every wave is past this barrier, so the first
    // phase may stage K tile 1 over them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    m0 = m0n; n0 = n0n;
    dma_mr(m0);      // every wave is past the barrier above: the epilogue's reads of the previous tile's pairs are done
    pos += nl;
    has_next = pos + nl < xcnt;
    if (has_next) decode(xbase + pos + nl, m0n, n0n);
  }
}

}  // namespace
