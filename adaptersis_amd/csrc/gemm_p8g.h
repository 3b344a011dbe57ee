// GROUPED form of the persistent 8-phase dense GEMM (included by gemm.hip): ONE launch walks the tiles of up to
// ASIS_GEMM_GROUP_MAX independent problems (each optionally batched), e.g. the q|k projection and the two batched V^T
// projections of a stacked attention (`dinov2/layers/attention.py:58`: same A rows / same weights, three outputs):
//
//     q|k   42 348 x 2048 x 1024  = 1328 tiles = 5.19 rounds of 256 CUs   (13 % idle tail alone)
//     V^T   12 x (1024 x 1768 x 1024), twice = 2 x 336 tiles = 1.31 rounds each (34 % idle tail each)
//     together 2000 tiles = 7.81 rounds (2.3 % tail), no launch boundary, no side stream.
//
// Same main loop, LDS image, staging order, epilogues and arithmetic order as gemm_p8.h (results are bit-identical to the
// per-problem launches); what changes is that a workgroup's NEXT tile may belong to another problem: the operand bases, leading
// dimensions, clamps (M, N) and the K stream length are per-tile state, switched where gemm_p8.h switches its per-lane source
// offsets (the wrap of `advance`), and the epilogue reads its parameters from the problem record of the tile it finishes.
// Problem records live in the kernel-argument segment (scalar loads, wave-uniform index).
// Tile order: the problems' tile lists concatenated (each in the grouped raster of gemm_big.h, batch outermost); an XCD takes a
// contiguous range of that list, workgroup l of the XCD positions l, l + 32, ... (see gemm_p8.h).
#pragma once
#include "asis_common.h"

#define ASIS_GEMM_GROUP_MAX 8

namespace {

struct p8g_prob {
  const char* A; const char* B; char* C;
  const char* A_lo; const char* B_lo;
  const float* bias_n; const float* bias_m; const float* scale_n; const float* res; const char* aux;
  int64_t strideA, strideB, strideC, strideR;   // bytes per batch step (A, B, C) / floats (R)
  uint32_t lda_b, ldb_b;                         // bytes
  int32_t ldc, ldr, ld_aux;                      // elements
  int32_t M, N, K, nparts;
  int32_t act, out_f32;
  int32_t tile0, tiles_m, tiles_n, per_batch;    // first position in the group's list; tiles per batch entry
  int32_t pad_;
};
struct p8g_args {
  int32_t nprob, ntiles;
  p8g_prob p[ASIS_GEMM_GROUP_MAX];
};

template <typename T>
__global__ __launch_bounds__(512, 1) void gemm_p8g_kernel(const p8g_args g, const int GROUP_M) {
  typedef typename T16<T>::v8 v8;
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int STAGE = (BM + BN) * BK;  // elements per LDS stage (64 KB)
  __shared__ __attribute__((aligned(16))) T lds[2 * STAGE];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int r16 = lane & 15, q16 = lane >> 4;

  // ---- this workgroup's positions ----------------------------------------------------------------------------------
  const int ntiles = g.ntiles;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nl = gridDim.x >> 3;
  const int xq = ntiles >> 3, xr = ntiles & 7;
  const int xbase = (xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xcnt = xq + (xcd < xr ? 1 : 0);
  if (wl >= xcnt) return;
  // position -> (problem, batch entry, tile origin); everything wave-uniform (scalar)
  auto locate = [&](int p, int& pi, int& bz, int& m0, int& n0) {
    pi = 0;
#pragma unroll
    for (int i = 1; i < ASIS_GEMM_GROUP_MAX; ++i)
      if (i < g.nprob && p >= g.p[i].tile0) pi = i;
    const int tiles_m = g.p[pi].tiles_m, tiles_n = g.p[pi].tiles_n;
    int local = p - g.p[pi].tile0;
    bz = local / g.p[pi].per_batch;
    local -= bz * g.p[pi].per_batch;
    const int band = local / (GROUP_M * tiles_n);
    const int first_m = band * GROUP_M;
    const int band_m = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_band = local - band * GROUP_M * tiles_n;
    const int tn = in_band / band_m;
    m0 = (first_m + (in_band - tn * band_m)) * BM;
    n0 = tn * BN;
  };

  auto grp_a = [&](int j) -> int { const int gg = wid * 2 + (j & 1); return (gg < 8 ? 0 : 128) + (j >> 1) * 64 + (gg & 7) * 8; };
  auto grp_b = [&](int j) -> int { const int gg = wid * 2 + (j & 1); return (gg >> 2) * 64 + (j >> 1) * 32 + (gg & 3) * 8; };
  // per-lane source of a DMA instruction as a 32-bit BYTE offset from the (wave-uniform) operand base (see gemm_p8.h)
  auto a_ptr = [&](int j, int m0, int ln, int M, uint32_t lda_b) -> uint32_t {
    int gr = m0 + grp_a(j) + (ln >> 3);
    gr = gr < M ? gr : M - 1;
    return (uint32_t)gr * lda_b + ((((ln & 7) ^ (ln >> 4)) << 4) ^ ((j & 1) << 6));
  };
  auto b_ptr = [&](int j, int n0, int ln, int N, uint32_t ldb_b) -> uint32_t {
    int gr = n0 + grp_b(j) + (ln >> 3);
    gr = gr < N ? gr : N - 1;
    return (uint32_t)gr * ldb_b + ((((ln & 7) ^ (ln >> 4)) << 4) ^ ((j & 1) << 6));
  };

  int pi, bz, m0, n0, pin = 0, bzn = 0, m0n = 0, n0n = 0;
  int pos = wl;
  locate(xbase + pos, pi, bz, m0, n0);
  bool has_next = pos + nl < xcnt;
  if (has_next) locate(xbase + pos + nl, pin, bzn, m0n, n0n);

  // ---- staging state: the problem / batch entry whose K stream is being staged --------------------------------------
  const char *sA_hi, *sB_hi, *sA_lo, *sB_lo;   // part bases of the tile being staged
  const char *A, *B;                           // bases of the K part being staged
  uint32_t sKb;                                // K bytes of that problem
  int s_nparts, spart = 0;
  auto stage_problem = [&](int p, int z) {
    const int64_t oa = (int64_t)z * g.p[p].strideA, ob = (int64_t)z * g.p[p].strideB;
    sA_hi = g.p[p].A + oa;
    sB_hi = g.p[p].B + ob;
    sA_lo = g.p[p].A_lo ? g.p[p].A_lo + oa : nullptr;
    sB_lo = g.p[p].B_lo ? g.p[p].B_lo + ob : nullptr;
    sKb = (uint32_t)g.p[p].K * 2u;
    s_nparts = g.p[p].nparts;
    A = sA_hi;
    B = sB_hi;
  };
  stage_problem(pi, bz);

  uint32_t asrc[4], bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    asrc[j] = a_ptr(j, m0, lane, g.p[pi].M, g.p[pi].lda_b);
    bsrc[j] = b_ptr(j, n0, lane, g.p[pi].N, g.p[pi].ldb_b);
  }
  uint32_t sk0 = 0;
  int gk = 0;           // running K tile count: K tile gk lives in stage gk & 1
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
  // source of the next K tile to stage: K offset, then the next part, wrapping to part 0 of the NEXT tile (its problem's bases)
  auto advance = [&]() {
    sk0 += BK * 2;
    if (sk0 == sKb) {
      sk0 = 0;
      spart = spart + 1 == s_nparts ? 0 : spart + 1;
      if (spart == 0) {
        if (has_next) stage_problem(pin, bzn);
      } else {
        const bool alo = spart == 1 && sA_lo, blo = !alo;
        A = alo ? sA_lo : sA_hi;
        B = blo ? sB_lo : sB_hi;
      }
    }
  };
  advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc[8][4];
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

    const int nt = g.p[pi].nparts * (g.p[pi].K / BK);
    for (int t = 0; t < nt; ++t, ++gk) {
      const T* As = lds + (gk & 1) * STAGE;
      const T* Bs = As + BM * BK;
      const int ns = (gk + 1) & 1;
      const bool last = t + 1 == nt;
      const bool more = !last || has_next;
      const bool sw = last && has_next;          // the K tile being staged is the NEXT tile's first: switch the per-lane sources
      const bool first = t == 0;
      // phase 0
      rd_a(As, 0);
      rd_b(Bs, 0, b0f);
      if (more) {
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); asrc[0] = a_ptr(0, m0n, ln, g.p[pin].M, g.p[pin].lda_b); asrc[1] = a_ptr(1, m0n, ln, g.p[pin].M, g.p[pin].lda_b); }
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
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); bsrc[0] = b_ptr(0, n0n, ln, g.p[pin].N, g.p[pin].ldb_b); bsrc[1] = b_ptr(1, n0n, ln, g.p[pin].N, g.p[pin].ldb_b); }
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
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); bsrc[2] = b_ptr(2, n0n, ln, g.p[pin].N, g.p[pin].ldb_b); bsrc[3] = b_ptr(3, n0n, ln, g.p[pin].N, g.p[pin].ldb_b); }
        dma_b(ns, 2); dma_b(ns, 3);
      }
      __builtin_amdgcn_s_barrier();
      mma(1, 1, b1f);
      __builtin_amdgcn_s_barrier();
      // phase 3
      if (more) {
        if (sw) { int ln = lane; asm volatile("" : "+v"(ln)); asrc[2] = a_ptr(2, m0n, ln, g.p[pin].M, g.p[pin].lda_b); asrc[3] = a_ptr(3, m0n, ln, g.p[pin].M, g.p[pin].lda_b); }
        dma_a(ns, 2); dma_a(ns, 3);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      advance();
      __builtin_amdgcn_s_barrier();
      mma(1, 0, b0f);
      __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // level the two wave rows: all 8 waves run the epilogue together

    // ---- epilogue of the tile (problem pi, batch entry bz): through the stage the last K tile has vacated --------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T* const stage_free = lds + ((gk - 1) & 1) * STAGE;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int r16 = lane_e & 15, q16 = lane_e >> 4;
    const int pM = g.p[pi].M, pN = g.p[pi].N, p_act = g.p[pi].act, p_ldc = g.p[pi].ldc;
    const float* const p_bias_n = g.p[pi].bias_n;
    const float* const p_bias_m = g.p[pi].bias_m;
    const float* const p_scale_n = g.p[pi].scale_n;
    const float* const p_res = g.p[pi].res ? g.p[pi].res + (int64_t)bz * g.p[pi].strideR : nullptr;
    char* const p_C = g.p[pi].C + (int64_t)bz * g.p[pi].strideC;
    const bool p_o32 = g.p[pi].out_f32 != 0;
    if (!p_o32 && !p_res && !p_scale_n && p_act != ASIS_ACT_GELU_GRAD) {
      // 16-bit outputs (q|k, V^T, fc1 + GELU): bias (per column and / or per row) and activation in the accumulator layout,
      // converted to 16 bits BEFORE the LDS transposition, 16-byte stores of 8 rows x 128 B
      constexpr int SW16 = 72;
      T* slab16 = stage_free + wid * 4096;
      const int rr8 = lane_e >> 3, c8 = lane_e & 7;
      float4 bj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int colj = n0 + wn * 64 + j * 16 + 4 * q16;
        bj[j] = (p_bias_n && colj < pN) ? *reinterpret_cast<const float4*>(p_bias_n + colj) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int colr = n0 + wn * 64 + c8 * 8;
      const bool cokr = colr < pN;
      T* const Cw = reinterpret_cast<T*>(p_C) + colr;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          float bm1 = 0.f;
          if (p_bias_m) {   // per-row bias (V^T problems: rows are the output features)
            const int rowb = m0 + (wm * 4 + i) * 32 + ii * 16 + r16;
            bm1 = p_bias_m[rowb < pM ? rowb : pM - 1];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 v = make_float4(acc[2 * i + ii][j][0] + bj[j].x + bm1, acc[2 * i + ii][j][1] + bj[j].y + bm1,
                                   acc[2 * i + ii][j][2] + bj[j].z + bm1, acc[2 * i + ii][j][3] + bj[j].w + bm1);
            if (p_act == ASIS_ACT_GELU) gelu_erf4(v.x, v.y, v.z, v.w);
            else if (p_act == ASIS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
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
          if (row < pM && cokr) *reinterpret_cast<uint4*>(Cw + (int64_t)row * p_ldc) = w;
        }
      }
    } else {
      // fp32 slab epilogue (LayerScale + fp32 residual, GELU' fused input gradients, fp32 outputs): as gemm_p8.h
      float* slab = reinterpret_cast<float*>(stage_free) + wid * 2048;
      const int rr = lane_e >> 4, ch = lane_e & 15;
      const int col = n0 + wn * 64 + ch * 4;
      const bool cok = col < pN;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (cok && p_bias_n) b4 = *reinterpret_cast<const float4*>(p_bias_n + col);
      if (cok && p_scale_n) s4 = *reinterpret_cast<const float4*>(p_scale_n + col);
      const bool has_aux = p_act == ASIS_ACT_GELU_GRAD;
      const int colc = cok ? col : 0;
      const float* const resp = p_res ? p_res + colc : zp;
      const int64_t ldr_e = p_res ? g.p[pi].ldr : 0;
      const T* const auxp = reinterpret_cast<const T*>(g.p[pi].aux);
      const int ld_aux = g.p[pi].ld_aux;
      const float* const bmp = p_bias_m ? p_bias_m : zp;
      const int bm_e = p_bias_m ? 1 : 0;
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
        float bmv[8];
        uint2 pw[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int row = m0 + (wm * 4 + i) * 32 + p * 4 + rr;
          const int rowc = row < pM ? row : pM - 1;
          r4[p] = *reinterpret_cast<const float4*>(resp + (int64_t)rowc * ldr_e);
          bmv[p] = bmp[rowc * bm_e];
          pw[p] = make_uint2(0u, 0u);
        }
        if (has_aux) {
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int row = m0 + (wm * 4 + i) * 32 + p * 4 + rr;
            const int rowc = row < pM ? row : pM - 1;
            pw[p] = *reinterpret_cast<const uint2*>(auxp + (int64_t)rowc * ld_aux + colc);
          }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int lrow = p * 4 + rr;
          const int row = m0 + (wm * 4 + i) * 32 + lrow;
          float4 v = *reinterpret_cast<const float4*>(slab + lrow * 64 + ((ch ^ (lrow & 7)) << 2));
          if (row < pM && cok) {
            const float bm1 = bmv[p];
            v.x += b4.x + bm1; v.y += b4.y + bm1; v.z += b4.z + bm1; v.w += b4.w + bm1;
            if (p_act == ASIS_ACT_GELU) gelu_erf4(v.x, v.y, v.z, v.w);
            else if (has_aux) {
              float g0, g1, g2, g3;
              unpack2<T>(pw[p].x, g0, g1);
              unpack2<T>(pw[p].y, g2, g3);
              v.x *= gelu_erf_grad_fast(g0); v.y *= gelu_erf_grad_fast(g1); v.z *= gelu_erf_grad_fast(g2); v.w *= gelu_erf_grad_fast(g3);
            }
            else if (p_act == ASIS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w;
            v.x += r4[p].x; v.y += r4[p].y; v.z += r4[p].z; v.w += r4[p].w;
            if (p_o32) {
              *reinterpret_cast<float4*>(reinterpret_cast<float*>(p_C) + (int64_t)row * p_ldc + col) = v;
            } else {
              uint2 pk;
              pk.x = pack2<T>(v.x, v.y);
              pk.y = pack2<T>(v.z, v.w);
              *reinterpret_cast<uint2*>(reinterpret_cast<T*>(p_C) + (int64_t)row * p_ldc + col) = pk;
            }
          }
        }
      }
    }
    if (!has_next) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    pi = pin; bz = bzn; m0 = m0n; n0 = n0n;
    pos += nl;
    has_next = pos + nl < xcnt;
    if (has_next) locate(xbase + pos + nl, pin, bzn, m0n, n0n);
  }
}

}  // namespace
