// Fused softmax-attention backward for the DINOv2 blocks (head dim 64), software-pipelined LDS-DMA form (round 5).
//
//   P = softmax(c Q K^T),  O = P V          (c = head_dim^-0.5, dinov2/layers/attention.py:60-66)
//   dV = P^T dO ;  dP = dO V^T ;  dS = c P (dP - D),  D[q] = sum_d dO[q,d] O[q,d] ;  dQ = dS K ;  dK = dS^T Q
//
// Same mathematics and the same two-kernel split as the round-1 form it replaces (query-stationary dQ kernel, key-stationary dK / dV
// kernel: seven MFMA products instead of five, no atomics, bit-reproducible — with head dim 64 the f32 atomics of a
// one-kernel form would move 86 MB of adds per key block, above the chip's atomic rate for the whole pass), rebuilt on
// the forward kernel's structure (attention.hip: attn_fwd_pipe_kernel):
//   * every operand is row-major [tokens, H * 64]: the streamed tiles are staged [token][d] by LDS-DMA
//     (global_load_lds, no staging registers, no ds_write) into rings of three 8-KiB slots, two tiles ahead; the products
//     that reduce over the TOKEN index (dQ += dS K, dV += P^T dO, dK += dS^T Q) take their A operand from the same LDS
//     image by the transposing read ds_read_b64_tr_b16 — the K^T / Q^T / dO^T tensors of the old form and the three
//     asis_transpose_tokens passes per block are gone;
//   * one swizzle serves both kinds of read: chunk slot = chunk ^ f(row), f(row) = row bits (1, 2, 3) in slot bits
//     (2, 1, 0) — the eight same-parity rows of a ds_read_b128 service group (rows in the bit-2/3-swapped order of the
//     accumulator -> B-operand hand-off) take eight different slots, and the eight 32-byte pieces a 32-lane half of a
//     transposing read touches (4 rows x 2 column blocks) fall into eight different bank octets (row bit 0 picks the
//     128-byte half of the bank line, row bit 1 the slot's bit 2, the column block its bit 1).  The first version put row
//     bit 1 into slot bit 1 and measured 2-way conflicts on every transposing read (SQ_LDS_BANK_CONFLICT = 25 % of
//     SQ_LDS_IDX_ACTIVE);
//   * the DMA is issued from an asm statement (hipcc puts an s_waitcnt vmcnt(0) in front of the first LDS read behind a
//     DMA builtin it cannot disambiguate; here the only wait is the counted one in front of the tile's barrier) and the
//     tile loop is unrolled three times, so every LDS address is ONE of six per-lane registers (four row-read bases, two
//     transposed-read bases: the swizzle is additive in everything but the k-step of a row read and the d block of a
//     transposed one) plus an immediate;
//   * pipelined in half tiles of 32 tokens: the dP chain of half-step g and the score chain of half-step g + 1 go to the
//     matrix pipe first, the exponentials of half-step g run under them, then dS (and P) feed the token-reducing products;
//     one barrier per 64-token tile;
//   * row constants ride in the MFMA chain: dP starts from -c D (a lane-constant accumulator block in the dQ kernel, four
//     broadcast LDS reads per chain in the dK / dV kernel), c is folded into the dO / V fragments when it is a power of
//     two (head dim 64: exactly 1/8), so dS = P * dP' is one multiply per score.
// Two stacked token batches per launch (B1 images of N1 tokens, then B2 of N2: the two ViT passes of a step) as in the
// forward: 21 504 wave-sized units on 2 048 wave slots are 10.5 rounds instead of 2 x 5.25.
#include <type_traits>

#include "asis_common.h"
#include "attn_tiles.h"

namespace {

using namespace attn_tiles;

// ---- D'[b, h, q] = -c * sum_d dO[q, h*64 + d] * O[q, h*64 + d]  (the initial accumulator of the dP chains) -----------
template <typename T>
__global__ __launch_bounds__(256) void attn_rowdot_neg_kernel(const T* __restrict__ o, int64_t ldo, const T* __restrict__ dO,
                                                              int64_t lddo, float* __restrict__ D, int B1, int N1, int B2,
                                                              int N2, int H, float negc) {
  typedef typename T16<T>::v8 v8;
  const int64_t R1 = (int64_t)B1 * N1, total = (R1 + (int64_t)B2 * N2) * H * 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int part = (int)(i & 7);
    const int64_t th = i >> 3;
    const int h = (int)(th % H);
    const int64_t tok = th / H;  // stacked row
    const v8 a = *reinterpret_cast<const v8*>(o + tok * ldo + h * HD + part * 8);
    const v8 g = *reinterpret_cast<const v8*>(dO + tok * lddo + h * HD + part * 8);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (float)a[k] * (float)g[k];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (part == 0) {
      int64_t off;
      if (tok < R1) {
        const int b = (int)(tok / N1), q = (int)(tok - (int64_t)b * N1);
        off = ((int64_t)b * H + h) * N1 + q;
      } else {
        const int64_t t2 = tok - R1;
        const int b = (int)(t2 / N2), q = (int)(t2 - (int64_t)b * N2);
        off = R1 * H + ((int64_t)b * H + h) * N2 + q;
      }
      D[off] = s * negc;
    }
  }
}

// ---- dQ ---------------------------------------------------------------------------------------------------------------
// workgroup = 4 waves = 128 queries of one (image, head); streams 64-key tiles of K and V.
//   S^T = K Q^T, dP'^T = V (c dO)^T - c D   (A = K / V rows from LDS, B = Q / dO fragments in registers; the accumulator's
//                                            column is the query, so lse2[q] and D[q] are per-lane scalars)
//   dS'^T = P^T * dP'^T,  dQ^T += K^T dS'^T  (A = K^T by transposing reads of the K tile, B = dS'^T from the accumulator)
template <typename T, bool CFOLD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_pipe_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                  const T* __restrict__ v, int64_t ld,
                                                                  const T* __restrict__ dO, int64_t lddo,
                                                                  const float* __restrict__ lse2, const float* __restrict__ Dn,
                                                                  T* __restrict__ dq, int64_t lddq, int H, int B1, int N1,
                                                                  int N2, float scale, float scale_log2e) {
  typedef typename T16<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T lds[6 * TILE];   // K ring [3][64][64] | V ring [3][64][64] = 48 KiB
  T *const K0 = lds, *const K1 = lds + TILE, *const K2 = lds + 2 * TILE;
  T *const V0 = lds + 3 * TILE, *const V1 = lds + 4 * TILE, *const V2 = lds + 5 * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  // XCD-aware order (attention.hip): the query tiles of one (image, head) share K / V and stay on one XCD's L2
  const int ntl = gridDim.x;
  const int lin = xcd_remap(blockIdx.x + ntl * (blockIdx.y + gridDim.y * blockIdx.z), ntl * gridDim.y * gridDim.z);
  const int head = (lin / ntl) % gridDim.y, b = lin / (ntl * gridDim.y);
  const int N = b < B1 ? N1 : N2;
  const int64_t row0 = b < B1 ? (int64_t)b * N1 : (int64_t)B1 * N1 + (int64_t)(b - B1) * N2;
  const int64_t st0 = (b < B1 ? (int64_t)b * H * N1 : (int64_t)B1 * H * N1 + (int64_t)(b - B1) * H * N2) + (int64_t)head * N;
  const int q_base = (lin % ntl) * 128 + wid * 32;
  if ((lin % ntl) * 128 >= N) return;   // workgroup-uniform (the shorter of two stacked batches)
  const int qi = q_base + fr;
  const bool qok = qi < N;
  const int qc = qok ? qi : N - 1;

  v8 qf[4], gf[4];
  {
    const T* qp = q + (row0 + qc) * ld + head * HD + 8 * fh;
    const T* gp = dO + (row0 + qc) * lddo + head * HD + 8 * fh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(qp + 16 * s));
      gf[s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(gp + 16 * s));
    }
    if (CFOLD) {   // c is a power of two: c dO is exact
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) gf[s][j] = (T)((float)gf[s][j] * scale);
    }
  }
  const float nlse = -lse2[st0 + qc];
  f32x16 nd;   // -c D[q] in every element: the C operand that opens each dP chain
  {
    const float dn = CFOLD ? Dn[st0 + qc] : Dn[st0 + qc] / scale;   // Dn = -c D; unfolded form: -D
#pragma unroll
    for (int r = 0; r < 16; ++r) nd[r] = dn;
  }

  TileDma<T> kd, vd;
  kd.init(k + row0 * ld + head * HD, ld, wid, lane);
  vd.init(v + row0 * ld + head * HD, ld, wid, lane);
  const int nt = (N + TT - 1) / TT;
  LaneOff lo;
  lo.init(lane);
  const int hstep = ((lane >> 4) & 1) ? -16 : 16;

  f32x16 acc[2] = {zero16(), zero16()};
  f32x16 sA, sB;   // scores of the half-step in flight and of the next one

  auto s_chain = [&](const T* Kt, int h, f32x16& s_out) {
    f32x16 sa = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) sa = T16<T>::mfma32(row_frag<T>(Kt, lo, h, s), qf[s], sa);
    s_out = sa;
  };
  // One half-step (32 keys): dP' of this half and the scores of the next one (Kn, hn -> s_nxt) go to the matrix pipe first, the
  // exponentials of this half run under them, then dS' and the dQ products.  No masks: keys >= N are zero rows of the K tile.
  auto half = [&](const T* Kt, const T* Vt, int h, const f32x16& sc, const T* Kn, int hn, f32x16& s_nxt) {
    f32x16 dp = nd;
#pragma unroll
    for (int s = 0; s < 4; ++s) dp = T16<T>::mfma32(row_frag<T>(Vt, lo, h, s), gf[s], dp);
    s_chain(Kn, hn, s_nxt);
    __builtin_amdgcn_sched_barrier(0);
    float pr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], scale_log2e, nlse));
    v8 dsf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float ds = pr[r] * dp[r];
      if (!CFOLD) ds *= scale;
      dsf[r >> 3][r & 7] = (T)ds;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) acc[db] = T16<T>::mfma32(tr_frag<T>(Kt, lo, hstep, h, s2, db), dsf[s2], acc[db]);
  };
  // one 64-key tile.  On entry sA holds the scores of half 0 of tile t; on exit of half 0 of tile t + 1 (behind the last tile:
  // of whatever the next slot holds — never used).
  auto step = [&](int t, const T* Kc, const T* Vc, const T* Kn, T* Kd, T* Vd) {
    if (t + 2 < nt) {
      kd.issue(Kd, wid, t + 2, N);
      vd.issue(Vd, wid, t + 2, N);
    }
    half(Kc, Vc, 0, sA, Kc, 1, sB);
    half(Kc, Vc, 1, sB, Kn, 0, sA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  kd.issue(K0, wid, 0, N);
  vd.issue(V0, wid, 0, N);
  if (nt > 1) {
    kd.issue(K1, wid, 1, N);
    vd.issue(V1, wid, 1, N);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  s_chain(K0, 0, sA);

  for (int t = 0;;) {   // unrolled over the three ring slots: every LDS address is a per-lane base + an immediate
    step(t, K0, V0, K1, K2, V2);
    if (++t == nt) break;
    step(t, K1, V1, K2, K0, V0);
    if (++t == nt) break;
    step(t, K2, V2, K0, K1, V1);
    if (++t == nt) break;
  }

  if (qok) {
    T* op = dq + (row0 + qi) * lddq + head * HD + 4 * fh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2<T>(acc[db][4 * g + 0], acc[db][4 * g + 1]);
        w.y = pack2<T>(acc[db][4 * g + 2], acc[db][4 * g + 3]);
        *reinterpret_cast<uint2*>(op + db * 32 + g * 8) = w;
      }
  }
}

// ---- dK, dV -------------------------------------------------------------------------------------------------------------
// workgroup = 4 waves = 128 keys of one (image, head); streams 64-query tiles of Q and dO (+ their lse2 / -c D rows).
//   S = Q K^T, dP' = dO (c V)^T - c D    (A = Q / dO rows from LDS, B = K / V fragments in registers; the accumulator's
//                                         rows are queries: -c D[q] opens the dP chain, lse2[q] is the fma addend)
//   dV^T += dO^T P, dK^T += Q^T dS'      (A = dO^T / Q^T by transposing reads of the same tiles, B = P / dS' accumulators)
template <typename T, bool CFOLD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_pipe_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                   const T* __restrict__ v, int64_t ld,
                                                                   const T* __restrict__ dO, int64_t lddo,
                                                                   const float* __restrict__ lse2, const float* __restrict__ Dn,
                                                                   T* __restrict__ dk, T* __restrict__ dv, int64_t lddk, int H,
                                                                   int B1, int N1, int N2, float scale, float scale_log2e) {
  typedef typename T16<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T lds[6 * TILE];      // Q ring [3][64][64] | dO ring [3][64][64] = 48 KiB
  __shared__ __attribute__((aligned(16))) float stat[3 * 2 * TT];   // [slot][lse2 | -c D][64]
  T *const Q0 = lds, *const Q1 = lds + TILE, *const Q2 = lds + 2 * TILE;
  T *const G0 = lds + 3 * TILE, *const G1 = lds + 4 * TILE, *const G2 = lds + 5 * TILE;
  float *const S0 = stat, *const S1 = stat + 2 * TT, *const S2 = stat + 4 * TT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int ntl = gridDim.x;  // XCD-aware order: the key tiles of one (image, head) share Q / dO
  const int lin = xcd_remap(blockIdx.x + ntl * (blockIdx.y + gridDim.y * blockIdx.z), ntl * gridDim.y * gridDim.z);
  const int head = (lin / ntl) % gridDim.y, b = lin / (ntl * gridDim.y);
  const int N = b < B1 ? N1 : N2;
  const int64_t row0 = b < B1 ? (int64_t)b * N1 : (int64_t)B1 * N1 + (int64_t)(b - B1) * N2;
  const int64_t st0 = (b < B1 ? (int64_t)b * H * N1 : (int64_t)B1 * H * N1 + (int64_t)(b - B1) * H * N2) + (int64_t)head * N;
  const int key_base = (lin % ntl) * 128 + wid * 32;
  if ((lin % ntl) * 128 >= N) return;   // workgroup-uniform
  const int ki = key_base + fr;
  const bool kok = ki < N;
  const int kc = kok ? ki : N - 1;

  v8 kf[4], vf[4];
  {
    const T* kp = k + (row0 + kc) * ld + head * HD + 8 * fh;
    const T* vp = v + (row0 + kc) * ld + head * HD + 8 * fh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(kp + 16 * s));
      vf[s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(vp + 16 * s));
    }
    if (CFOLD) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) vf[s][j] = (T)((float)vf[s][j] * scale);
    }
  }

  TileDma<T> qd, gd;
  qd.init(q + row0 * ld + head * HD, ld, wid, lane);
  gd.init(dO + row0 * lddo + head * HD, lddo, wid, lane);
  const float* const lsrc = lse2 + st0;
  const float* const dsrc = Dn + st0;
  const int nt = (N + TT - 1) / TT;
  LaneOff lo;
  lo.init(lane);
  const int hstep = ((lane >> 4) & 1) ? -16 : 16;
  // statistics of tile t: wave 0 stages lse2[64 t ..], wave 1 stages -c D[64 t ..] (4 bytes per lane, one DMA each)
  auto stat_issue = [&](float* slot, int t) {
    if (wid < 2) {   // wave-uniform
      const int qq = t * TT + lane;
      const float* src = qq < N ? (wid == 0 ? lsrc : dsrc) + qq : reinterpret_cast<const float*>(g_zero_page_ab);
      glds4(src, slot + wid * TT);
    }
  };

  f32x16 dvacc[2] = {zero16(), zero16()}, dkacc[2] = {zero16(), zero16()};
  f32x16 sA, sB;

  auto s_chain = [&](const T* Qt, int h, f32x16& s_out) {
    f32x16 sa = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) sa = T16<T>::mfma32(row_frag<T>(Qt, lo, h, s), kf[s], sa);
    s_out = sa;
  };
  // One half-step (32 queries).  Accumulator register r of lane half fh is the query 32 h + (r & 7) + 8 fh + 16 (r >> 3) of the
  // tile: -c D of those rows opens the dP chain (four broadcast reads), lse2 of those rows is the fma addend.  No masks:
  // queries >= N are zero rows of the Q and dO tiles (and zero statistics).
  auto half = [&](const T* Qt, const T* Gt, const float* St, int h, const f32x16& sc, const T* Qn, int hn, f32x16& s_nxt) {
    f32x16 dp;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int qo = h * 32 + 8 * fh + 16 * hh;
      const float4 d0 = *reinterpret_cast<const float4*>(St + TT + qo), d1 = *reinterpret_cast<const float4*>(St + TT + qo + 4);
      dp[8 * hh + 0] = d0.x; dp[8 * hh + 1] = d0.y; dp[8 * hh + 2] = d0.z; dp[8 * hh + 3] = d0.w;
      dp[8 * hh + 4] = d1.x; dp[8 * hh + 5] = d1.y; dp[8 * hh + 6] = d1.z; dp[8 * hh + 7] = d1.w;
    }
    if (!CFOLD) {
      const float inv = 1.0f / scale;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] *= inv;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) dp = T16<T>::mfma32(row_frag<T>(Gt, lo, h, s), vf[s], dp);
    s_chain(Qn, hn, s_nxt);
    __builtin_amdgcn_sched_barrier(0);
    float pr[16];
    v8 pf[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int qo = h * 32 + 8 * fh + 16 * hh;
      const float4 l0 = *reinterpret_cast<const float4*>(St + qo), l1 = *reinterpret_cast<const float4*>(St + qo + 4);
      const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * hh + e;
        pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], scale_log2e, -lv[e]));
        pf[hh][e] = (T)pr[r];
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) dvacc[db] = T16<T>::mfma32(tr_frag<T>(Gt, lo, hstep, h, s2, db), pf[s2], dvacc[db]);
    __builtin_amdgcn_sched_barrier(0);
    v8 dsf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float ds = pr[r] * dp[r];
      if (!CFOLD) ds *= scale;
      dsf[r >> 3][r & 7] = (T)ds;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int db = 0; db < 2; ++db) dkacc[db] = T16<T>::mfma32(tr_frag<T>(Qt, lo, hstep, h, s2, db), dsf[s2], dkacc[db]);
  };
  auto step = [&](int t, const T* Qc, const T* Gc, const float* Sc, const T* Qn, T* Qd, T* Gd, float* Sd) {
    if (t + 2 < nt) {
      qd.issue(Qd, wid, t + 2, N);
      gd.issue(Gd, wid, t + 2, N);
      stat_issue(Sd, t + 2);
    }
    half(Qc, Gc, Sc, 0, sA, Qc, 1, sB);
    half(Qc, Gc, Sc, 1, sB, Qn, 0, sA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  qd.issue(Q0, wid, 0, N);
  gd.issue(G0, wid, 0, N);
  stat_issue(S0, 0);
  if (nt > 1) {
    qd.issue(Q1, wid, 1, N);
    gd.issue(G1, wid, 1, N);
    stat_issue(S1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  s_chain(Q0, 0, sA);

  for (int t = 0;;) {
    step(t, Q0, G0, S0, Q1, Q2, G2, S2);
    if (++t == nt) break;
    step(t, Q1, G1, S1, Q2, Q0, G0, S0);
    if (++t == nt) break;
    step(t, Q2, G2, S2, Q0, Q1, G1, S1);
    if (++t == nt) break;
  }

  if (kok) {
    T* kp = dk + (row0 + ki) * lddk + head * HD + 4 * fh;
    T* vp = dv + (row0 + ki) * lddk + head * HD + 4 * fh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2<T>(dkacc[db][4 * g + 0], dkacc[db][4 * g + 1]);
        w.y = pack2<T>(dkacc[db][4 * g + 2], dkacc[db][4 * g + 3]);
        *reinterpret_cast<uint2*>(kp + db * 32 + g * 8) = w;
        w.x = pack2<T>(dvacc[db][4 * g + 0], dvacc[db][4 * g + 1]);
        w.y = pack2<T>(dvacc[db][4 * g + 2], dvacc[db][4 * g + 3]);
        *reinterpret_cast<uint2*>(vp + db * 32 + g * 8) = w;
      }
  }
}

template <typename T>
static void launch_rows(hipStream_t s, const void* q, const void* k, const void* v, int64_t ld, const void* o, int64_t ldo,
                        const void* dO, int64_t lddo, const float* lse2, float* D, void* dq, void* dk, void* dv, int64_t lddq,
                        int B1, int N1, int B2, int N2, int H, float scale, bool cfold) {
  const int B = B1 + B2, N = N1 > N2 ? N1 : N2;
  const float sl = scale * 1.4426950408889634f;
  int64_t nd = (((int64_t)B1 * N1 + (int64_t)B2 * N2) * H * 8 + 255) / 256;
  if (nd > 65535 * 8) nd = 65535 * 8;
  hipLaunchKernelGGL((attn_rowdot_neg_kernel<T>), dim3((unsigned)nd), dim3(256), 0, s, (const T*)o, ldo, (const T*)dO, lddo, D,
                     B1, N1, B2, N2, H, -scale);
  dim3 grid((N + 127) / 128, H, B), block(256);
  if (cfold) {
    hipLaunchKernelGGL((attn_bwd_dq_pipe_kernel<T, true>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld,
                       (const T*)dO, lddo, lse2, D, (T*)dq, lddq, H, B1, N1, N2, scale, sl);
    hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<T, true>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld,
                       (const T*)dO, lddo, lse2, D, (T*)dk, (T*)dv, lddq, H, B1, N1, N2, scale, sl);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_pipe_kernel<T, false>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld,
                       (const T*)dO, lddo, lse2, D, (T*)dq, lddq, H, B1, N1, N2, scale, sl);
    hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<T, false>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld,
                       (const T*)dO, lddo, lse2, D, (T*)dk, (T*)dv, lddq, H, B1, N1, N2, scale, sl);
  }
}

}  // namespace

extern "C" int asis_attention_bwd_rows(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ld,
                                       const void* o, int64_t ldo, const void* dO, int64_t lddo, const float* lse2, float* D,
                                       void* dq, void* dk, void* dv, int64_t lddq, int B1, int N1, int B2, int N2, int H,
                                       float scale) {
  ASIS_REQUIRE(q && k && v && o && dO && lse2 && D && dq && dk && dv, "asis_attention_bwd_rows: null pointer");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_attention_bwd_rows: bad dtype %d", dtype);
  ASIS_REQUIRE(B1 > 0 && B2 >= 0 && H > 0 && N1 > 0 && (B2 == 0 || N2 > 0) && B1 + B2 <= 65535 && H <= 65535,
               "asis_attention_bwd_rows: bad shape B1=%d N1=%d B2=%d N2=%d H=%d", B1, N1, B2, N2, H);
  ASIS_REQUIRE(scale > 0.f, "asis_attention_bwd_rows: scale must be positive");
  const int64_t W = (int64_t)H * HD;
  ASIS_REQUIRE(ld % 8 == 0 && ld >= W && ldo % 8 == 0 && ldo >= W && lddo % 8 == 0 && lddo >= W && lddq % 4 == 0 && lddq >= W,
               "asis_attention_bwd_rows: row strides must be multiples of 8 (dq/dk/dv: 4) and >= H*64");
  ASIS_REQUIRE(asis_aligned16(q) && asis_aligned16(k) && asis_aligned16(v) && asis_aligned16(o) && asis_aligned16(dO),
               "asis_attention_bwd_rows: inputs must be 16-byte aligned");
  ASIS_REQUIRE((((uintptr_t)dq) & 7) == 0 && (((uintptr_t)dk) & 7) == 0 && (((uintptr_t)dv) & 7) == 0 &&
                   (((uintptr_t)lse2) & 3) == 0 && (((uintptr_t)D) & 3) == 0,
               "asis_attention_bwd_rows: outputs must be 8-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int ex;
  const bool cfold = frexpf(scale, &ex) == 0.5f;   // a power of two: c dO / c V are exact in 16 bits (barring subnormals)
  if (dtype == ASIS_F16)
    launch_rows<f16>(s, q, k, v, ld, o, ldo, dO, lddo, lse2, D, dq, dk, dv, lddq, B1, N1, B2, N2, H, scale, cfold);
  else
    launch_rows<bf16>(s, q, k, v, ld, o, ldo, dO, lddo, lse2, D, dq, dk, dv, lddq, B1, N1, B2, N2, H, scale, cfold);
  ASIS_CHECK_LAUNCH("asis_attention_bwd_rows");
  return ASIS_OK;
}
