// Fused softmax-attention backward for the DINOv2 blocks (head dim 64), the transpose of attention.hip:
//
//   P = softmax(c Q K^T),  O = P V          (c = head_dim^-0.5, dinov2/layers/attention.py:60-66)
//   dV = P^T dO ;  dP = dO V^T ;  dS = c P (dP - D),  D[q] = sum_d dO[q,d] O[q,d] ;  dQ = dS K ;  dK = dS^T Q
//
// Scores are recomputed from Q, K and the forward's per-query log-sum-exp (log2 domain, attention.hip) — nothing of
// size N^2 is stored.  Two kernels, both in the forward's "accumulator column = the stationary index" form so that
// every MFMA operand is either a K-contiguous 16-byte LDS read or the previous product's accumulator handed over as
// the B operand (cdna_hip_programming.md §3), with no atomics and a deterministic result:
//
//   dq  kernel (query stationary, = forward with two more products): per 64-key tile
//        S^T = K Q^T,  dP^T = V dO^T  (A = K / V rows from LDS, B = Q / dO fragments in registers)
//        dS^T = c P^T (dP^T - D[q])   (lse2[q], D[q] are per-lane scalars: the column of the accumulator is the query)
//        dQ^T += K^T dS^T             (A = K^T rows from LDS, B = dS^T from the accumulator)
//   dkv kernel (key stationary): per 64-query tile
//        S = Q K^T,  dP = dO V^T      (A = Q / dO rows from LDS, B = K / V fragments in registers)
//        P = exp2(c' S - lse2[q]),  dS = c P (dP - D[q])   (per-row lse2 / D staged in LDS)
//        dV^T += dO^T P,  dK^T += Q^T dS                    (A = dO^T / Q^T rows from LDS, B = P / dS accumulators)
//
// The transposed operands (K^T for dq; Q^T, dO^T for dkv: [B, H*64, ldt], token index contiguous, zero-padded to
// ldt % 64 == 0) come from asis_transpose_tokens, one streaming pass each, in the layout the forward already uses for
// V^T.  As in the forward, the rows of a 32-row LDS block are read in the bit-2/3-swapped order so that the
// accumulator -> B-operand hand-off needs no shuffle.
#include <type_traits>

#include "asis_common.h"

namespace {

constexpr int HD = 64;
constexpr int TT = 64;  // tokens per streamed tile

__device__ __forceinline__ int perm23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// XOR-swizzled 128-byte rows (conflict-free ds_read_b128), as attention.hip
__device__ __forceinline__ int sw_off(int row, int ch) { return row * HD + ((ch ^ ((row >> 1) & 7)) << 3); }

// ---- [B, N, ld] (columns c0 .. c0+C-1) -> [B, C, ldt] with tokens contiguous, zero-padded to ldt ------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_tokens_kernel(const T* __restrict__ src, int64_t ld, T* __restrict__ dst,
                                                               int64_t ldt, int N, int C) {
  __shared__ T tile[64][64 + 2];
  const int b = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  // load 64 tokens x 64 columns: 512 16-byte chunks
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 3, ch = c & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n0 + row < N) v = *reinterpret_cast<const uint4*>(src + ((int64_t)b * N + n0 + row) * ld + c0 + ch * 8);
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[row][ch * 8 + k] = e[k];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 3, ch = c & 7;  // row = column of src, ch*8 = first token
    T e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = tile[ch * 8 + k][row];
    if (n0 + ch * 8 < ldt)
      *reinterpret_cast<uint4*>(dst + ((int64_t)b * C + c0 + row) * ldt + n0 + ch * 8) = *reinterpret_cast<const uint4*>(e);
  }
}

// ---- D[b, h, q] = sum_d dO[q, h*64 + d] * O[q, h*64 + d] ------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const T* __restrict__ o, int64_t ldo, const T* __restrict__ dO,
                                                          int64_t lddo, float* __restrict__ D, int B, int H, int N) {
  typedef typename T16<T>::v8 v8;
  // 8 lanes per (token, head): 8 x 16 bytes = 64 elements
  const int64_t total = (int64_t)B * N * H * 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int part = (int)(i & 7);
    const int64_t th = i >> 3;
    const int h = (int)(th % H);
    const int64_t tok = th / H;  // b*N + q
    const v8 a = *reinterpret_cast<const v8*>(o + tok * ldo + h * HD + part * 8);
    const v8 g = *reinterpret_cast<const v8*>(dO + tok * lddo + h * HD + part * 8);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (float)a[k] * (float)g[k];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (part == 0) {
      const int b = (int)(tok / N), q = (int)(tok - (int64_t)b * N);
      D[((int64_t)b * H + h) * N + q] = s;
    }
  }
}

// ---- dQ ---------------------------------------------------------------------------------------------------------------
// workgroup = 4 waves = 128 queries of one (image, head); streams 64-key tiles of K, V (rows) and K^T.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const T* __restrict__ v, int64_t ld, const T* __restrict__ kt,
                                                             int64_t ldt, const T* __restrict__ dO, int64_t lddo,
                                                             const float* __restrict__ lse2, const float* __restrict__ Dv,
                                                             T* __restrict__ dq, int64_t lddq, int H, int N, float scale,
                                                             float scale_log2e) {
  typedef typename T16<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T lds[2 * 3 * TT * HD];  // [buf][K | V | K^T][64][64] = 48 KiB
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  // XCD-aware order (see attention.hip): the tiles of one (image, head) share K / V / K^T and stay on one XCD's L2
  const int ntl = gridDim.x;
  const int lin = xcd_remap(blockIdx.x + ntl * (blockIdx.y + gridDim.y * blockIdx.z), ntl * gridDim.y * gridDim.z);
  const int head = (lin / ntl) % gridDim.y, b = lin / (ntl * gridDim.y);
  const int q_base = (lin % ntl) * 128 + wid * 32;
  const int qi = q_base + fr;
  const bool qok = qi < N;

  v8 qf[4], gf[4];
  {
    const int64_t row = (int64_t)b * N + (qok ? qi : 0);
    const T* qp = q + row * ld + head * HD + 8 * fh;
    const T* gp = dO + row * lddo + head * HD + 8 * fh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 a = make_uint4(0, 0, 0, 0), g = a;
      if (qok) {
        a = *reinterpret_cast<const uint4*>(qp + 16 * s);
        g = *reinterpret_cast<const uint4*>(gp + 16 * s);
      }
      qf[s] = __builtin_bit_cast(v8, a);
      gf[s] = __builtin_bit_cast(v8, g);
    }
  }
  const int64_t bh = (int64_t)b * H + head;
  const float lse = qok ? lse2[bh * N + qi] : 0.f;
  const float dsum = qok ? Dv[bh * N + qi] : 0.f;

  const T* kbase = k + (int64_t)b * N * ld + head * HD;
  const T* vbase = v + (int64_t)b * N * ld + head * HD;
  const T* ktbase = kt + bh * HD * ldt;
  // staging in named registers, loads unconditional from clamped rows, rows >= N zeroed when the tile goes to LDS
  // (see attention.hip: the predicated / array forms wait on every load right behind its issue)
  uint4 rk0, rk1, rv0, rv1, rt0, rt1;
  const int lrow0 = tid >> 3, lrow1 = lrow0 + 32, lch = tid & 7;
  auto load_tile = [&](int key0) {
    const int ka = key0 + lrow0 < N ? key0 + lrow0 : N - 1;
    const int kb = key0 + lrow1 < N ? key0 + lrow1 : N - 1;
    rk0 = *reinterpret_cast<const uint4*>(kbase + (int64_t)ka * ld + lch * 8);
    rv0 = *reinterpret_cast<const uint4*>(vbase + (int64_t)ka * ld + lch * 8);
    rk1 = *reinterpret_cast<const uint4*>(kbase + (int64_t)kb * ld + lch * 8);
    rv1 = *reinterpret_cast<const uint4*>(vbase + (int64_t)kb * ld + lch * 8);
    rt0 = *reinterpret_cast<const uint4*>(ktbase + (int64_t)lrow0 * ldt + key0 + lch * 8);  // zero-padded to ldt
    rt1 = *reinterpret_cast<const uint4*>(ktbase + (int64_t)lrow1 * ldt + key0 + lch * 8);
  };
  auto store_tile = [&](int buf, int key0) {
    T* Ks = lds + buf * (3 * TT * HD);
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint4 a0 = rk0, a1 = rk1, w0 = rv0, w1 = rv1;
    if (key0 + TT > N) {  // workgroup-uniform
      if (key0 + lrow0 >= N) a0 = w0 = z;
      if (key0 + lrow1 >= N) a1 = w1 = z;
    }
    const int o0 = sw_off(lrow0, lch), o1 = sw_off(lrow1, lch);
    *reinterpret_cast<uint4*>(Ks + o0) = a0;
    *reinterpret_cast<uint4*>(Ks + TT * HD + o0) = w0;
    *reinterpret_cast<uint4*>(Ks + 2 * TT * HD + o0) = rt0;
    *reinterpret_cast<uint4*>(Ks + o1) = a1;
    *reinterpret_cast<uint4*>(Ks + TT * HD + o1) = w1;
    *reinterpret_cast<uint4*>(Ks + 2 * TT * HD + o1) = rt1;
  };

  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;

  const int nt = (N + TT - 1) / TT;
  load_tile(0);
  store_tile(0, 0);
  __syncthreads();
  const int prow = perm23(fr);

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int key0 = t * TT;
    if (t + 1 < nt) load_tile(key0 + TT);
    const T* Ks = lds + buf * (3 * TT * HD);
    const T* Vs = Ks + TT * HD;
    const T* KTs = Ks + 2 * TT * HD;
    const bool tail = key0 + TT > N;
    v8 dsf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = pacc[r] = 0.f;
      const int row = kb * 32 + prow;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int o = sw_off(row, 2 * s + fh);
        sacc = T16<T>::mfma32(__builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Ks + o)), qf[s], sacc);
        pacc = T16<T>::mfma32(__builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Vs + o)), gf[s], pacc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], scale_log2e, -lse));
        if (tail) {
          const int key = key0 + kb * 32 + perm23((r & 3) + 8 * (r >> 2) + 4 * fh);
          if (key >= N) p = 0.f;
        }
        dsf[kb][r >> 3][r & 7] = (T)(p * (pacc[r] - dsum) * scale);
      }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = db * 32 + fr;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const v8 a = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(KTs + sw_off(row, 4 * kb + 2 * s2 + fh)));
          acc[db] = T16<T>::mfma32(a, dsf[kb][s2], acc[db]);
        }
    }
    if (t + 1 < nt) store_tile(buf ^ 1, key0 + TT);
    __syncthreads();
  }

  if (qok) {
    T* op = dq + ((int64_t)b * N + qi) * lddq + head * HD + 4 * fh;
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
// workgroup = 4 waves = 128 keys of one (image, head); streams 64-query tiles of Q, dO (rows) and Q^T, dO^T.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ v, int64_t ld, const T* __restrict__ qt,
                                                              const T* __restrict__ dot, int64_t ldt,
                                                              const T* __restrict__ dO, int64_t lddo,
                                                              const float* __restrict__ lse2, const float* __restrict__ Dv,
                                                              T* __restrict__ dk, T* __restrict__ dv, int64_t lddk, int H,
                                                              int N, float scale, float scale_log2e) {
  typedef typename T16<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T lds[2 * 4 * TT * HD];  // [buf][Q | dO | Q^T | dO^T][64][64] = 64 KiB
  __shared__ __attribute__((aligned(16))) float stat[2][2][TT];     // [buf][lse2 | D][64]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntl = gridDim.x;  // XCD-aware order: the key tiles of one (image, head) share Q / dO and their transposes
  const int lin = xcd_remap(blockIdx.x + ntl * (blockIdx.y + gridDim.y * blockIdx.z), ntl * gridDim.y * gridDim.z);
  const int head = (lin / ntl) % gridDim.y, b = lin / (ntl * gridDim.y);
  const int key_base = (lin % ntl) * 128 + wid * 32;
  const int ki = key_base + fr;
  const bool kok = ki < N;

  v8 kf[4], vf[4];
  {
    const int64_t row = (int64_t)b * N + (kok ? ki : 0);
    const T* kp = k + row * ld + head * HD + 8 * fh;
    const T* vp = v + row * ld + head * HD + 8 * fh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 a = make_uint4(0, 0, 0, 0), g = a;
      if (kok) {
        a = *reinterpret_cast<const uint4*>(kp + 16 * s);
        g = *reinterpret_cast<const uint4*>(vp + 16 * s);
      }
      kf[s] = __builtin_bit_cast(v8, a);
      vf[s] = __builtin_bit_cast(v8, g);
    }
  }
  const int64_t bh = (int64_t)b * H + head;
  const T* qbase = q + (int64_t)b * N * ld + head * HD;
  const T* gbase = dO + (int64_t)b * N * lddo + head * HD;
  const T* qtbase = qt + bh * HD * ldt;
  const T* gtbase = dot + bh * HD * ldt;
  uint4 rq0, rq1, rg0, rg1, rqt0, rqt1, rgt0, rgt1;
  float rstat = 0.f;
  const int lrow0 = tid >> 3, lrow1 = lrow0 + 32, lch = tid & 7;
  auto load_tile = [&](int q0) {
    const int qa = q0 + lrow0 < N ? q0 + lrow0 : N - 1;
    const int qb = q0 + lrow1 < N ? q0 + lrow1 : N - 1;
    rq0 = *reinterpret_cast<const uint4*>(qbase + (int64_t)qa * ld + lch * 8);
    rg0 = *reinterpret_cast<const uint4*>(gbase + (int64_t)qa * lddo + lch * 8);
    rq1 = *reinterpret_cast<const uint4*>(qbase + (int64_t)qb * ld + lch * 8);
    rg1 = *reinterpret_cast<const uint4*>(gbase + (int64_t)qb * lddo + lch * 8);
    rqt0 = *reinterpret_cast<const uint4*>(qtbase + (int64_t)lrow0 * ldt + q0 + lch * 8);  // zero-padded to ldt
    rgt0 = *reinterpret_cast<const uint4*>(gtbase + (int64_t)lrow0 * ldt + q0 + lch * 8);
    rqt1 = *reinterpret_cast<const uint4*>(qtbase + (int64_t)lrow1 * ldt + q0 + lch * 8);
    rgt1 = *reinterpret_cast<const uint4*>(gtbase + (int64_t)lrow1 * ldt + q0 + lch * 8);
    if (tid < 2 * TT) {  // threads 0..63: lse2, 64..127: D.  Queries >= N: lse2 = +huge -> P = 0
      const int j = tid & (TT - 1);
      const bool isD = tid >= TT;
      const int qq = q0 + j;
      const int qc = qq < N ? qq : N - 1;
      const float v = isD ? Dv[bh * N + qc] : lse2[bh * N + qc];
      rstat = v;
    }
  };
  auto store_tile = [&](int buf, int q0) {
    T* Qs = lds + buf * (4 * TT * HD);
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint4 a0 = rq0, a1 = rq1, g0 = rg0, g1 = rg1;
    if (q0 + TT > N) {  // workgroup-uniform
      if (q0 + lrow0 >= N) a0 = g0 = z;
      if (q0 + lrow1 >= N) a1 = g1 = z;
    }
    const int o0 = sw_off(lrow0, lch), o1 = sw_off(lrow1, lch);
    *reinterpret_cast<uint4*>(Qs + o0) = a0;
    *reinterpret_cast<uint4*>(Qs + TT * HD + o0) = g0;
    *reinterpret_cast<uint4*>(Qs + 2 * TT * HD + o0) = rqt0;
    *reinterpret_cast<uint4*>(Qs + 3 * TT * HD + o0) = rgt0;
    *reinterpret_cast<uint4*>(Qs + o1) = a1;
    *reinterpret_cast<uint4*>(Qs + TT * HD + o1) = g1;
    *reinterpret_cast<uint4*>(Qs + 2 * TT * HD + o1) = rqt1;
    *reinterpret_cast<uint4*>(Qs + 3 * TT * HD + o1) = rgt1;
    if (tid < 2 * TT) {
      const int qq = q0 + (tid & (TT - 1));
      stat[buf][tid >> 6][tid & (TT - 1)] = qq < N ? rstat : (tid >= TT ? 0.f : 1e30f);
    }
  };

  f32x16 dvacc[2], dkacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) dvacc[0][r] = dvacc[1][r] = dkacc[0][r] = dkacc[1][r] = 0.f;

  const int nt = (N + TT - 1) / TT;
  load_tile(0);
  store_tile(0, 0);
  __syncthreads();
  const int prow = perm23(fr);

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load_tile((t + 1) * TT);
    const T* Qs = lds + buf * (4 * TT * HD);
    const T* Gs = Qs + TT * HD;
    const T* QTs = Qs + 2 * TT * HD;
    const T* GTs = Qs + 3 * TT * HD;
    v8 pf[2][2], dsf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = pacc[r] = 0.f;
      const int row = kb * 32 + prow;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int o = sw_off(row, 2 * s + fh);
        sacc = T16<T>::mfma32(__builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Qs + o)), kf[s], sacc);
        pacc = T16<T>::mfma32(__builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Gs + o)), vf[s], pacc);
      }
      // accumulator register r of lane half fh is the query 32 kb + (r & 7) + 8 fh + 16 (r >> 3) of the tile
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int qo = kb * 32 + 8 * fh + 16 * hh;
        const float4 l0 = *reinterpret_cast<const float4*>(&stat[buf][0][qo]), l1 = *reinterpret_cast<const float4*>(&stat[buf][0][qo + 4]);
        const float4 d0 = *reinterpret_cast<const float4*>(&stat[buf][1][qo]), d1 = *reinterpret_cast<const float4*>(&stat[buf][1][qo + 4]);
        const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 8 * hh + e;
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], scale_log2e, -lv[e]));
          pf[kb][hh][e] = (T)p;
          dsf[kb][hh][e] = (T)(p * (pacc[r] - dd[e]) * scale);
        }
      }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = db * 32 + fr;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int o = sw_off(row, 4 * kb + 2 * s2 + fh);
          dvacc[db] = T16<T>::mfma32(__builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(GTs + o)), pf[kb][s2], dvacc[db]);
          dkacc[db] = T16<T>::mfma32(__builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(QTs + o)), dsf[kb][s2], dkacc[db]);
        }
    }
    if (t + 1 < nt) store_tile(buf ^ 1, (t + 1) * TT);
    __syncthreads();
  }

  if (kok) {
    T* kp = dk + ((int64_t)b * N + ki) * lddk + head * HD + 4 * fh;
    T* vp = dv + ((int64_t)b * N + ki) * lddk + head * HD + 4 * fh;
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

}  // namespace

#define DT_OK(dtype, name) ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, name ": bad dtype %d", dtype)

extern "C" int asis_transpose_tokens(void* stream, int dtype, const void* src, int64_t ld, void* dst, int64_t ldt, int B,
                                     int N, int C) {
  ASIS_REQUIRE(src && dst, "asis_transpose_tokens: null pointer");
  DT_OK(dtype, "asis_transpose_tokens");
  ASIS_REQUIRE(B > 0 && N > 0 && C > 0 && C % 64 == 0, "asis_transpose_tokens: C=%d must be a positive multiple of 64", C);
  ASIS_REQUIRE(ld % 8 == 0 && ld >= C, "asis_transpose_tokens: ld=%ld must be a multiple of 8 and >= C", (long)ld);
  ASIS_REQUIRE(ldt % 64 == 0 && ldt >= N, "asis_transpose_tokens: ldt=%ld must be a multiple of 64 and >= N=%d", (long)ldt, N);
  ASIS_REQUIRE(asis_aligned16(src) && asis_aligned16(dst), "asis_transpose_tokens: pointers must be 16-byte aligned");
  ASIS_REQUIRE(B <= 65535 && C / 64 <= 65535, "asis_transpose_tokens: B / C too large");
  dim3 grid((unsigned)(ldt / 64), C / 64, B), block(256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((transpose_tokens_kernel<f16>), grid, block, 0, s, reinterpret_cast<const f16*>(src), ld,
                       reinterpret_cast<f16*>(dst), ldt, N, C);
  else
    hipLaunchKernelGGL((transpose_tokens_kernel<bf16>), grid, block, 0, s, reinterpret_cast<const bf16*>(src), ld,
                       reinterpret_cast<bf16*>(dst), ldt, N, C);
  ASIS_CHECK_LAUNCH("asis_transpose_tokens");
  return ASIS_OK;
}

extern "C" int asis_attention_bwd(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ld,
                                  const void* qt, const void* kt, const void* dot, int64_t ldt, const void* o, int64_t ldo,
                                  const void* dO, int64_t lddo, const float* lse2, float* D, void* dq, void* dk, void* dv,
                                  int64_t lddq, int B, int H, int N, float scale) {
  ASIS_REQUIRE(q && k && v && qt && kt && dot && o && dO && lse2 && D && dq && dk && dv, "asis_attention_bwd: null pointer");
  DT_OK(dtype, "asis_attention_bwd");
  ASIS_REQUIRE(B > 0 && H > 0 && N > 0 && B <= 65535 && H <= 65535, "asis_attention_bwd: bad shape B=%d H=%d N=%d", B, H, N);
  const int64_t W = (int64_t)H * HD;
  ASIS_REQUIRE(ld % 8 == 0 && ld >= W && ldo % 8 == 0 && ldo >= W && lddo % 8 == 0 && lddo >= W && lddq % 4 == 0 && lddq >= W,
               "asis_attention_bwd: row strides must be multiples of 8 (dq/dk/dv: 4) and >= H*64");
  ASIS_REQUIRE(ldt % 64 == 0 && ldt >= N, "asis_attention_bwd: ldt=%ld must be a multiple of 64 and >= N=%d", (long)ldt, N);
  ASIS_REQUIRE(asis_aligned16(q) && asis_aligned16(k) && asis_aligned16(v) && asis_aligned16(qt) && asis_aligned16(kt) &&
                   asis_aligned16(dot) && asis_aligned16(o) && asis_aligned16(dO),
               "asis_attention_bwd: inputs must be 16-byte aligned");
  ASIS_REQUIRE((((uintptr_t)dq) & 7) == 0 && (((uintptr_t)dk) & 7) == 0 && (((uintptr_t)dv) & 7) == 0,
               "asis_attention_bwd: outputs must be 8-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float sl = scale * 1.4426950408889634f;
  dim3 grid((N + 127) / 128, H, B), block(256);
  int64_t nd = ((int64_t)B * N * H * 8 + 255) / 256;
  if (nd > 65535 * 8) nd = 65535 * 8;
  if (dtype == ASIS_F16) {
    typedef f16 T;
    hipLaunchKernelGGL((attn_rowdot_kernel<T>), dim3((unsigned)nd), dim3(256), 0, s, (const T*)o, ldo, (const T*)dO, lddo, D, B, H, N);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (const T*)kt, ldt,
                       (const T*)dO, lddo, lse2, D, (T*)dq, lddq, H, N, scale, sl);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (const T*)qt,
                       (const T*)dot, ldt, (const T*)dO, lddo, lse2, D, (T*)dk, (T*)dv, lddq, H, N, scale, sl);
  } else {
    typedef bf16 T;
    hipLaunchKernelGGL((attn_rowdot_kernel<T>), dim3((unsigned)nd), dim3(256), 0, s, (const T*)o, ldo, (const T*)dO, lddo, D, B, H, N);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (const T*)kt, ldt,
                       (const T*)dO, lddo, lse2, D, (T*)dq, lddq, H, N, scale, sl);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (const T*)qt,
                       (const T*)dot, ldt, (const T*)dO, lddo, lse2, D, (T*)dk, (T*)dv, lddq, H, N, scale, sl);
  }
  ASIS_CHECK_LAUNCH("asis_attention_bwd");
  return ASIS_OK;
}
