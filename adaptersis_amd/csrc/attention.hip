// Fused softmax attention forward for the DINOv2 blocks (head dim 64, N = 42*42 (+1 cls) tokens).
//
//   O = softmax(scale * Q K^T) V        dinov2/layers/attention.py:60-66
//
// Flash-style, scores never leave registers.  Layout chosen so that everything per QUERY is
// lane-local on a wave64 with v_mfma_f32_32x32x16:
//   * S^T = K Q^T        (A = K rows from LDS, B = Q^T fragments held in registers)
//       accumulator: column (lane&31) = query, the 16 registers = keys
//       -> row max / row sum are in-lane reductions + one lane^32 exchange.
//   * O^T = V^T P^T      (A = V^T rows from LDS, B = P^T straight from the S^T accumulator,
//       cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand")
//       accumulator: column = query again, so the online-softmax rescale and the final 1/l
//       are lane-local too.
//   The accumulator->operand hand-off permutes k inside each 16-key step; instead of shuffling
//   P we load the K rows of a 32-key block in the inverse permutation (swap bits 2 and 3 of
//   the row index), which makes the matching V^T fragment 8 contiguous keys = one ds_read_b128.
//   V arrives already transposed ([B, H*64, ldvt], keys contiguous) from the QKV GEMM, so no
//   transposed LDS reads are needed.
// Workgroup = 4 waves = 128 queries of one (image, head); K / V^T tiles of 64 keys are
// double-buffered in LDS (32 KiB, XOR-swizzled 128-B rows, conflict-free ds_read_b128), the
// next tile's global loads are in flight during the MFMAs of the current one.
#include <type_traits>

#include "asis_common.h"

namespace {

constexpr int QT = 128;  // queries per workgroup
constexpr int KT = 64;   // keys per tile
constexpr int HD = 64;   // head dim

__device__ __forceinline__ int perm23(int r) {  // swap bits 2 and 3
  return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
}

__device__ __forceinline__ uint64_t clk_after(float dep) {  // s_memtime once `dep` exists (lab timing, ABL == 5)
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}

template <typename T, int OCC, int ABL = 0>   // ABL (lab only, wrong results): 1 = no exp2, 2 = no P.V MFMAs, 3 = no S MFMAs
__global__ __launch_bounds__(256, OCC) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, int64_t ldqk,
                                                       const T* __restrict__ vt, int64_t ldvt, T* __restrict__ o, T* __restrict__ o_lo,
                                                       int64_t ldo, int H, int N1, float scale_log2e,
                                                       float* __restrict__ lse2, int B1, int N2) {
  typedef typename T16<T>::v8 v8;
  __shared__ __attribute__((aligned(16))) T lds[2 * 2 * KT * HD];  // [buf][K | Vt][64][64] = 32 KiB

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  // XCD-aware order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so the q-tiles of one
  // (image, head) would land on 8 different L2s and each would fetch that head's K / V^T for itself (measured: 1.6 GB
  // of L2 fills per launch against 0.35 GB of tensors).  xcd_remap gives every XCD a contiguous run of the logical
  // (q-tile fastest) order instead.
  const int nqt = gridDim.x;
  const int lin = xcd_remap(blockIdx.x + nqt * (blockIdx.y + gridDim.y * blockIdx.z), nqt * gridDim.y * gridDim.z);
  const int qt_idx = lin % nqt;
  const int head = (lin / nqt) % gridDim.y, b = lin / (nqt * gridDim.y);
  // two stacked token batches (images 0..B1-1 with N1 tokens, the rest with N2): row0 = first row of image b
  const int N = b < B1 ? N1 : N2;
  const int64_t row0 = b < B1 ? (int64_t)b * N1 : (int64_t)B1 * N1 + (int64_t)(b - B1) * N2;
  const int q_base = qt_idx * QT + wid * 32;

  // ---- Q^T fragments (B operand of S^T = K Q^T): lane (fr, fh) holds Q[q][16s + 8fh .. +7] ----
  v8 qf[4];
  {
    const int qi = q_base + fr;
    const T* qp = q + (row0 + (qi < N ? qi : 0)) * ldqk + head * HD + 8 * fh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qi < N) v = *reinterpret_cast<const uint4*>(qp + 16 * s);
      qf[s] = __builtin_bit_cast(v8, v);
    }
  }

  // ---- K / V^T tile loaders: 512 16-byte chunks each, 2 per thread ----
  const T* kbase = k + row0 * ldqk + head * HD;
  const T* vbase = vt + ((int64_t)b * H + head) * HD * ldvt;
  // The loads are unconditional (addresses clamped into the tensors) and nothing touches the loaded registers before
  // store_tile: a predicated `v = load` leaves a phi behind, whose register copies put a vmcnt(0) wait right behind
  // the issue and expose the whole global latency on every tile.  K rows >= N repeat row N-1 (their scores are
  // overwritten with -1e30 in the tail tile); V^T columns >= N are zeroed when the tile goes to LDS.  Four named
  // registers, not arrays: the array form ended up in scratch memory (a store right behind each load).
  uint4 rk0, rk1, rv0, rv1;
  const int lrow0 = tid >> 3, lrow1 = lrow0 + 32, lch = tid & 7;
  auto load_tile = [&](int key0) {
    const int ka = key0 + lrow0 < N ? key0 + lrow0 : N - 1;
    const int kb = key0 + lrow1 < N ? key0 + lrow1 : N - 1;
    const int kk = key0 + lch * 8;
    const int kc = kk < N ? kk : 0;
    rk0 = *reinterpret_cast<const uint4*>(kbase + (int64_t)ka * ldqk + lch * 8);
    rk1 = *reinterpret_cast<const uint4*>(kbase + (int64_t)kb * ldqk + lch * 8);
    rv0 = *reinterpret_cast<const uint4*>(vbase + (int64_t)lrow0 * ldvt + kc);
    rv1 = *reinterpret_cast<const uint4*>(vbase + (int64_t)lrow1 * ldvt + kc);
  };
  auto store_tile = [&](int buf, int key0) {
    T* Ks = lds + buf * (2 * KT * HD);
    T* Vs = Ks + KT * HD;
    uint4 w0 = rv0, w1 = rv1;
    if (key0 + KT > N) {  // workgroup-uniform; V^T pad columns may hold anything: force exact zeros
      const int valid = N - (key0 + lch * 8);  // keys of this 8-key chunk that exist (<= 0: none, >= 8: all)
      const uint32_t m0 = valid > 1 ? 0xFFFFFFFFu : (valid > 0 ? 0xFFFFu : 0u);
      const uint32_t m1 = valid > 3 ? 0xFFFFFFFFu : (valid > 2 ? 0xFFFFu : 0u);
      const uint32_t m2 = valid > 5 ? 0xFFFFFFFFu : (valid > 4 ? 0xFFFFu : 0u);
      const uint32_t m3 = valid > 7 ? 0xFFFFFFFFu : (valid > 6 ? 0xFFFFu : 0u);
      w0.x &= m0, w0.y &= m1, w0.z &= m2, w0.w &= m3;
      w1.x &= m0, w1.y &= m1, w1.z &= m2, w1.w &= m3;
    }
    const int sw0 = (lch ^ ((lrow0 >> 1) & 7)) << 3, sw1 = (lch ^ ((lrow1 >> 1) & 7)) << 3;
    *reinterpret_cast<uint4*>(Ks + lrow0 * HD + sw0) = rk0;
    *reinterpret_cast<uint4*>(Vs + lrow0 * HD + sw0) = w0;
    *reinterpret_cast<uint4*>(Ks + lrow1 * HD + sw1) = rk1;
    *reinterpret_cast<uint4*>(Vs + lrow1 * HD + sw1) = w1;
  };

  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (N + KT - 1) / KT;
  uint64_t clk0 = 0, rt0 = 0;
  if (ABL == 5) {
    clk0 = clk_after(0.f);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
  }
  load_tile(0);
  store_tile(0, 0);
  __syncthreads();

  const int prow = perm23(fr);
  // One K/V tile: S^T = K Q^T, online softmax, O^T += V^T P^T.  TAIL masks keys >= N (last tile only, so
  // the full tiles carry no compare/select work).  The running max is only raised when it grows by more than
  // RESCALE_THR (log2 units): P stays <= 2^THR (exact in fp32 sums, same relative precision in fp16) and the
  // 32-register O rescale is skipped on most tiles (cdna_hip_programming.md T13).
  constexpr float RESCALE_THR = 6.0f;
  uint64_t tk[6] = {0, 0, 0, 0, 0, 0}, tk6 = 0, tk7 = 0;
  auto tile = [&](const T* Ks, const T* Vs, int key0, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    if (ABL == 5) tk[0] = clk_after(l_run);
    f32x16 sacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[0][r] = sacc[1][r] = 0.f;
    // the two 32-key blocks alternate so that consecutive MFMAs never accumulate into the same registers (a chain of
    // four dependent MFMAs per block exposes their latency right in front of the softmax, which needs all of S)
    {
      v8 ka[2][4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + prow;
        const int rsw = (row >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          ka[kb][s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Ks + row * HD + (((2 * s + fh) ^ rsw) << 3)));
      }
      if (ABL == 4) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          if (ABL != 3) sacc[kb] = T16<T>::mfma32(ka[kb][s], qf[s], sacc[kb]);
          else sacc[kb][s] += (float)ka[kb][s][0];
        }
      if (ABL == 4) __builtin_amdgcn_s_setprio(0);
    }
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (TAIL) {
          const int key = key0 + kb * 32 + perm23((r & 3) + 8 * (r >> 2) + 4 * fh);
          if (key >= N) sacc[kb][r] = -1e30f;
        }
        mx = fmaxf(mx, sacc[kb][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;  // scale > 0: max commutes with the scaling
    if (ABL == 5) tk[1] = clk_after(mx);
    if (__any(mx > m_run + RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        oacc[0][r] *= alpha;
        oacc[1][r] *= alpha;
      }
    }
    // two scores at a time: the scale-and-shift and the row-sum accumulate as packed fp32 ops (v_pk_fma_f32 /
    // v_pk_add_f32: half the VALU issue slots of the scalar forms; the exp2 itself has no packed form)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 sc2 = {scale_log2e, scale_log2e}, nm2 = {-m_run, -m_run};
    f32x2 ps2 = {0.f, 0.f};
    v8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 s2 = {sacc[kb][r], sacc[kb][r + 1]};
        const f32x2 t = __builtin_elementwise_fma(s2, sc2, nm2);
        f32x2 p2;
        p2.x = ABL == 1 ? t.x : __builtin_amdgcn_exp2f(t.x);
        p2.y = ABL == 1 ? t.y : __builtin_amdgcn_exp2f(t.y);
        ps2 += p2;
        pf[kb][r >> 3][r & 7] = (T)p2.x;
        pf[kb][r >> 3][(r & 7) + 1] = (T)p2.y;
      }
    l_run += ps2.x + ps2.y;
    if (ABL == 5) tk[2] = clk_after(l_run + (float)pf[1][1][7]);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = db * 32 + fr;
      const int rsw = (row >> 1) & 7;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const v8 a = __builtin_bit_cast(
              v8, *reinterpret_cast<const uint4*>(Vs + row * HD + (((4 * kb + 2 * s2 + fh) ^ rsw) << 3)));
          if (ABL != 2) oacc[db] = T16<T>::mfma32(a, pf[kb][s2], oacc[db]);
          else oacc[db][s2] += (float)a[0] * (float)pf[kb][s2][0];
        }
    }
    if (ABL == 5) tk[3] = clk_after(oacc[1][15]);
  };

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int key0 = t * KT;
    if (ABL == 5) tk6 = clk_after(l_run);
    if (t + 1 < nt) load_tile(key0 + KT);
    if (ABL == 5) tk7 = clk_after(l_run);
    const T* Ks = lds + buf * (2 * KT * HD);
    const T* Vs = Ks + KT * HD;
    if (key0 + KT > N) tile(Ks, Vs, key0, std::true_type{});
    else tile(Ks, Vs, key0, std::false_type{});
    if (t + 1 < nt) store_tile(buf ^ 1, key0 + KT);
    if (ABL == 5) tk[4] = clk_after(l_run);
    __syncthreads();
    if (ABL == 5) {
      tk[5] = clk_after(l_run);
      if (lse2 && lane == 0 && wid == 0 && qt_idx == 3 && head == 5 && (b == 0 || b == gridDim.z / 2)) {
        uint64_t* dbg = reinterpret_cast<uint64_t*>(lse2) + ((b ? 1 : 0) * 64 + t) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) dbg[i] = tk[i];
        dbg[6] = tk6;
        dbg[7] = tk7;
      }
    }
  }

  if (ABL == 5 && lse2 && lane == 0 && wid == 0 && qt_idx == 3 && head == 5 && (b == 0 || b == gridDim.z / 2)) {
    uint64_t rt1;
    const uint64_t clk1 = clk_after(l_run);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    uint64_t* dbg = reinterpret_cast<uint64_t*>(lse2) + 2 * 64 * 8 + (b ? 1 : 0) * 2;
    dbg[0] = clk1 - clk0;
    dbg[1] = rt1 - rt0;
  }
  // ---- normalise and store: lane (fr, fh) owns query q_base+fr, d = 32db + 8g + 4fh + (0..3) ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qi = q_base + fr;
  // log2-domain log-sum-exp of the scaled scores, per query: what the backward needs to rebuild P = exp2(s*c - lse2)
  if (ABL != 5 && lse2 && qi < N && fh == 0) lse2[((int64_t)b * H + head) * N1 + qi] = m_run + __builtin_amdgcn_logf(l_tot);  // single batch only
  if (qi < N) {
    T* op = o + (row0 + qi) * ldo + head * HD + 4 * fh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2<T>(oacc[db][4 * g + 0] * inv, oacc[db][4 * g + 1] * inv);
        w.y = pack2<T>(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + db * 32 + g * 8) = w;
      }
    if (o_lo) {  // rounding residual of the 16-bit output: o ~= o + o_lo feeds the projection GEMM as a split operand (A_lo)
      T* lp = o_lo + (row0 + qi) * ldo + head * HD + 4 * fh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = pack2<T>(lo_part<T>(oacc[db][4 * g + 0] * inv), lo_part<T>(oacc[db][4 * g + 1] * inv));
          w.y = pack2<T>(lo_part<T>(oacc[db][4 * g + 2] * inv), lo_part<T>(oacc[db][4 * g + 3] * inv));
          *reinterpret_cast<uint2*>(lp + db * 32 + g * 8) = w;
        }
    }
  }
}


// ---- software-pipelined form --------------------------------------------------------------------------------------
// Same fragments and maths as attn_fwd_kernel, different schedule.  One wave's tile used to be a serial chain
// (K reads -> S MFMAs -> softmax -> V reads -> P.V MFMAs -> staging stores -> barrier) with only the other resident wave
// of the SIMD to overlap with (measured with the ABL == 5 stamps: S 24 %, softmax 21 %, P.V 13 %, staging 19 %).
// Here S of tile t+1 is issued before the softmax of tile t, so the matrix pipe works under the exponentials of the
// same wave, and K / V^T go global -> LDS by LDS-DMA (no staging registers, no ds_write, nothing to wait for until the
// end of the iteration):
//   K ring of 3 tiles, V^T ring of 2 (40 KiB).  Iteration t: DMA K(t+2), V(t+1) | S(t+1) = K(t+1) Q^T |
//   softmax(S(t)) | O += V(t) P(t) | vmcnt(0) + barrier.
// Every buffer a DMA overwrites was last read before the previous barrier; everything read was waited for at it.
// Out-of-range K rows repeat row N-1 (scores masked in the tail tile), out-of-range V^T columns are zeroed in the
// fragment registers of the tail tile (P is exactly 0 there, but 0 * garbage must not make a NaN).
// VROWS (round 3): V comes row-major ([tokens, ld] next to q and k in ONE qkv GEMM output) instead of pre-transposed: the
// tile is staged [key][d] exactly like K and the P.V MFMA's A operand (8 consecutive keys of one d per lane) is assembled by
// the transposing LDS read ds_read_b64_tr_b16 (two per fragment).  That removes the two batched V^T GEMMs per block of the
// stacked trunk (149 us on a side stream against +88 us of the wider qkv GEMM).
template <typename T, int SCHED, int DBG = 0, bool VROWS = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(const T* __restrict__ q, const T* __restrict__ k, int64_t ldqk,
                                                              const T* __restrict__ vt, int64_t ldvt, T* __restrict__ o, T* __restrict__ o_lo,
                                                              int64_t ldo, int H, int N1, float scale_log2e,
                                                              float* __restrict__ lse2, int B1, int N2, int prescaled) {
  typedef typename T16<T>::v8 v8;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  __shared__ __attribute__((aligned(16))) T lds[5 * KT * HD];  // K ring [3][64][64] | V^T ring [2][64][64] = 40 KiB
  // FOLD (SCHED 3): q is pre-scaled by scale * log2(e) and the running maximum enters the score MFMA chain as its initial
  // accumulator (a 16-register block holding -m, rewritten only when m moves), so P = exp2(S') with no per-element fma.
  constexpr bool FOLD = SCHED == 3;
  T* const Kr = lds;
  T* const Vr = lds + 3 * KT * HD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int nqt = gridDim.x;  // XCD-aware order, see attn_fwd_kernel
  const int lin = xcd_remap(blockIdx.x + nqt * (blockIdx.y + gridDim.y * blockIdx.z), nqt * gridDim.y * gridDim.z);
  const int qt_idx = lin % nqt;
  const int head = (lin / nqt) % gridDim.y, b = lin / (nqt * gridDim.y);
  const int N = b < B1 ? N1 : N2;
  const int64_t row0 = b < B1 ? (int64_t)b * N1 : (int64_t)B1 * N1 + (int64_t)(b - B1) * N2;
  const int q_base = qt_idx * QT + wid * 32;

  v8 qf[4];
  {
    const int qi = q_base + fr;
    const T* qp = q + (row0 + (qi < N ? qi : N - 1)) * ldqk + head * HD + 8 * fh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(qp + 16 * s));
    if (FOLD && !prescaled) {  // q' = q * scale * log2(e): the scores leave the MFMA already in exp2 units
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (T)((float)qf[s][j] * scale_log2e);
    }
  }

  const T* kbase = k + row0 * ldqk + head * HD;
  const T* vbase = VROWS ? vt + row0 * ldvt + head * HD : vt + ((int64_t)b * H + head) * HD * ldvt;
  // LDS-DMA: one wave-instruction lands 1 KiB = 8 rows x 8 chunks linearly; lane (lr, lc) therefore fetches the chunk
  // that the XOR swizzle wants at slot lc of row r.  Wave `wid` stages rows 16 wid .. 16 wid + 15 of every tile.
  const int lr = lane >> 3, lc = lane & 7;
  const int r0 = wid * 16 + lr, r1 = r0 + 8;
  const int nt = (N + KT - 1) / KT;
  const int c0 = (lc ^ ((r0 >> 1) & 7)) << 3, c1 = (lc ^ ((r1 >> 1) & 7)) << 3;
  // per-lane source pointers of tile 0; a full tile is a constant step away (rows += 64 for K, columns += 64 for
  // V^T), only the last tile needs the clamped addresses
  const T* const kp0 = kbase + (int64_t)r0 * ldqk + c0;
  const T* const kp1 = kbase + (int64_t)r1 * ldqk + c1;
  // VROWS: rows are keys, 128-byte rows whose 16-byte chunks are XORed with (key & 3) << 1 -- the four key rows of one
  // transposing read then sit in four different 32-byte bank groups
  const int cv0 = VROWS ? (lc ^ ((r0 & 3) << 1)) << 3 : c0, cv1 = VROWS ? (lc ^ ((r1 & 3) << 1)) << 3 : c1;
  const T* const vp0 = vbase + (int64_t)r0 * ldvt + cv0;
  const T* const vp1 = vbase + (int64_t)r1 * ldvt + cv1;
  const int64_t kstep = (int64_t)KT * ldqk;
  auto dma_k = [&](int t, int slot) {
    T* dst = Kr + slot * (KT * HD) + wid * 16 * HD;
    const T *a = kp0 + t * kstep, *bb = kp1 + t * kstep;
    if (t == nt - 1) {  // uniform
      const int key0 = t * KT;
      const int ka = key0 + r0 < N ? key0 + r0 : N - 1;
      const int kb = key0 + r1 < N ? key0 + r1 : N - 1;
      a = kbase + (int64_t)ka * ldqk + c0;
      bb = kbase + (int64_t)kb * ldqk + c1;
    }
    __builtin_amdgcn_global_load_lds((glb_ptr)a, (lds_ptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)bb, (lds_ptr)(dst + 8 * HD), 16, 0, 0);
  };
  auto dma_v = [&](int t) {
    T* dst = Vr + (t & 1) * (KT * HD) + wid * 16 * HD;
    const T *a = VROWS ? vp0 + (int64_t)t * KT * ldvt : vp0 + t * KT, *bb = VROWS ? vp1 + (int64_t)t * KT * ldvt : vp1 + t * KT;
    if (t == nt - 1) {
      const int key0 = t * KT;
      if (VROWS) {  // keys past N repeat row N-1 (finite values; their probabilities are exactly 0)
        const int ka = key0 + r0 < N ? key0 + r0 : N - 1, kb = key0 + r1 < N ? key0 + r1 : N - 1;
        a = vbase + (int64_t)ka * ldvt + cv0;
        bb = vbase + (int64_t)kb * ldvt + cv1;
      } else {
        if (key0 + c0 >= N) a = vbase + (int64_t)r0 * ldvt;
        if (key0 + c1 >= N) bb = vbase + (int64_t)r1 * ldvt;
      }
    }
    __builtin_amdgcn_global_load_lds((glb_ptr)a, (lds_ptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)bb, (lds_ptr)(dst + 8 * HD), 16, 0, 0);
  };

  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = 0.f;
  float m_run = FOLD ? 0.f : -1e30f, l_run = 0.f;
  f32x16 nmb;  // FOLD: -m_run in every element (the C operand that opens each score chain)
#pragma unroll
  for (int r = 0; r < 16; ++r) nmb[r] = 0.f;
  const int prow = perm23(fr);
  constexpr float RESCALE_THR = 6.0f;

  auto k_frags = [&](int slot, v8 (&ka)[2][4]) {
    const T* Ks = Kr + slot * (KT * HD);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = kb * 32 + prow;
      const int rsw = (row >> 1) & 7;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        ka[kb][s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(Ks + row * HD + (((2 * s + fh) ^ rsw) << 3)));
    }
  };
  auto s_mfma = [&](const v8 (&ka)[2][4], f32x16 (&sacc)[2]) {
    if (FOLD) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sacc[kb] = T16<T>::mfma32(ka[kb][s], qf[s], s == 0 ? nmb : sacc[kb]);
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[0][r] = sacc[1][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sacc[kb] = T16<T>::mfma32(ka[kb][s], qf[s], sacc[kb]);
  };

  // one iteration: `cur` holds S(t); `nxt` receives S(t+1)
  auto step = [&](int t, int kslot_next, f32x16 (&cur)[2], f32x16 (&nxt)[2], auto tail_tag) {
    constexpr bool tail = decltype(tail_tag)::value;  // the last tile: masks, no successor
    const int key0 = t * KT;
    const bool more = !tail;
    uint64_t tk[8];
    if (DBG) tk[0] = clk_after(l_run);
    if (t + 2 < nt) dma_k(t + 2, kslot_next == 2 ? 0 : kslot_next + 1);
    if (more) dma_v(t + 1);
    v8 ka[2][4];
    if (DBG) tk[1] = clk_after(l_run);
    if (!tail) k_frags(kslot_next, ka);
    if (tail) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + perm23((r & 3) + 8 * (r >> 2) + 4 * fh);
          if (key >= N) cur[kb][r] = -1e30f;
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, cur[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (!FOLD) mx *= scale_log2e;
    if (DBG) tk[2] = clk_after(mx);
    if (FOLD) {
      // cur holds s - m_run (the offset its chain was opened with); mx is the tile maximum relative to m_run
      if (t == 0 || __any(mx > RESCALE_THR)) {
        const float m_new = t == 0 ? mx : fmaxf(m_run, m_run + mx);
        const float delta = m_run - m_new;
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f(delta);
          l_run *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            oacc[0][r] *= alpha;
            oacc[1][r] *= alpha;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          cur[0][r] += delta;
          cur[1][r] += delta;
          nmb[r] = -m_new;
        }
        m_run = m_new;
      }
    } else if (__any(mx > m_run + RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        oacc[0][r] *= alpha;
        oacc[1][r] *= alpha;
      }
    }
    if (DBG) tk[3] = clk_after(oacc[1][15]);
    // S(t+1) goes to the matrix pipe here, in the same basic block as the exponentials of tile t
    if (!tail) s_mfma(ka, nxt);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    v8 pf[2][2];
    if (FOLD) {
      // (row sums as four more MFMAs per tile, ones[32 x keys] . P, were tried: slower than the 32 adds -- the matrix pipe is
      // the shared resource of the two waves on a SIMD)
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(cur[kb][r]), p1 = __builtin_amdgcn_exp2f(cur[kb][r + 1]);
          ps0 += p0;
          ps1 += p1;
          pf[kb][r >> 3][r & 7] = (T)p0;
          pf[kb][r >> 3][(r & 7) + 1] = (T)p1;
        }
      l_run += ps0 + ps1;
    } else if (SCHED == 2) {
      // single-issue f32 ops only: beside MFMAs a v_pk_fma_f32 / v_pk_add_f32 costs several times two plain ops
      // (MI355X_MICROARCH.md, per-instruction constants); this file is built with -fno-slp-vectorize so they stay unpacked
      const float nm = -m_run;
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float t0 = __builtin_fmaf(cur[kb][r], scale_log2e, nm), t1 = __builtin_fmaf(cur[kb][r + 1], scale_log2e, nm);
          const float p0 = __builtin_amdgcn_exp2f(t0), p1 = __builtin_amdgcn_exp2f(t1);
          ps0 += p0;
          ps1 += p1;
          pf[kb][r >> 3][r & 7] = (T)p0;
          pf[kb][r >> 3][(r & 7) + 1] = (T)p1;
        }
      l_run += ps0 + ps1;
    } else {
    const f32x2 sc2 = {scale_log2e, scale_log2e}, nm2 = {-m_run, -m_run};
    f32x2 ps2 = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 s2 = {cur[kb][r], cur[kb][r + 1]};
        const f32x2 tt = __builtin_elementwise_fma(s2, sc2, nm2);
        f32x2 p2;
        p2.x = __builtin_amdgcn_exp2f(tt.x);
        p2.y = __builtin_amdgcn_exp2f(tt.y);
        ps2 += p2;
        pf[kb][r >> 3][r & 7] = (T)p2.x;
        pf[kb][r >> 3][(r & 7) + 1] = (T)p2.y;
      }
    l_run += ps2.x + ps2.y;
    }
    if (DBG) tk[4] = clk_after(l_run + (float)pf[1][1][7]);
    const T* Vs = Vr + (t & 1) * (KT * HD);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = db * 32 + fr;
      const int rsw = (row >> 1) & 7;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int ch = 4 * kb + 2 * s2 + fh;
          if (VROWS) {
            // transposed read: the 16-lane group reads 4 keys x 16 d; lane li supplies key row li >> 2, d quad li & 3 and
            // receives 4 consecutive keys of d = 16 (group & 1) + li; two reads = the 8 keys 8 ch .. 8 ch + 7
            typedef s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const int li = lane & 15, dcol = db * 32 + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
            s16x4 h2[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int key = 8 * ch + 4 * half + (li >> 2);
              h2[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (lds_tr_ptr)(Vs + key * HD + ((((dcol >> 3) ^ ((key & 3) << 1)) << 3) | (dcol & 7))));
            }
            const v8 vfr = __builtin_bit_cast(v8, (s16x8)__builtin_shufflevector(h2[0], h2[1], 0, 1, 2, 3, 4, 5, 6, 7));
            oacc[db] = T16<T>::mfma32(vfr, pf[kb][s2], oacc[db]);
            continue;
          }
          uint4 w = *reinterpret_cast<const uint4*>(Vs + row * HD + ((ch ^ rsw) << 3));
          if (tail) {
            const int valid = N - (key0 + ch * 8);
            w.x &= valid > 1 ? 0xFFFFFFFFu : (valid > 0 ? 0xFFFFu : 0u);
            w.y &= valid > 3 ? 0xFFFFFFFFu : (valid > 2 ? 0xFFFFu : 0u);
            w.z &= valid > 5 ? 0xFFFFFFFFu : (valid > 4 ? 0xFFFFu : 0u);
            w.w &= valid > 7 ? 0xFFFFFFFFu : (valid > 6 ? 0xFFFFu : 0u);
          }
          oacc[db] = T16<T>::mfma32(__builtin_bit_cast(v8, w), pf[kb][s2], oacc[db]);
        }
    }
    if (SCHED == 1) {
      // 8 S MFMAs spread over the softmax VALU work, then the P.V MFMAs with their reads
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  // VALU
      }
    }
    if (DBG) tk[5] = clk_after(oacc[1][15] + (tail ? 0.f : nxt[1][15]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DBG) tk[6] = clk_after(l_run);
    __syncthreads();
    if (DBG) {
      tk[7] = clk_after(l_run);
      if (lse2 && lane == 0 && wid == 0 && qt_idx == 3 && head == 5 && (b == 0 || b == gridDim.z / 2)) {
        uint64_t* dbg = reinterpret_cast<uint64_t*>(lse2) + ((b ? 1 : 0) * 64 + t) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) dbg[i] = tk[i];
      }
    }
  };

  dma_k(0, 0);
  dma_v(0);
  if (nt > 1) dma_k(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sA[2], sB[2];
  {
    v8 ka[2][4];
    k_frags(0, ka);
    s_mfma(ka, sA);
  }
  int slot = 1;  // ring slot of K(t+1)
  int t = 0;
  for (; t + 2 < nt; t += 2) {  // tiles 0 .. nt-2 are full and have a successor
    step(t, slot, sA, sB, std::false_type{});
    slot = slot == 2 ? 0 : slot + 1;
    step(t + 1, slot, sB, sA, std::false_type{});
    slot = slot == 2 ? 0 : slot + 1;
  }
  if (t + 1 < nt) {
    step(t, slot, sA, sB, std::false_type{});
    step(t + 1, 0, sB, sA, std::true_type{});
  } else {
    step(t, 0, sA, sB, std::true_type{});
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qi = q_base + fr;
  // log2-domain log-sum-exp per query, [B1, H, N1] followed by [B2, H, N2] (the layout asis_attention_bwd_rows reads)
  if (!DBG && lse2 && qi < N && fh == 0) {
    const int64_t st0 = (b < B1 ? (int64_t)b * H * N1 : (int64_t)B1 * H * N1 + (int64_t)(b - B1) * H * N2) + (int64_t)head * N;
    lse2[st0 + qi] = m_run + __builtin_amdgcn_logf(l_tot);
  }
  if (qi < N) {
    T* op = o + (row0 + qi) * ldo + head * HD + 4 * fh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2<T>(oacc[db][4 * g + 0] * inv, oacc[db][4 * g + 1] * inv);
        w.y = pack2<T>(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + db * 32 + g * 8) = w;
      }
    if (o_lo) {  // rounding residual of the 16-bit output: o ~= o + o_lo feeds the projection GEMM as a split operand (A_lo)
      T* lp = o_lo + (row0 + qi) * ldo + head * HD + 4 * fh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = pack2<T>(lo_part<T>(oacc[db][4 * g + 0] * inv), lo_part<T>(oacc[db][4 * g + 1] * inv));
          w.y = pack2<T>(lo_part<T>(oacc[db][4 * g + 2] * inv), lo_part<T>(oacc[db][4 * g + 3] * inv));
          *reinterpret_cast<uint2*>(lp + db * 32 + g * 8) = w;
        }
    }
  }
}

}  // namespace

static int attention_fwd_impl(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                              int64_t ldvt, void* o, void* o_lo, int64_t ldo, int B1, int N1, int B2, int N2, int H,
                              float scale, float* lse2, int prescaled, int vrows = 0) {
  const int B = B1 + B2;
  ASIS_REQUIRE(!o_lo || (((uintptr_t)o_lo) & 7) == 0, "asis_attention_fwd: o_lo must be 8-byte aligned");
  const int N = N1 > N2 ? N1 : N2;
  ASIS_REQUIRE(q && k && vt && o, "asis_attention_fwd: null pointer");
  ASIS_REQUIRE(B1 > 0 && B2 >= 0 && H > 0 && N1 > 0 && (B2 == 0 || N2 > 0), "asis_attention_fwd: bad shape");
  // (the pipelined kernel writes lse2 for both stacked batches; the register-staged lab kernel for one)
  ASIS_REQUIRE(B <= 65535 && H <= 65535, "asis_attention_fwd: B/H too large");
  ASIS_REQUIRE(ldqk % 8 == 0 && ldqk >= (int64_t)H * HD, "asis_attention_fwd: ldqk=%ld must be a multiple of 8 and >= H*64", (long)ldqk);
  ASIS_REQUIRE(ldvt % 8 == 0 && ldvt >= (vrows ? (int64_t)H * HD : (int64_t)N),
               "asis_attention_fwd: ldvt=%ld must be a multiple of 8 and >= %s", (long)ldvt, vrows ? "H*64" : "N");
  ASIS_REQUIRE(ldo % 4 == 0 && ldo >= (int64_t)H * HD, "asis_attention_fwd: ldo=%ld must be a multiple of 4 and >= H*64", (long)ldo);
  ASIS_REQUIRE(asis_aligned16(q) && asis_aligned16(k) && asis_aligned16(vt) && (((uintptr_t)o) & 7) == 0,
               "asis_attention_fwd: pointers must be 16-byte aligned");
  ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, "asis_attention_fwd: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((N + QT - 1) / QT, H, B), block(256);
  const float sl = scale * 1.4426950408889634f;
  // workgroups per CU the kernel is compiled for (register budget 256 / 168 / 128 VGPRs): ASIS_ATTN_OCC = 2 | 3 | 4
  static const int occ = [] { const char* e = getenv("ASIS_ATTN_OCC"); const int v = e ? atoi(e) : 2; return v < 2 ? 2 : (v > 4 ? 4 : v); }();
#define ASIS_ATTN_LAUNCH(TT, O)                                                                                        \
  hipLaunchKernelGGL((attn_fwd_kernel<TT, O>), grid, block, 0, s, reinterpret_cast<const TT*>(q), reinterpret_cast<const TT*>(k), \
                     ldqk, reinterpret_cast<const TT*>(vt), ldvt, reinterpret_cast<TT*>(o), reinterpret_cast<TT*>(o_lo), ldo, H, N1, sl, lse2, B1, N2)
  static const int abl = [] { const char* e = getenv("ASIS_ATTN_ABLATE"); return e ? atoi(e) : 0; }();
  // ASIS_ATTN_PIPE: 0 = the two-buffer register-staged kernel, 1 = software-pipelined LDS-DMA kernel, 2 = the same with
  // an explicit MFMA / VALU interleave
  // 3 = 1 with single-issue f32 softmax arithmetic, 4 = folded form (q pre-scaled in the kernel, -m in the score chain).
  // q already pre-scaled by the caller (`prescaled`): the folded form.
  static const int pipe = [] { const char* e = getenv("ASIS_ATTN_PIPE"); return e ? atoi(e) : 1; }();
  ASIS_REQUIRE(!prescaled || (!abl && (pipe == 1 || pipe == 4)),
               "asis_attention_fwd_prescaled: only the folded kernel takes a pre-scaled q (ASIS_ATTN_PIPE 1 | 4, no ablation)");
  if (vrows) {   // row-major V: the pipelined kernel in its default or folded form
    ASIS_REQUIRE(!abl && (pipe == 1 || pipe == 4), "asis_attention_fwd_qkv: needs the pipelined kernel (ASIS_ATTN_PIPE 1 | 4, no ablation)");
    const bool fold = pipe == 4 || prescaled;
#define ASIS_ATTN_VROWS_LAUNCH(TT, SC)                                                                                          \
  hipLaunchKernelGGL((attn_fwd_pipe_kernel<TT, SC, 0, true>), grid, block, 0, s, reinterpret_cast<const TT*>(q),                \
                     reinterpret_cast<const TT*>(k), ldqk, reinterpret_cast<const TT*>(vt), ldvt, reinterpret_cast<TT*>(o),      \
                     reinterpret_cast<TT*>(o_lo), ldo, H, N1, sl, lse2, B1, N2, prescaled)
    if (dtype == ASIS_F16) { if (fold) ASIS_ATTN_VROWS_LAUNCH(f16, 3); else ASIS_ATTN_VROWS_LAUNCH(f16, 0); }
    else { if (fold) ASIS_ATTN_VROWS_LAUNCH(bf16, 3); else ASIS_ATTN_VROWS_LAUNCH(bf16, 0); }
#undef ASIS_ATTN_VROWS_LAUNCH
    ASIS_CHECK_LAUNCH("asis_attention_fwd_qkv");
    return ASIS_OK;
  }
  if (pipe && abl == 6 && dtype == ASIS_F16) {
    hipLaunchKernelGGL((attn_fwd_pipe_kernel<f16, 0, 1>), grid, block, 0, s, (const f16*)q, (const f16*)k, ldqk, (const f16*)vt, ldvt, (f16*)o, (f16*)o_lo, ldo, H, N1, sl, lse2, B1, N2, 0);
  } else if (pipe && !abl) {
#define ASIS_ATTN_PIPE_LAUNCH(TT, SC)                                                                                   \
  hipLaunchKernelGGL((attn_fwd_pipe_kernel<TT, SC>), grid, block, 0, s, reinterpret_cast<const TT*>(q),                \
                     reinterpret_cast<const TT*>(k), ldqk, reinterpret_cast<const TT*>(vt), ldvt, reinterpret_cast<TT*>(o), \
                     reinterpret_cast<TT*>(o_lo), ldo, H, N1, sl, lse2, B1, N2, prescaled)
    const bool fold = pipe == 4 || prescaled;
    if (dtype == ASIS_F16) { if (fold) ASIS_ATTN_PIPE_LAUNCH(f16, 3); else if (pipe == 2) ASIS_ATTN_PIPE_LAUNCH(f16, 1); else if (pipe == 3) ASIS_ATTN_PIPE_LAUNCH(f16, 2); else ASIS_ATTN_PIPE_LAUNCH(f16, 0); }
    else { if (fold) ASIS_ATTN_PIPE_LAUNCH(bf16, 3); else if (pipe == 2) ASIS_ATTN_PIPE_LAUNCH(bf16, 1); else if (pipe == 3) ASIS_ATTN_PIPE_LAUNCH(bf16, 2); else ASIS_ATTN_PIPE_LAUNCH(bf16, 0); }
#undef ASIS_ATTN_PIPE_LAUNCH
  } else if (abl && dtype == ASIS_F16) {
    ASIS_REQUIRE(!lse2 || B2 == 0, "asis_attention_fwd: the lab kernels write the log-sum-exp of a single token batch");
    if (abl == 1) hipLaunchKernelGGL((attn_fwd_kernel<f16, 2, 1>), grid, block, 0, s, (const f16*)q, (const f16*)k, ldqk, (const f16*)vt, ldvt, (f16*)o, (f16*)o_lo, ldo, H, N1, sl, lse2, B1, N2);
    else if (abl == 2) hipLaunchKernelGGL((attn_fwd_kernel<f16, 2, 2>), grid, block, 0, s, (const f16*)q, (const f16*)k, ldqk, (const f16*)vt, ldvt, (f16*)o, (f16*)o_lo, ldo, H, N1, sl, lse2, B1, N2);
    else if (abl == 5) hipLaunchKernelGGL((attn_fwd_kernel<f16, 2, 5>), grid, block, 0, s, (const f16*)q, (const f16*)k, ldqk, (const f16*)vt, ldvt, (f16*)o, (f16*)o_lo, ldo, H, N1, sl, lse2, B1, N2);
    else if (abl == 4) hipLaunchKernelGGL((attn_fwd_kernel<f16, 2, 4>), grid, block, 0, s, (const f16*)q, (const f16*)k, ldqk, (const f16*)vt, ldvt, (f16*)o, (f16*)o_lo, ldo, H, N1, sl, lse2, B1, N2);
    else hipLaunchKernelGGL((attn_fwd_kernel<f16, 2, 3>), grid, block, 0, s, (const f16*)q, (const f16*)k, ldqk, (const f16*)vt, ldvt, (f16*)o, (f16*)o_lo, ldo, H, N1, sl, lse2, B1, N2);
  } else if (dtype == ASIS_F16) {
    ASIS_REQUIRE(!lse2 || B2 == 0, "asis_attention_fwd: the register-staged kernel writes the log-sum-exp of a single token batch");
    if (occ == 2) ASIS_ATTN_LAUNCH(f16, 2); else if (occ == 3) ASIS_ATTN_LAUNCH(f16, 3); else ASIS_ATTN_LAUNCH(f16, 4);
  } else {
    ASIS_REQUIRE(!lse2 || B2 == 0, "asis_attention_fwd: the register-staged kernel writes the log-sum-exp of a single token batch");
    if (occ == 2) ASIS_ATTN_LAUNCH(bf16, 2); else if (occ == 3) ASIS_ATTN_LAUNCH(bf16, 3); else ASIS_ATTN_LAUNCH(bf16, 4);
  }
#undef ASIS_ATTN_LAUNCH
  ASIS_CHECK_LAUNCH("asis_attention_fwd");
  return ASIS_OK;
}

extern "C" int asis_attention_fwd_split(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                                        int64_t ldvt, void* o, void* o_lo, int64_t ldo, int B1, int N1, int B2, int N2, int H,
                                        float scale, float* lse2) {
  return attention_fwd_impl(stream, dtype, q, k, ldqk, vt, ldvt, o, o_lo, ldo, B1, N1, B2, N2, H, scale, lse2, 0);
}

extern "C" int asis_attention_fwd_prescaled(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                                            int64_t ldvt, void* o, void* o_lo, int64_t ldo, int B1, int N1, int B2, int N2, int H,
                                            float* lse2) {
  return attention_fwd_impl(stream, dtype, q, k, ldqk, vt, ldvt, o, o_lo, ldo, B1, N1, B2, N2, H, 1.0f, lse2, 1);
}

extern "C" int asis_attention_fwd_qkv(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ld, void* o,
                                      void* o_lo, int64_t ldo, int B1, int N1, int B2, int N2, int H, float scale, int prescaled,
                                      float* lse2) {
  ASIS_REQUIRE(v && asis_aligned16(v), "asis_attention_fwd_qkv: v must be a 16-byte aligned pointer");
  return attention_fwd_impl(stream, dtype, q, k, ld, v, ld, o, o_lo, ldo, B1, N1, B2, N2, H, prescaled ? 1.0f : scale, lse2,
                            prescaled, 1);
}

extern "C" int asis_attention_fwd_seg(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                                      int64_t ldvt, void* o, int64_t ldo, int B1, int N1, int B2, int N2, int H, float scale,
                                      float* lse2) {
  return asis_attention_fwd_split(stream, dtype, q, k, ldqk, vt, ldvt, o, nullptr, ldo, B1, N1, B2, N2, H, scale, lse2);
}

extern "C" int asis_attention_fwd_lse(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                                      int64_t ldvt, void* o, int64_t ldo, int B, int H, int N, float scale, float* lse2) {
  return asis_attention_fwd_seg(stream, dtype, q, k, ldqk, vt, ldvt, o, ldo, B, N, 0, 0, H, scale, lse2);
}

extern "C" int asis_attention_fwd(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                                  int64_t ldvt, void* o, int64_t ldo, int B, int H, int N, float scale) {
  return asis_attention_fwd_seg(stream, dtype, q, k, ldqk, vt, ldvt, o, ldo, B, N, 0, 0, H, scale, nullptr);
}
