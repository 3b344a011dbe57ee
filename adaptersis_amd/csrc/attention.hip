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
// Workgroup = 4 waves = 128 queries of one (image, head); K / V^T tiles of 64 keys are staged by LDS-DMA into rings of three /
// two slots (40 KiB, XOR-swizzled 128-B rows, conflict-free ds_read_b128), software-pipelined: see attn_fwd_pipe_kernel.
// The backward lives in attn_bwd_pipe.hip; asis_transpose_tokens (the V^T image of a row-major V) at the end of this file.
#include <type_traits>

#include "asis_common.h"

namespace {

constexpr int QT = 128;  // queries per workgroup
constexpr int KT = 64;   // keys per tile
constexpr int HD = 64;   // head dim

__device__ __forceinline__ int perm23(int r) {  // swap bits 2 and 3
  return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
}

// ---- the kernel (software-pipelined since round 1; the register-staged form it replaced and its ablation builds left the tree
// in round 5) ------------------------------------------------------------------------------------------------------------
// S of tile t+1 is issued before the softmax of tile t, so the matrix pipe works under the exponentials of the same wave, and
// K / V^T go global -> LDS by LDS-DMA (no staging registers, no ds_write, nothing to wait for until the end of the iteration):
//   K ring of 3 tiles, V^T ring of 2 (40 KiB).  Iteration t: DMA K(t+2), V(t+1) | S(t+1) = K(t+1) Q^T |
//   softmax(S(t)) | O += V(t) P(t) | vmcnt(0) + barrier.
// Every buffer a DMA overwrites was last read before the previous barrier; everything read was waited for at it.
// Out-of-range K rows repeat row N-1 (scores masked in the tail tile), out-of-range V^T columns are zeroed in the
// fragment registers of the tail tile (P is exactly 0 there, but 0 * garbage must not make a NaN).
// SCHED 0: the scale is applied in the softmax (packed f32 multiply-adds); SCHED 3 (FOLD): q arrives pre-scaled, see below.
// VROWS (round 3): V comes row-major ([tokens, ld] next to q and k in ONE qkv GEMM output) instead of pre-transposed: the
// tile is staged [key][d] exactly like K and the P.V MFMA's A operand (8 consecutive keys of one d per lane) is assembled by
// the transposing LDS read ds_read_b64_tr_b16 (two per fragment).  The training forward uses it (both stacked passes and
// their log-sum-exp in one launch); in the frozen trunk the batched V^T GEMMs are hidden on a side stream and stay.
// A half-tile pipelined rebuild on the backward's structure was measured in round 5 and dropped: the kernel is bound by the
// vector issue of the two waves that share a SIMD (profiles/r05_attn_fwd_half_ab.txt).
template <typename T, int SCHED, bool VROWS = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(const T* __restrict__ q, const T* __restrict__ k, int64_t ldqk,
                                                              const T* __restrict__ vt, int64_t ldvt, T* __restrict__ o, T* __restrict__ o_lo,
                                                              int64_t ldo, int H, int N1, float scale_log2e,
                                                              float* __restrict__ lse2, int B1, int N2, int prescaled,
                                                              const float* __restrict__ mx_amax) {
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
    if (t + 2 < nt) dma_k(t + 2, kslot_next == 2 ? 0 : kslot_next + 1);
    if (more) dma_v(t + 1);
    v8 ka[2][4];
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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
  if (lse2 && qi < N && fh == 0) {
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
    if (o_lo) {  // rounding residual of the 16-bit output: o ~= o + o_lo feeds the projection GEMM as a split operand (A_lo);
                 // with mx_amax (an upper bound of |o|: o is a convex combination of V rows, so max |V| is one) in the MX form
      T* lp = o_lo + (row0 + qi) * ldo + head * HD + 4 * fh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = lo_word2<T>(oacc[db][4 * g + 0] * inv, oacc[db][4 * g + 1] * inv, mx_amax);
          w.y = lo_word2<T>(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv, mx_amax);
          *reinterpret_cast<uint2*>(lp + db * 32 + g * 8) = w;
        }
    }
  }
}

}  // namespace

static int attention_fwd_impl(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                              int64_t ldvt, void* o, void* o_lo, int64_t ldo, int B1, int N1, int B2, int N2, int H,
                              float scale, float* lse2, int prescaled, int vrows = 0, const float* mx_amax = nullptr) {
  const int B = B1 + B2;
  ASIS_REQUIRE(!o_lo || (((uintptr_t)o_lo) & 7) == 0, "asis_attention_fwd: o_lo must be 8-byte aligned");
  const int N = N1 > N2 ? N1 : N2;
  ASIS_REQUIRE(q && k && vt && o, "asis_attention_fwd: null pointer");
  ASIS_REQUIRE(B1 > 0 && B2 >= 0 && H > 0 && N1 > 0 && (B2 == 0 || N2 > 0), "asis_attention_fwd: bad shape");
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
  // one kernel, four forms: a pre-scaled q (the folded form: -m rides in the score chain) or the scale applied in the softmax;
  // V pre-transposed or row-major (transposing LDS reads)
  const bool fold = prescaled != 0;
#define ASIS_ATTN_PIPE_LAUNCH(TT, SC, VR)                                                                                 \
  hipLaunchKernelGGL((attn_fwd_pipe_kernel<TT, SC, VR>), grid, block, 0, s, reinterpret_cast<const TT*>(q),              \
                     reinterpret_cast<const TT*>(k), ldqk, reinterpret_cast<const TT*>(vt), ldvt, reinterpret_cast<TT*>(o), \
                     reinterpret_cast<TT*>(o_lo), ldo, H, N1, sl, lse2, B1, N2, prescaled, mx_amax)
  if (dtype == ASIS_F16) {
    if (vrows) { if (fold) ASIS_ATTN_PIPE_LAUNCH(f16, 3, true); else ASIS_ATTN_PIPE_LAUNCH(f16, 0, true); }
    else { if (fold) ASIS_ATTN_PIPE_LAUNCH(f16, 3, false); else ASIS_ATTN_PIPE_LAUNCH(f16, 0, false); }
  } else {
    if (vrows) { if (fold) ASIS_ATTN_PIPE_LAUNCH(bf16, 3, true); else ASIS_ATTN_PIPE_LAUNCH(bf16, 0, true); }
    else { if (fold) ASIS_ATTN_PIPE_LAUNCH(bf16, 3, false); else ASIS_ATTN_PIPE_LAUNCH(bf16, 0, false); }
  }
#undef ASIS_ATTN_PIPE_LAUNCH
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

extern "C" int asis_attention_fwd_qkv_mx(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ld, void* o,
                                         void* o_mx, int64_t ldo, int B1, int N1, int B2, int N2, int H, float scale, int prescaled,
                                         float* lse2, const float* amax) {
  ASIS_REQUIRE(v && asis_aligned16(v) && o_mx && amax, "asis_attention_fwd_qkv_mx: v (16-byte aligned), o_mx and amax are required");
  return attention_fwd_impl(stream, dtype, q, k, ld, v, ld, o, o_mx, ldo, B1, N1, B2, N2, H, prescaled ? 1.0f : scale, lse2,
                            prescaled, 1, amax);
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

// ==== token transpose: the V^T operand of the forward from a row-major V ====================================================
namespace {

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

