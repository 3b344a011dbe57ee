// Attention forward, one wave per SIMD: included by attention.hip inside its anonymous namespace (uses KT, HD, perm23,
// xcd_remap, T16, pack2, lo_part).  `dinov2/layers/attention.py:60-66` (MemEffAttention without bias / dropout).
//
// Why another form.  In attn_fwd_pipe_kernel two 4-wave workgroups share a CU, i.e. two waves share every SIMD and with it
// ONE matrix pipe and one vector issue port (MI355X_MICROARCH.md, "Two waves per SIMD"): a partner's MFMAs come straight out of
// a wave's own stream, and the counters of that kernel say so — per wave and 64-key tile 2 578 cycles of which 39 % issue,
// 38 % issue-stall, 24 % parked, matrix pipe 40 % busy.  Here a workgroup is 4 waves x 64 query rows (two 32-row blocks per
// wave) and owns the CU's whole register file (launch_bounds(256, 1): 512 registers per lane, accumulators that only the
// matrix pipe touches live in the accumulation half): every K / V^T fragment read from LDS feeds two MFMAs instead of one,
// there is one barrier per 256 x 64 tile instead of per 128 x 64, and the overlap of matrix and vector work comes from the
// wave's own software pipeline (S(t+1) issued before the exponentials of tile t), not from a partner.
//
// Arithmetic differences to the pipe kernel (both inside the 1e-3 contract, tests/test_gpu_vit.py):
//   * q is multiplied by scale * log2(e) once per workgroup (16-bit re-rounding of q) unless the caller did it in the
//     projection weights (`prescaled`), so the scores leave the MFMA in exp2 units;
//   * the running maximum enters the score chain as its initial accumulator (a 16-register block of -m per query block,
//     rewritten only when m moves by more than RESCALE_THR), so P = exp2(S') needs no per-element multiply-add.
template <typename T>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(const T* __restrict__ q, const T* __restrict__ k, int64_t ldqk,
                                                             const T* __restrict__ vt, int64_t ldvt, T* __restrict__ o,
                                                             T* __restrict__ o_lo, int64_t ldo, int H, int N1, float scale_log2e,
                                                             float* __restrict__ lse2, int B1, int N2, int prescaled) {
  typedef typename T16<T>::v8 v8;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  constexpr int QW = 256;  // queries per workgroup
  __shared__ __attribute__((aligned(16))) T lds[5 * KT * HD];  // K ring [3][64][64] | V^T ring [2][64][64] = 40 KiB
  T* const Kr = lds;
  T* const Vr = lds + 3 * KT * HD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int nqt = gridDim.x;
  const int lin = xcd_remap(blockIdx.x + nqt * (blockIdx.y + gridDim.y * blockIdx.z), nqt * gridDim.y * gridDim.z);
  const int qt_idx = lin % nqt;
  const int head = (lin / nqt) % gridDim.y, b = lin / (nqt * gridDim.y);
  const int N = b < B1 ? N1 : N2;
  const int64_t row0 = b < B1 ? (int64_t)b * N1 : (int64_t)B1 * N1 + (int64_t)(b - B1) * N2;
  const int q_base = qt_idx * QW + wid * 64;

  v8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = q_base + qb * 32 + fr;
    const T* qp = q + (row0 + (qi < N ? qi : N - 1)) * ldqk + head * HD + 8 * fh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s] = __builtin_bit_cast(v8, *reinterpret_cast<const uint4*>(qp + 16 * s));
    if (!prescaled) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qb][s][j] = (T)((float)qf[qb][s][j] * scale_log2e);
    }
  }

  // the Q fragments are MFMA sources only: park them in the accumulation half (sources may come from either half)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+a"(qf[qb][s]));

  const T* kbase = k + row0 * ldqk + head * HD;
  const T* vbase = vt + ((int64_t)b * H + head) * HD * ldvt;
  // LDS-DMA staging exactly as in attn_fwd_pipe_kernel: wave `wid` lands rows 16 wid .. 16 wid + 15 of every tile
  const int lr = lane >> 3, lc = lane & 7;
  const int r0 = wid * 16 + lr, r1 = r0 + 8;
  const int nt = (N + KT - 1) / KT;
  const int c0 = (lc ^ ((r0 >> 1) & 7)) << 3, c1 = (lc ^ ((r1 >> 1) & 7)) << 3;
  const T* const kp0 = kbase + (int64_t)r0 * ldqk + c0;
  const T* const kp1 = kbase + (int64_t)r1 * ldqk + c1;
  const T* const vp0 = vbase + (int64_t)r0 * ldvt + c0;
  const T* const vp1 = vbase + (int64_t)r1 * ldvt + c1;
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
    const T *a = vp0 + t * KT, *bb = vp1 + t * KT;
    if (t == nt - 1) {
      const int key0 = t * KT;
      if (key0 + c0 >= N) a = vbase + (int64_t)r0 * ldvt;
      if (key0 + c1 >= N) bb = vbase + (int64_t)r1 * ldvt;
    }
    __builtin_amdgcn_global_load_lds((glb_ptr)a, (lds_ptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)bb, (lds_ptr)(dst + 8 * HD), 16, 0, 0);
  };

  f32x16 oacc[2][2];  // [query block][d block]
  f32x16 nmb[2];      // -m of the query block in every element: opens each score chain
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    oacc[0][0][r] = oacc[0][1][r] = oacc[1][0][r] = oacc[1][1][r] = 0.f;
    nmb[0][r] = nmb[1][r] = 0.f;
  }
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
  auto s_mfma = [&](const v8 (&ka)[2][4], f32x16 (&sacc)[2][2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) sacc[qb][kb] = T16<T>::mfma32(ka[kb][s], qf[qb][s], s == 0 ? nmb[qb] : sacc[qb][kb]);
  };

  // O^T += V^T P with the accumulator pinned to the accumulation half.  The file is built with -amdgpu-mfma-vgpr-form, so the
  // builtin MFMAs (the scores, read by the vector unit) produce architectural registers; C and D of one MFMA share a class,
  // hence this one is written out.  Its sources come from ds_read (the compiler waits for asm inputs) and v_cvt_pk (no
  // VALU -> MFMA hazard on gfx950); dependent MFMAs on one accumulator chain in hardware.
  auto pv_mfma = [&](f32x16& acc, v8 a, v8 bfrag) {
    if (sizeof(T) == 2 && __is_same(T, f16))
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(bfrag));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(bfrag));
  };

  // one iteration: `cur` holds S'(t) = S(t) - m (per query block), `nxt` receives S'(t+1)
  auto step = [&](int t, int kslot_next, f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], auto tail_tag) {
    constexpr bool tail = decltype(tail_tag)::value;
    const int key0 = t * KT;
    // (the compiler drains the DMA counter in front of the first V^T read further down: by then the transfers issued here
    // have had the whole score + exponential phase to land)
    if (t + 2 < nt) dma_k(t + 2, kslot_next == 2 ? 0 : kslot_next + 1);
    if (!tail) dma_v(t + 1);
    v8 ka[2][4];
    if (!tail) k_frags(kslot_next, ka);
    if (tail) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + perm23((r & 3) + 8 * (r >> 2) + 4 * fh);
          if (key >= N) cur[0][kb][r] = cur[1][kb][r] = -1e30f;
        }
    }
    float mx[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float m = -1e30f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, cur[qb][kb][r]);
      mx[qb] = fmaxf(m, __shfl_xor(m, 32, 64));
    }
    // cur is relative to m_run (the offset its chain was opened with): move m when a tile maximum exceeds it by the threshold
    if (t == 0 || __any(fmaxf(mx[0], mx[1]) > RESCALE_THR)) {
      // Rare path.  The output accumulators live in the accumulation half and are written by the asm MFMAs below, which the
      // compiler's hazard recogniser cannot see: the padded statement orders their results before the reads (data dependence
      // through its operands), the second one orders the rewritten values before the next MFMA.
      asm volatile("s_nop 15\n\ts_nop 15" : "+a"(oacc[0][0]), "+a"(oacc[0][1]), "+a"(oacc[1][0]), "+a"(oacc[1][1]));
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float m_new = t == 0 ? mx[qb] : fmaxf(m_run[qb], m_run[qb] + mx[qb]);
        const float delta = m_run[qb] - m_new;
        const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(delta);
        l_run[qb] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          oacc[qb][0][r] *= alpha;
          oacc[qb][1][r] *= alpha;
          cur[qb][0][r] += delta;
          cur[qb][1][r] += delta;
          nmb[qb][r] = -m_new;
        }
        m_run[qb] = m_new;
      }
      asm volatile("s_nop 7" : "+a"(oacc[0][0]), "+a"(oacc[0][1]), "+a"(oacc[1][0]), "+a"(oacc[1][1]));
    }
    if (!tail) s_mfma(ka, nxt);  // S'(t+1) goes to the matrix pipe under the exponentials of tile t
    v8 pf[2][2][2];              // [query block][key block][k step]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(cur[qb][kb][r]), p1 = __builtin_amdgcn_exp2f(cur[qb][kb][r + 1]);
          ps0 += p0;
          ps1 += p1;
          pf[qb][kb][r >> 3][r & 7] = (T)p0;
          pf[qb][kb][r >> 3][(r & 7) + 1] = (T)p1;
        }
      l_run[qb] += ps0 + ps1;
    }
    __builtin_amdgcn_sched_barrier(0);  // the exponentials, their sums and conversions end here (register lifetimes)
    const T* Vs = Vr + (t & 1) * (KT * HD);
    uint4 vf[2][2][2];  // all eight V^T fragments first: the asm MFMAs below are scheduling fences for the reads
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = db * 32 + fr;
      const int rsw = (row >> 1) & 7;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int ch = 4 * kb + 2 * s2 + fh;
          uint4 w = *reinterpret_cast<const uint4*>(Vs + row * HD + ((ch ^ rsw) << 3));
          if (tail) {
            const int valid = N - (key0 + ch * 8);
            w.x &= valid > 1 ? 0xFFFFFFFFu : (valid > 0 ? 0xFFFFu : 0u);
            w.y &= valid > 3 ? 0xFFFFFFFFu : (valid > 2 ? 0xFFFFu : 0u);
            w.z &= valid > 5 ? 0xFFFFFFFFu : (valid > 4 ? 0xFFFFu : 0u);
            w.w &= valid > 7 ? 0xFFFFFFFFu : (valid > 6 ? 0xFFFFu : 0u);
          }
          vf[db][kb][s2] = w;
        }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) pv_mfma(oacc[qb][db], __builtin_bit_cast(v8, vf[db][kb][s2]), pf[qb][kb][s2]);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  dma_k(0, 0);
  dma_v(0);
  if (nt > 1) dma_k(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sA[2][2], sB[2][2];
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

  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(oacc[0][0]), "+a"(oacc[0][1]), "+a"(oacc[1][0]), "+a"(oacc[1][1]));  // MFMA -> read
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = 1.0f / l_tot;
    const int qi = q_base + qb * 32 + fr;
    if (lse2 && qi < N && fh == 0) lse2[((int64_t)b * H + head) * N1 + qi] = m_run[qb] + __builtin_amdgcn_logf(l_tot);
    if (qi < N) {
      T* op = o + (row0 + qi) * ldo + head * HD + 4 * fh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = pack2<T>(oacc[qb][db][4 * g + 0] * inv, oacc[qb][db][4 * g + 1] * inv);
          w.y = pack2<T>(oacc[qb][db][4 * g + 2] * inv, oacc[qb][db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + db * 32 + g * 8) = w;
        }
      if (o_lo) {  // rounding residual of the 16-bit output (config.split_attn_out)
        T* lp = o_lo + (row0 + qi) * ldo + head * HD + 4 * fh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack2<T>(lo_part<T>(oacc[qb][db][4 * g + 0] * inv), lo_part<T>(oacc[qb][db][4 * g + 1] * inv));
            w.y = pack2<T>(lo_part<T>(oacc[qb][db][4 * g + 2] * inv), lo_part<T>(oacc[qb][db][4 * g + 3] * inv));
            *reinterpret_cast<uint2*>(lp + db * 32 + g * 8) = w;
          }
      }
    }
  }
}
