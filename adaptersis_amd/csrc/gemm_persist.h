// Persistent dense GEMM with the epilogue of tile i drained under the K loop of tile i+1 (included by gemm.hip).
//
// One workgroup per CU (grid = #CUs), 8 waves of 64x64 (v_mfma_f32_16x16x32), 256x128x32 tiles, FOUR LDS stages filled by
// LDS-DMA three K tiles ahead, fragments of K tile g+1 read from LDS while the MFMAs of K tile g run (register double
// buffer), so the matrix pipe sees back-to-back MFMAs from the SIMD's two waves.  A workgroup walks its tiles as ONE stream
// of K tiles: the DMA of the next output tile's first K tiles is issued under the last K tiles of the current one.
// DEFER: at the end of an output tile the 64 accumulator registers are copied to `old` and the old tile's epilogue is cut
// into 16 pieces, one accumulator fragment each, that ride in the first 16 K iterations of the next tile: bias / GELU /
// LayerScale / residual in registers and one store per lane straight from the fragment (the residual arrives by LDS-DMA
// two pieces ahead).  The last tile of a workgroup has nothing to hide under and goes through the per-wave LDS slab.
//
// STATUS: lab form, off by default (ASIS_GEMM_PERSIST=1 selects it).  Results match the default kernel bit for bit on the
// shapes it covers (tests/test_gpu_kernels.py), but it is NOT faster: on the stacked ViT-L shapes (M = 42348) the main
// loop alone is 4 % ahead of the default's (194 / 97 / 340 us against 203 / 101 / 358 for qkv / proj / fc1), the whole
// kernel 10-30 % behind (259 / 211 / 539 us against 226 / 158 / 450).  The default runs TWO 256-thread workgroups per
// CU, so one workgroup's epilogue already overlaps the other's K loop and each has its own barrier; here all eight waves
// meet at one barrier per K iteration, so whatever a piece costs any wave (16-way fragment select, address arithmetic,
// the store's issue, GELU) is added to the iteration instead of being absorbed.  Lab switches (ASIS_PERSIST_LAB bits)
// put numbers on the parts, qkv shape: no pieces at all 221 us (of which ~20 us is the pipeline drain behind the
// compiler-scheduled bias loads at every tile end), pieces without LDS reads and with their stores sent to an L2-resident
// page 247 us, real stores 259 us; staggering the two waves of a SIMD (pieces before / after the MFMA block) and an
// LDS-transposed variant with full 256-byte row stores (251 / 192 / 510 us) moved nothing.  DESIGN.md section 8 has the
// list of compiler hazards met on the way (inline-asm store data hazard, vmcnt(0) behind LDS-DMA, asm load destinations).
#pragma once
#include <type_traits>

#include "asis_common.h"

namespace {

// Absorbs the always-issued stores of lanes without a valid output element (never read): 1 KB per wave of the persistent
// grid, so that these stores do not all hit one L2 channel (a single shared 1-KB page made every store wait for its turn)
__device__ __attribute__((aligned(16))) uint4 g_trash_page[64 * 8 * 256];

template <typename T, int ACT, bool RES, bool OUT32, bool DEFER, int DBG>
__global__ __launch_bounds__(512, 2) void gemm_persist_kernel(const asis_gemm_desc d, const int GROUP_M) {
  typedef typename T16<T>::v8 v8;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  constexpr int BM2 = 256, BN2 = 128, BKB = 32, NS = 4, G = 3;
  constexpr int STAGE = (BM2 + BN2) * BKB;   // elements per stage (24 KB)
  constexpr int SW = 64 + 4;                 // slab row in floats (conflict-free b128 writes)
  __shared__ __attribute__((aligned(16))) T lds[NS * STAGE];
  __shared__ __attribute__((aligned(16))) float slabs[8 * 16 * SW];   // one 16-row slab per wave (its own object: no alias with the stages)

  const int lab = d.ksplit >= 1000 ? d.ksplit - 1000 : 0;   // lab switches (ASIS_PERSIST_LAB), 0 in production
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int r16 = lane & 15, q16 = lane >> 4;
  const int lr = lane >> 2, lc = lane & 3;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (-(row >> 2)) & 3; };

  // ---- this workgroup's tiles: the XCD's contiguous chunk of the grouped raster order, dealt round-robin to its workgroups
  const int tiles_m = (d.M + BM2 - 1) / BM2, tiles_n = (d.N + BN2 - 1) / BN2;
  const int ntiles = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, P = gridDim.x >> 3;
  const int q0 = (int)(((int64_t)xcd * ntiles) >> 3), q1 = (int)(((int64_t)(xcd + 1) * ntiles) >> 3);
  const int my_n = (q1 - q0 - jx + P - 1) / P > 0 ? (q1 - q0 - jx + P - 1) / P : 0;   // tiles q0 + jx + P*i < q1
  const int nt = d.K / BKB;
  auto tile_origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    const int bid = q0 + jx + P * i;
    const int band = bid / (GROUP_M * tiles_n);
    const int first_m = band * GROUP_M;
    const int band_m = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_band = bid - band * GROUP_M * tiles_n;
    const int tn = in_band / band_m;
    m0 = (first_m + (in_band - tn * band_m)) * BM2;
    n0 = tn * BN2;
  };
  const T* __restrict__ A = reinterpret_cast<const T*>(d.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(d.B);

  // ---- DMA issue stream: (tile ii, K tile ki) runs three K tiles ahead of the compute stream
  uint32_t asrc[2], bsrc;       // per-lane BYTE offsets of the source rows (uniform base + 32-bit offset addressing: the operands are < 4 GB)
  int ii = 0, ki = 0, si = 0;   // issue tile, its K tile, LDS stage
  auto set_src = [&](int i) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(i, m0, n0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wid * 2 + j) * 16 + lr;
      int gr = m0 + row;
      gr = gr < d.M ? gr : d.M - 1;
      asrc[j] = (uint32_t)(((int64_t)gr * d.lda + ((lc ^ swz(row)) << 3)) * sizeof(T));
    }
    const int row = wid * 16 + lr;
    int gr = n0 + row;
    gr = gr < d.N ? gr : d.N - 1;
    bsrc = (uint32_t)(((int64_t)gr * d.ldb + ((lc ^ swz(row)) << 3)) * sizeof(T));
  };
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Bb = reinterpret_cast<const char*>(B);
  auto issue = [&]() __attribute__((always_inline)) {   // exactly G = 3 vector-memory operations, or none once the stream has ended
    if (ii < my_n) {
      T* st = lds + si * STAGE;
      const uint32_t k0 = (uint32_t)(ki * BKB * sizeof(T));
      __builtin_amdgcn_global_load_lds((glb_ptr)(Ab + (asrc[0] + k0)), (lds_ptr)(st + (wid * 2 + 0) * 16 * BKB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)(Ab + (asrc[1] + k0)), (lds_ptr)(st + (wid * 2 + 1) * 16 * BKB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)(Bb + (bsrc + k0)), (lds_ptr)(st + BM2 * BKB + wid * 16 * BKB), 16, 0, 0);
      si = (si + 1) & (NS - 1);
      if (++ki == nt) {
        ki = 0;
        if (++ii < my_n) set_src(ii);
      }
    }
  };

  // ---- fragments
  // Fragment reads are issued as inline asm: the compiler's wait-count pass puts an `s_waitcnt vmcnt(0)` in front of every
  // LDS access it sees behind an LDS-DMA issue (it cannot tell the stage being filled from the stage being read), which
  // would drain the whole DMA pipeline every iteration.  Visibility of the stage is established by the counted vmcnt +
  // barrier at the top of the iteration; completion of these reads by the lgkmcnt(0) behind the MFMA block.
  const uint32_t a_off = (uint32_t)(((wm * 64 + r16) * BKB + ((q16 ^ swz(r16)) << 3)) * sizeof(T));   // rows of a 16-row tile share swz
  const uint32_t b_off = (uint32_t)(((BM2 + wn * 64 + r16) * BKB + ((q16 ^ swz(r16)) << 3)) * sizeof(T));
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) T*)lds;
  auto rd_frags = [&](int stage, v8 (&af)[4], v8 (&bf)[4]) __attribute__((always_inline)) {
    const uint32_t sa = lds_base + (uint32_t)stage * (STAGE * sizeof(T)) + a_off;
    const uint32_t sb = lds_base + (uint32_t)stage * (STAGE * sizeof(T)) + b_off;
    // (straight into the destination registers: a compiler-made copy of an in-flight destination would copy stale data)
#define ASIS_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
    ASIS_DSR(af[0], sa, 0 * 16 * BKB * 2); ASIS_DSR(af[1], sa, 1 * 16 * BKB * 2);
    ASIS_DSR(af[2], sa, 2 * 16 * BKB * 2); ASIS_DSR(af[3], sa, 3 * 16 * BKB * 2);
    ASIS_DSR(bf[0], sb, 0 * 16 * BKB * 2); ASIS_DSR(bf[1], sb, 1 * 16 * BKB * 2);
    ASIS_DSR(bf[2], sb, 2 * 16 * BKB * 2); ASIS_DSR(bf[3], sb, 3 * 16 * BKB * 2);
#undef ASIS_DSR
  };
  auto mma = [&](f32x4 (&acc)[4][4], const v8 (&af)[4], const v8 (&bf)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = T16<T>::mfma16(bf[j], af[i], acc[i][j]);   // D[n][m]: lane = output row
  };
  auto zero = [&](f32x4 (&acc)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- epilogue of one 16-row slab i of the wave's 64x64 tile (rows m0 + wm*64 + 16 i .., columns n0 + wn*64 ..)
  float* slab = slabs + wid * (16 * SW);
  const int rr = lane >> 4, ch = lane & 15;   // read-back: 16 lanes cover one 64-column row (4 columns each), 4 rows per pass
  const float* __restrict__ res = RES ? d.res : nullptr;
  // (inline asm for the same reason as the fragment reads: no compiler-inserted vmcnt(0) inside the K loop)
  const uint32_t slab_w = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)slab + (uint32_t)((r16 * SW + 4 * q16) * 4);
  auto slab_write = [&](const f32x4 (&acc)[4][4], int i) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = acc[i][j];
      if (j == 0) asm volatile("ds_write_b128 %0, %1 offset:0" ::"v"(slab_w), "v"(v) : "memory");
      else if (j == 1) asm volatile("ds_write_b128 %0, %1 offset:64" ::"v"(slab_w), "v"(v) : "memory");
      else if (j == 2) asm volatile("ds_write_b128 %0, %1 offset:128" ::"v"(slab_w), "v"(v) : "memory");
      else asm volatile("ds_write_b128 %0, %1 offset:192" ::"v"(slab_w), "v"(v) : "memory");
    }
  };
  auto res_load = [&](int m0, int n0, int i, int p) __attribute__((always_inline)) -> float4 {
    const int row = m0 + wm * 64 + i * 16 + p * 4 + rr;
    const int rowc = row < d.M ? row : d.M - 1;
    const int col = n0 + wn * 64 + ch * 4;
    const int colc = col < d.N ? col : 0;
    return *reinterpret_cast<const float4*>(res + (int64_t)rowc * d.ldr + colc);
  };
  auto pass_out = [&](int m0, int n0, int i, int p, float4 r4) __attribute__((always_inline)) {
    const int lrow = p * 4 + rr;
    const int row = m0 + wm * 64 + i * 16 + lrow;
    const int col = n0 + wn * 64 + ch * 4;
    float4 v = *reinterpret_cast<const float4*>(slab + lrow * SW + ch * 4);
    if (row < d.M && col < d.N) {
      if (d.bias_n) {
        const float4 b4 = *reinterpret_cast<const float4*>(d.bias_n + col);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      }
      if (ACT == ASIS_ACT_GELU) gelu_erf4(v.x, v.y, v.z, v.w);
      if (d.scale_n) {
        const float4 s4 = *reinterpret_cast<const float4*>(d.scale_n + col);
        v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w;
      }
      if (RES) { v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
      if (OUT32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.C) + (int64_t)row * d.ldc + col) = v;
      } else {
        uint2 pk;
        pk.x = pack2<T>(v.x, v.y);
        pk.y = pack2<T>(v.z, v.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<T*>(d.C) + (int64_t)row * d.ldc + col) = pk;
      }
    }
  };
  auto epilogue_now = [&](const f32x4 (&acc)[4][4], int m0, int n0) __attribute__((always_inline)) {   // the whole epilogue of a tile, not overlapped
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 r4[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) r4[p] = RES ? res_load(m0, n0, i, p) : make_float4(0.f, 0.f, 0.f, 0.f);
      slab_write(acc, i);
#pragma unroll
      for (int p = 0; p < 4; ++p) pass_out(m0, n0, i, p, r4[p]);
    }
  };

  if (my_n == 0) return;
  set_src(0);
  // ---- prologue: three K tiles in flight, the first one landed, its fragments in registers
  issue();
  issue();
  issue();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
  __builtin_amdgcn_s_barrier();
  v8 afA[4], bfA[4], afB[4], bfB[4];
  rd_frags(0, afA, bfA);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f32x4 accA[4][4];
  zero(accA);
  int sc = 1;   // LDS stage of the NEXT K tile to read fragments from

  // ---- deferred epilogue: the old tile's 16 accumulator fragments (i, j) leave one per K iteration, straight from the
  // registers: in the D[n][m] orientation a lane holds 4 consecutive output columns (j*16 + q16*4 ..) of row i*16 + r16, so
  // a fragment is one 8-byte (16-bit output) or 16-byte (fp32) store per lane, 32 / 64 contiguous bytes per row, and the
  // four fragments of a row group complete its 128-byte lines in L2 within four iterations.  No LDS transpose: per piece the
  // LDS sees one broadcast read of the bias (+ LayerScale) columns and, for RES kernels, the residual fragment.
  // Inside the K loop every vector-memory operation of the epilogue is inline asm and ALWAYS issued — one store and (RES)
  // one residual DMA per piece, a lane without a valid output element stores to a trash page — because the counted wait at
  // the top of an iteration is only right if the number of operations behind the awaited DMA is known exactly.
  __shared__ __attribute__((aligned(16))) float bs_lds[8 * 128];
  float* const bsl = bs_lds + wid * 128;
  char* const trash_base = reinterpret_cast<char*>(g_trash_page) + (size_t)((blockIdx.x & 255) * 8 + wid) * 1024;
  char* const Cb = reinterpret_cast<char*>(d.C);
  const char* const Rb = reinterpret_cast<const char*>(res);
  const bool has_scale = d.scale_n != nullptr, has_bias = d.bias_n != nullptr;
  constexpr int ESZ = OUT32 ? 4 : 2;
  int d_row = 0, d_col = 0;        // the lane's row of fragment row group 0 and its first column of fragment column 0
  // Residual fragments travel global -> LDS by LDS-DMA (lane-linear: each lane's own 16 bytes) into two per-wave buffers, two
  // pieces ahead of their use: an asynchronous load into REGISTERS cannot be expressed safely here (the compiler copies an
  // inline-asm destination before the data has landed and then reuses the registers, e.g. as a store address, which the
  // late data overwrites: this faulted).
  __shared__ __attribute__((aligned(16))) float rbuf[8 * 2 * 256];
  float* const rb_wave = rbuf + wid * 512;
  const uint32_t rb_rd = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)rb_wave + (uint32_t)(lane * 16);
  auto ld_res = [&](int q) __attribute__((always_inline)) {   // exactly one vector-memory operation: the residual of piece q
    const int qq = q < 16 ? q : 15;                            // (beyond the tile: a harmless re-read)
    const int row = d_row + 16 * (qq >> 2), col = d_col + 16 * (qq & 3);
    const uint32_t off = (uint32_t)(((int64_t)(row < d.M ? row : d.M - 1) * d.ldr + (col < d.N ? col : 0)) * 4);
    __builtin_amdgcn_global_load_lds((glb_ptr)(Rb + off), (lds_ptr)(rb_wave + (q & 1) * 256), 16, 0, 0);
  };
  auto epi_begin = [&](int m0o, int n0o) __attribute__((always_inline)) {
    d_row = m0o + wm * 64 + r16;
    d_col = n0o + wn * 64 + q16 * 4;
    if (lane < 16 && !(lab & 4)) {   // this wave's 64 bias / scale columns -> LDS (read back per piece: no registers held over the K loop)
      const int col = n0o + wn * 64 + lane * 4;
      const int colc = col < d.N ? col : 0;
      const float4 b = has_bias ? *reinterpret_cast<const float4*>(d.bias_n + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 sc4 = has_scale ? *reinterpret_cast<const float4*>(d.scale_n + colc) : make_float4(1.f, 1.f, 1.f, 1.f);
      *reinterpret_cast<float4*>(bsl + lane * 4) = b;
      *reinterpret_cast<float4*>(bsl + 64 + lane * 4) = sc4;
    }
    if (RES) {
      ld_res(0);
      ld_res(1);
    }
  };
  const uint32_t bs_rd = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)bsl + (uint32_t)(q16 * 16);
  f32x4 old[4][4];                // the previous tile's accumulators while its epilogue pieces ride in this tile's K loop
  // (each case passes its fragment through an empty asm statement: a plain switch over the array is turned into a load from
  // a computed address, which moves the 64 registers of `old` into scratch memory)
  auto old_frag = [&](int q) __attribute__((always_inline)) -> f32x4 {
    f32x4 v;
#define ASIS_FRAG(Q) case Q: asm volatile("" : "=v"(v) : "0"(old[(Q) >> 2][(Q) & 3])); break;
    switch (q) {
      ASIS_FRAG(0) ASIS_FRAG(1) ASIS_FRAG(2) ASIS_FRAG(3) ASIS_FRAG(4) ASIS_FRAG(5) ASIS_FRAG(6) ASIS_FRAG(7)
      ASIS_FRAG(8) ASIS_FRAG(9) ASIS_FRAG(10) ASIS_FRAG(11) ASIS_FRAG(12) ASIS_FRAG(13) ASIS_FRAG(14)
      default: asm volatile("" : "=v"(v) : "0"(old[3][3])); break;
    }
#undef ASIS_FRAG
    return v;
  };
  // piece q (0..15) = fragment (q / 4, q % 4).  LDS reads in one asm block that ends with the wait: the outputs are valid
  // when the statement ends, whatever the compiler does with them afterwards.
  auto piece = [&](int q) __attribute__((always_inline)) {
    const int i = q >> 2, j = q & 3;
    f32x4 b4, s4, r4;
    const uint32_t ba = bs_rd + (uint32_t)(j * 64);
    if (lab & 32) {   // lab: no LDS reads in the piece
      b4 = s4 = r4 = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (RES) {
      const uint32_t ra = rb_rd + (uint32_t)((q & 1) * 1024);
      asm volatile(
          "ds_read_b128 %0, %3\n\t"
          "ds_read_b128 %1, %3 offset:256\n\t"
          "ds_read_b128 %2, %4\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(b4), "=&v"(s4), "=&v"(r4)
          : "v"(ba), "v"(ra)
          : "memory");
    } else {
      asm volatile(
          "ds_read_b128 %0, %2\n\t"
          "ds_read_b128 %1, %2 offset:256\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(b4), "=&v"(s4)
          : "v"(ba)
          : "memory");
      r4 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 v = old_frag(q);
    v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
    if (ACT == ASIS_ACT_GELU) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
    if (has_scale) { v[0] *= s4[0]; v[1] *= s4[1]; v[2] *= s4[2]; v[3] *= s4[3]; }
    if (RES) { v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
    // a lane without a valid element stores to the trash page: base and offset are selected together
    const int row = d_row + 16 * i, col = d_col + 16 * j;
    const bool ok = row < d.M && col < d.N && !(lab & 1);   // lab bit 0: every deferred store goes to the trash page
    const uint32_t off = ok ? (uint32_t)(((int64_t)row * d.ldc + col) * ESZ) : (uint32_t)(lane * 16);
    char* const base = ok ? Cb : trash_base;
    if (OUT32) {
      // (s_nop: a VALU write of a > 8-byte store's data registers needs 2 wait states behind the store on gfx940+; the
      // compiler's hazard recognizer does not look inside inline asm and reused v[0..1] in the very next instruction)
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(base + off), "v"(v) : "memory");
    } else {
      u32x2 pk;
      pk[0] = pack2<T>(v[0], v[1]);
      pk[1] = pack2<T>(v[2], v[3]);
      asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(base + off), "v"(pk) : "memory");
    }
    if (RES) ld_res(q + 2);   // into the buffer just read
  };
  constexpr int NP = 16;   // pieces per tile

  // one K iteration: [K tile g+1 landed] barrier | epilogue piece (early waves) | DMA of g+3 | fragments of g+1 | MFMAs of g
  // | epilogue piece (late waves).  The two waves of a SIMD run their pieces at opposite ends of the iteration, so that
  // one wave's LDS wait and epilogue VALU work sit beside the other wave's MFMAs instead of both stalling in front of them.
  // `last` = no K tile g+1 exists in the whole stream; `tail` = no K tile g+2 (the youngest operations are K tile g+1's own).
  // Counted wait at the top; a piece issues 1 + RES vector-memory operations (a pass: store + residual DMA, a slab piece: as
  // many trash stores), the DMA issue G:
  //  early waves, per iteration [piece, G]: behind K tile g+1 (issued two iterations ago) come [piece of q-1, G]:
  //      q in 1..NP: vmcnt(G + 1 + RES), which also covers the store and the residual DMA of two iterations ago; else vmcnt(G);
  //  late waves, per iteration [G, piece]: behind K tile g+1 come [piece of q-2, G, piece of q-1]; the residual buffer of a
  //      piece was fetched at the end of iteration q-2 or earlier, with [G, piece of q-1] = G + 2 behind it:
  //      q in 2..NP: vmcnt(G + 2) (RES or not), q = 1 or NP + 1: vmcnt(G + 1), else vmcnt(G).
  // (measured: staggering changes nothing, the pieces are not bound by the SIMD's issue slots; default: all waves early)
  const bool late = DEFER && ((lab & 8) ? wid >= 4 : (lab & 16) ? (wid & 1) != 0 : false);
  auto k_iter = [&](int q, f32x4 (&acc)[4][4], v8 (&afc)[4], v8 (&bfc)[4], v8 (&afn)[4], v8 (&bfn)[4], bool last, bool tail)
      __attribute__((always_inline)) {
    if (tail) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (DEFER && !late && q >= 1 && q <= NP) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G + 1 + (RES ? 1 : 0)) : "memory");
    } else if (DEFER && late && q >= 2 && q <= NP) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G + 2) : "memory");
    } else if (DEFER && late && (q == 1 || q == NP + 1)) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G + 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bool has_piece = DEFER && q >= 0 && q < NP && !(lab & 64);   // lab bit 6: no pieces at all (wrong results, timing only)
    if constexpr (DEFER) {
      if (has_piece && !late) piece(q);
    }
    issue();
    if (!last) rd_frags(sc, afn, bfn);
    sc = (sc + 1) & (NS - 1);
    mma(acc, afc, bfc);
    // the MFMAs are queued; by the time the wave gets here the fragment reads issued above have long returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (DEFER) {
      if (has_piece && late) piece(q);
    }
  };

  const int total = my_n * nt;
  int g = 0;
  auto lab_sink = [&](const f32x4 (&acc)[4][4]) __attribute__((always_inline)) {   // lab (DBG & 4): main loop only, keep the accumulators live
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) z += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (z == 123.456f) reinterpret_cast<float*>(d.C)[tid] = z;
  };
  zero(old);
  for (int t = 0; t < my_n; ++t) {
    int m0, n0;
    tile_origin(t, m0, n0);
    const int qoff = (DEFER && t > 0 && !(DBG & 4)) ? 0 : -1000000;   // first tile: no old tile to drain
    // K tiles of this output tile, two per trip so the fragment double buffer keeps static register names (nt is even)
    for (int k = 0; k < nt; k += 2) {
      k_iter(qoff + k, accA, afA, bfA, afB, bfB, g + 1 >= total, g + 2 >= total);
      ++g;
      k_iter(qoff + k + 1, accA, afB, bfB, afA, bfA, g + 1 >= total, g + 2 >= total);
      ++g;
    }
    if (DBG & 4) {
      lab_sink(accA);
    } else if (DEFER && t + 1 < my_n) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) old[i][j] = accA[i][j];
      epi_begin(m0, n0);
    } else {
      epilogue_now(accA, m0, n0);
    }
    zero(accA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace
