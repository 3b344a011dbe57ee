// Tile machinery shared by the pipelined attention kernels (attn_fwd_half.hip, attn_bwd_pipe.hip): head dim 64, streamed
// [64 tokens][64 d] 16-bit tiles in LDS, staged by LDS-DMA from an asm statement, read by rows (ds_read_b128) and by columns
// (ds_read_b64_tr_b16) through ONE swizzled image.  gfx950 only.
#pragma once
#include "asis_common.h"

namespace attn_tiles {

constexpr int HD = 64;
constexpr int TT = 64;          // tokens per streamed tile
constexpr int TILE = TT * HD;   // elements per ring slot (8 KiB)

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef s16x4 __attribute__((address_space(3))) * lds_tr_ptr;
typedef short s16x8 __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ int perm23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }
// chunk-slot swizzle of a [64][64] 16-bit tile: slot = chunk ^ fsw(row), fsw = row bits (1, 2, 3) -> slot bits (2, 1, 0)
static __device__ __forceinline__ int fsw(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

// LDS-DMA from an asm statement (cdna_hip_programming.md §5.7): 16 bytes per lane to lds_dst + 16 * lane; M0 is written and
// restored inside the statement.  Not counted by hipcc: completion = our own s_waitcnt vmcnt(0) in front of the barrier.
static __device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
static __device__ __forceinline__ void glds4(const void* gsrc, void* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit byte offset per lane (one address VGPR instead of two)
static __device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, void* lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr)lds_dst);
  const uint64_t b64 = (uint64_t)(uintptr_t)sbase;
  // (readfirstlane returns int: without the unsigned casts a low word with bit 31 set sign-extends into the high word)
  const uint64_t sb = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32)) << 32) |
                      (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b64);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sb) : "memory");
}

static __device__ __attribute__((aligned(16))) uint4 g_zero_page_ab[1];   // source of the rows past the end of a ragged tile

// LDS-DMA of one [64 tokens][64 d] tile: wave `wid` stages rows 16 wid .. 16 wid + 15; one wave-instruction lands 1 KiB =
// 8 rows x 8 chunk slots linearly, so lane (lr, lc) fetches the chunk the swizzle wants at slot lc of its row.  Rows past
// the end of the token range (the ragged last tile) come from a zero page: a zero K / Q / dO row adds exactly nothing to
// the token-reducing products and keeps every intermediate finite, so the consumers carry no masks.
template <typename T>
struct TileDma {
  const T* base;       // row 0 of the (image, head) slice (wave-uniform)
  int64_t ld;
  unsigned off0, off1; // this lane's byte offsets inside tile 0 (row groups r0, r0 + 8)
  int r0, r1;
  __device__ __forceinline__ void init(const T* base_, int64_t ld_, int wid, int lane) {
    base = base_;
    ld = ld_;
    const int lr = lane >> 3, lc = lane & 7;
    r0 = wid * 16 + lr;
    r1 = r0 + 8;
    off0 = (unsigned)((r0 * ld + ((lc ^ fsw(r0)) << 3)) * (int64_t)sizeof(T));
    off1 = (unsigned)((r1 * ld + ((lc ^ fsw(r1)) << 3)) * (int64_t)sizeof(T));
  }
  // tile t (tokens 64 t ..) -> slot
  __device__ __forceinline__ void issue(T* slot, int wid, int t, int N) const {
    T* dst = slot + wid * 16 * HD;
    const T* tb = base + (int64_t)t * TT * ld;
    if ((t + 1) * TT <= N) {  // wave-uniform: a full tile
      glds16_s(tb, off0, dst);
      glds16_s(tb, off1, dst + 8 * HD);
    } else {
      const char* z = reinterpret_cast<const char*>(g_zero_page_ab);
      const char* a = t * TT + r0 < N ? reinterpret_cast<const char*>(tb) + off0 : z;
      const char* b = t * TT + r1 < N ? reinterpret_cast<const char*>(tb) + off1 : z;
      glds16(a, dst);
      glds16(b, dst + 8 * HD);
    }
  }
};

// Per-lane LDS offsets (elements, relative to a tile's first element).  Everything else of an address is a compile-time
// constant of the unrolled tile loop (slot, 32-token half, k-step, d block).
struct LaneOff {
  int row[4];   // row read of k-step s: row perm23(fr), chunk (2 s + fh) ^ f(row)
  int tr[2];    // transposed read of d block db: row 8 (g >> 1) + q, slot of chunk 4 db + 2 (g & 1) + (p >> 1), 4-element half
  __device__ __forceinline__ void init(int lane) {
    const int fr = lane & 31, fh = lane >> 5, prow = perm23(fr);
#pragma unroll
    for (int s = 0; s < 4; ++s) row[s] = prow * HD + (((2 * s + fh) ^ fsw(prow)) << 3);
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // row = 32 h + 16 s2 + 8 (g >> 1) + 4 half + q, chunk = 4 db + 2 (g & 1) + (p >> 1):
    //   f(row) = (q >> 1) << 2 | half << 1 | (g >> 1), so slot = (db ^ (q >> 1)) << 2 | ((g & 1) ^ half) << 1 | ((p >> 1) ^ (g >> 1)):
    //   `half` flips slot bit 1 of a lane-constant -> one base per db, the half's bit is applied with an xor-free +/- below
#pragma unroll
    for (int db = 0; db < 2; ++db)
      tr[db] = (8 * (g >> 1) + q) * HD + ((((db ^ (q >> 1)) << 2) | ((g & 1) << 1) | ((p >> 1) ^ (g >> 1))) << 3) + ((p & 1) << 2);
  }
};

// A fragment of a row-reducing product (S = X K^T style: reduces over d): rows of the 32-token half `h` in the
// bit-2/3-swapped order, 8 consecutive d of k-step s
template <typename T>
__device__ __forceinline__ typename T16<T>::v8 row_frag(const T* tile, const LaneOff& lo, int h, int s) {
  return __builtin_bit_cast(typename T16<T>::v8, *reinterpret_cast<const uint4*>(tile + h * 32 * HD + lo.row[s]));
}

// A fragment of a token-reducing product (X^T . dS): A[d = 32 db + fr][token = 32 h + 16 s2 + 8 fh + j], j = 0..7, by two
// transposing reads of 4 tokens x 16 d each.  Rows 4 half + q: `half` toggles bit 1 of the slot, whose lane part is g & 1 —
// +16 elements where that bit is clear, -16 where it is set, i.e. a per-lane constant sign: tr[db] carries the half = 0
// slot and `hstep` = +-16 moves to the half = 1 slot.
template <typename T>
__device__ __forceinline__ typename T16<T>::v8 tr_frag(const T* tile, const LaneOff& lo, int hstep, int h, int s2, int db) {
  s16x4 hh[2];
  hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(tile + (32 * h + 16 * s2) * HD + lo.tr[db]));
  hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(tile + (32 * h + 16 * s2 + 4) * HD + lo.tr[db] + hstep));
  return __builtin_bit_cast(typename T16<T>::v8, (s16x8)__builtin_shufflevector(hh[0], hh[1], 0, 1, 2, 3, 4, 5, 6, 7));
}

static __device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}


}  // namespace attn_tiles
