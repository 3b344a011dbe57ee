// Backward kernels of the adapter modules (`train_adapters` mode: the gradients the reference's optimiser lists but
// its graph cut never produces — SURVEY.md fact 1 — and that its MSDeformAttnFunction cannot compute — fact 2):
//
//   msda_bwd        transpose of msda_fwd (msda.hip; ms_deform_attn.py:33-54,155-166 under autograd):
//                     out[q,m,:] = sum_j A_j S_j,  A = softmax_j(logit),  S_j = bilinear(value_l(j)[.., m, :], pix_j),
//                     pix_j = ref*(W,H) + off_j - 0.5
//                   d value  += A_j * bw_t * d out        (scatter over the 4 taps: fp32 global atomics — the one
//                                                          place in this library whose summation order is not fixed)
//                   d off_j   = A_j * <d out, dS_j/d pix>  (zero-padded taps contribute nothing)
//                   d logit_j = A_j * (<d out, S_j> - sum_k A_k <d out, S_k>)
//   dwconv_gelu_bwd transpose of dwconv_gelu (adapter_blocks.py:67-80,95-97): g = d y * gelu'(dwconv(x) + b) with the
//                   pre-activation recomputed, per-block partial sums of d bias and d w[9]; dwconv_t then correlates g
//                   with the taps (depthwise transposed conv) into the 16-bit operand of fc1's backward.
#include "asis_common.h"

namespace {

constexpr int MAX_LP = 16;
constexpr int MAX_M = 32;

template <typename T, bool DVALUE>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const T* __restrict__ value, const float* __restrict__ offaw,
                                                       int64_t ld_offaw, const float* __restrict__ ref,
                                                       const int* __restrict__ shapes, const int* __restrict__ starts,
                                                       const float* __restrict__ dout, float* __restrict__ dvalue,
                                                       float* __restrict__ doffaw, int B, int Lq, int Lin, int M, int L,
                                                       int P, int Dh) {
  __shared__ float red[MAX_M * MAX_LP * 3];  // [m][j][{<dout,S>, <dout,dS/dx>, <dout,dS/dy>}]
  const int D = M * Dh;
  const int cpq = D >> 3, cph = Dh >> 3;
  const int LP = L * P;
  const int c = threadIdx.x;
  const bool live = c < cpq;
  const int m = live ? c / cph : 0;
  const bool pow2 = (cph & (cph - 1)) == 0 && cph <= 64 && (cpq & 63) == 0;   // whole waves of aligned head groups
  for (int64_t bq = blockIdx.x; bq < (int64_t)B * Lq; bq += gridDim.x) {
    const int q = (int)(bq % Lq);
    const int b = (int)(bq / Lq);
    const float* orow = offaw + bq * ld_offaw;
    for (int i = threadIdx.x; i < M * LP * 3; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    if (live) {
      const float* lg = orow + (int64_t)M * LP * 2 + m * LP;
      float w[MAX_LP];
      float mx = -1e30f;
#pragma unroll
      for (int j = 0; j < MAX_LP; ++j)
        if (j < LP) {
          w[j] = lg[j];
          mx = fmaxf(mx, w[j]);
        }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < MAX_LP; ++j)
        if (j < LP) {
          w[j] = __expf(w[j] - mx);
          den += w[j];
        }
      const float inv = 1.0f / den;
      const float rx = ref[2 * q], ry = ref[2 * q + 1];
      float g[8];
      {
        const float4 g0 = *reinterpret_cast<const float4*>(dout + bq * D + c * 8);
        const float4 g1 = *reinterpret_cast<const float4*>(dout + bq * D + c * 8 + 4);
        g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
      }
      const T* vb = value + (int64_t)b * Lin * D + c * 8;
      float* dvb = dvalue + (int64_t)b * Lin * D + c * 8;
      for (int l = 0; l < L; ++l) {
        const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
        const T* vl = vb + (int64_t)starts[l] * D;
        float* dvl = dvb + (int64_t)starts[l] * D;
#pragma unroll 4
        for (int p = 0; p < P; ++p) {
          const int j = l * P + p;
          const float ox = orow[(m * LP + j) * 2], oy = orow[(m * LP + j) * 2 + 1];
          const float lx = rx + ox / (float)Wl, ly = ry + oy / (float)Hl;   // as msda_fwd (same taps chosen)
          const float px = lx * (float)Wl - 0.5f, py = ly * (float)Hl - 0.5f;
          const float fx0 = floorf(px), fy0 = floorf(py);
          const float ax = px - fx0, ay = py - fy0;
          const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)Wl + 1.f);
          const int y0 = (int)fminf(fmaxf(fy0, -2.f), (float)Hl + 1.f);
          const float aw = w[j] * inv;
          float s_dot = 0.f, sx = 0.f, sy = 0.f;
          // the four corners are fetched unconditionally from coordinates clamped into the level (as msda_fwd): a load
          // under `if (in range)` is waited for on the spot, 48 exposed latencies per query
          uint4 raw4[4];
          bool inb4[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
            inb4[t] = (unsigned)xx < (unsigned)Wl && (unsigned)yy < (unsigned)Hl;
            const int xc = xx < 0 ? 0 : (xx >= Wl ? Wl - 1 : xx), yc = yy < 0 ? 0 : (yy >= Hl ? Hl - 1 : yy);
            raw4[t] = *reinterpret_cast<const uint4*>(vl + ((int64_t)yc * Wl + xc) * D);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
            if (inb4[t]) {
              const float bx = (t & 1) ? ax : 1.f - ax, by = (t >> 1) ? ay : 1.f - ay;
              const float dbx = (t & 1) ? 1.f : -1.f, dby = (t >> 1) ? 1.f : -1.f;
              const int64_t o = ((int64_t)yy * Wl + xx) * D;
              const uint32_t* pw = reinterpret_cast<const uint32_t*>(&raw4[t]);
              float dotv = 0.f;
              const float wv = aw * bx * by;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float f0, f1;
                unpack2<T>(pw[e], f0, f1);
                dotv += g[2 * e] * f0 + g[2 * e + 1] * f1;
                if (DVALUE) {
                  atomicAdd(dvl + o + 2 * e, wv * g[2 * e]);
                  atomicAdd(dvl + o + 2 * e + 1, wv * g[2 * e + 1]);
                }
              }
              s_dot += bx * by * dotv;
              sx += dbx * by * dotv;
              sy += bx * dby * dotv;
            }
          }
          float* r = red + (m * LP + j) * 3;
          if (pow2) {   // the cph lanes of a head are an aligned lane group: xor-shuffle sum, one writer (no LDS atomics)
            for (int o = 1; o < cph; o <<= 1) {
              s_dot += __shfl_xor(s_dot, o, 64);
              sx += __shfl_xor(sx, o, 64);
              sy += __shfl_xor(sy, o, 64);
            }
            if ((c & (cph - 1)) == 0) { r[0] = s_dot; r[1] = sx; r[2] = sy; }
          } else {
            atomicAdd(r + 0, s_dot);
            atomicAdd(r + 1, sx);
            atomicAdd(r + 2, sy);
          }
        }
      }
    }
    __syncthreads();
    // one thread per head: softmax backward over (l, p) and the offset gradients
    if ((int)threadIdx.x < M) {
      const int mm = threadIdx.x;
      const float* lg = orow + (int64_t)M * LP * 2 + mm * LP;
      float w[MAX_LP];
      float mx = -1e30f;
      for (int j = 0; j < LP; ++j) {
        w[j] = lg[j];
        mx = fmaxf(mx, w[j]);
      }
      float den = 0.f;
      for (int j = 0; j < LP; ++j) {
        w[j] = __expf(w[j] - mx);
        den += w[j];
      }
      const float inv = 1.0f / den;
      float dot = 0.f;
      for (int j = 0; j < LP; ++j) dot += w[j] * inv * red[(mm * LP + j) * 3];
      float* drow = doffaw + bq * ld_offaw;
      for (int j = 0; j < LP; ++j) {
        const float aw = w[j] * inv;
        drow[(mm * LP + j) * 2 + 0] = aw * red[(mm * LP + j) * 3 + 1];
        drow[(mm * LP + j) * 2 + 1] = aw * red[(mm * LP + j) * 3 + 2];
        drow[M * LP * 2 + mm * LP + j] = aw * (red[(mm * LP + j) * 3] - dot);
      }
    }
    __syncthreads();
  }
}

// d value without global atomics: one workgroup owns the gradient of CH consecutive channels of ALL Lin value pixels of
// one image in LDS (Lin * CH * 4 bytes), every thread walks queries, recomputes their sampling taps and scatters
// A_j * bw_t * d out into the tile with LDS atomics; the tile is then stored once.  ~27 ms -> well under 1 ms per call
// against one fp32 global atomic per (tap, channel).  Summation order inside the tile still varies from run to run.
template <int CH>
__global__ __launch_bounds__(256) void msda_bwd_dvalue_kernel(const float* __restrict__ offaw, int64_t ld_offaw,
                                                              const float* __restrict__ ref, const int* __restrict__ shapes,
                                                              const int* __restrict__ starts, const float* __restrict__ dout,
                                                              float* __restrict__ dvalue, int B, int Lq, int Lin, int M,
                                                              int L, int P, int Dh) {
  extern __shared__ float tile[];  // [Lin][CH]
  const int D = M * Dh;
  const int LP = L * P;
  const int chunks = D / CH;
  const int b = blockIdx.x / chunks, cc = blockIdx.x - b * chunks;
  const int ch0 = cc * CH;
  const int m = ch0 / Dh;
  for (int i = threadIdx.x; i < Lin * CH; i += blockDim.x) tile[i] = 0.f;
  __syncthreads();
  for (int q = threadIdx.x; q < Lq; q += blockDim.x) {
    const int64_t bq = (int64_t)b * Lq + q;
    const float* orow = offaw + bq * ld_offaw;
    const float* lg = orow + (int64_t)M * LP * 2 + m * LP;
    float w[MAX_LP];
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < MAX_LP; ++j)
      if (j < LP) {
        w[j] = lg[j];
        mx = fmaxf(mx, w[j]);
      }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_LP; ++j)
      if (j < LP) {
        w[j] = __expf(w[j] - mx);
        den += w[j];
      }
    const float inv = 1.0f / den;
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    float g[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) g[e] = dout[bq * D + ch0 + e];
    for (int l = 0; l < L; ++l) {
      const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
      float* tl = tile + (int64_t)starts[l] * CH;
      for (int p = 0; p < P; ++p) {
        const int j = l * P + p;
        const float ox = orow[(m * LP + j) * 2], oy = orow[(m * LP + j) * 2 + 1];
        const float lx = rx + ox / (float)Wl, ly = ry + oy / (float)Hl;
        const float px = lx * (float)Wl - 0.5f, py = ly * (float)Hl - 0.5f;
        const float fx0 = floorf(px), fy0 = floorf(py);
        const float ax = px - fx0, ay = py - fy0;
        const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)Wl + 1.f);
        const int y0 = (int)fminf(fmaxf(fy0, -2.f), (float)Hl + 1.f);
        const float aw = w[j] * inv;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
          if ((unsigned)xx < (unsigned)Wl && (unsigned)yy < (unsigned)Hl) {
            const float wv = aw * ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
            float* dst = tl + (yy * Wl + xx) * CH;
#pragma unroll
            for (int e = 0; e < CH; ++e) atomicAdd(dst + e, wv * g[e]);
          }
        }
      }
    }
  }
  __syncthreads();
  float* out = dvalue + (int64_t)b * Lin * D + ch0;
  for (int i = threadIdx.x; i < Lin * CH; i += blockDim.x) {
    const int pix = i / CH, e = i - pix * CH;
    out[(int64_t)pix * D + e] = tile[i];
  }
}

// The sampling operator of one (image, head) as a dense matrix, transposed: ST[b, m, pix, q] = sum over the points j and
// taps t of query q that land on pixel pix of A_j * bw_t  (16-bit, q contiguous, row stride ldt, caller-zeroed).
// d value[b, :, m, :] = ST[b, m] . d out[b, :, m, :] is then ONE batched MFMA GEMM (K = Lq) instead of Lq*L*P*4*Dh scattered
// atomic adds: 96 x (1764 x 6949 x 128) = 0.3 TFLOP per call at GEMM speed against 6 ms of LDS atomics.
// One thread per (b, q, m); its <= 4*L*P entries are merged in registers (two points of a query can share a pixel) so
// every ST element is written by exactly one thread: no atomics, deterministic.
template <typename T>
__global__ __launch_bounds__(256) void msda_sampling_matrix_kernel(const float* __restrict__ offaw, int64_t ld_offaw,
                                                                   const float* __restrict__ ref, const int* __restrict__ shapes,
                                                                   const int* __restrict__ starts, T* __restrict__ ST,
                                                                   int64_t ldt, int B, int Lq, int Lin, int M, int L, int P) {
  const int LP = L * P;
  const int64_t total = (int64_t)B * Lq * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i % M);
    const int64_t bq = i / M;
    const int q = (int)(bq % Lq);
    const int b = (int)(bq / Lq);
    const float* orow = offaw + bq * ld_offaw;
    const float* lg = orow + (int64_t)M * LP * 2 + m * LP;
    float w[MAX_LP];
    float mx = -1e30f;
    for (int j = 0; j < LP; ++j) {
      w[j] = lg[j];
      mx = fmaxf(mx, w[j]);
    }
    float den = 0.f;
    for (int j = 0; j < LP; ++j) {
      w[j] = __expf(w[j] - mx);
      den += w[j];
    }
    const float inv = 1.0f / den;
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    int pix[MAX_LP * 4];
    float val[MAX_LP * 4];
    int n = 0;
    for (int l = 0; l < L; ++l) {
      const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1], s0 = starts[l];
      for (int p = 0; p < P; ++p) {
        const int j = l * P + p;
        const float ox = orow[(m * LP + j) * 2], oy = orow[(m * LP + j) * 2 + 1];
        const float lx = rx + ox / (float)Wl, ly = ry + oy / (float)Hl;
        const float px = lx * (float)Wl - 0.5f, py = ly * (float)Hl - 0.5f;
        const float fx0 = floorf(px), fy0 = floorf(py);
        const float ax = px - fx0, ay = py - fy0;
        const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)Wl + 1.f);
        const int y0 = (int)fminf(fmaxf(fy0, -2.f), (float)Hl + 1.f);
        const float aw = w[j] * inv;
        for (int t = 0; t < 4; ++t) {
          const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
          if ((unsigned)xx < (unsigned)Wl && (unsigned)yy < (unsigned)Hl) {
            const int pp = s0 + yy * Wl + xx;
            const float wv = aw * ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
            int k = 0;
            for (; k < n; ++k)
              if (pix[k] == pp) break;
            if (k == n) {
              pix[n] = pp;
              val[n] = wv;
              ++n;
            } else {
              val[k] += wv;
            }
          }
        }
      }
    }
    T* base = ST + ((int64_t)(b * M + m) * Lin) * ldt + q;
    for (int k = 0; k < n; ++k) base[(int64_t)pix[k] * ldt] = to_t16<T>(val[k]);
  }
}

// ---- d value as a gather over taps grouped by destination pixel (round 4) ------------------------------------------------------
// The dense form above spends a 2.4 GB matrix (99 % zeros), its memset and 8 batched GEMMs on what is a sparse sum:
//     d value[b, pix, m, :] = sum over the taps (q, l, p, corner) that land on pix of  w_tap * d out[b, q, m, :].
// Here the <= 4 L P taps of every (b, q, m) are bucketed by pixel with a counting sort — (1) histogram per (b, m, pix),
// (2) exclusive scan per (b, m), (3) fill of (q, weight) records at positions drawn from per-pixel cursors — and (4) one wave
// per (b, m, pix) sums its records.  The positions inside a bucket depend on the atomics' arrival order; the SUM does not: it is
// taken in 64-bit integers (each product rounded to a fixed-point grid of 2^-30 x the operand's power-of-two range), and integer
// addition is associative — bitwise reproducible run to run, no atomics on the output, every output element written.
// msda_taps_kernel<FILL>: one thread per (b, q, m); the same tap arithmetic as msda_sampling_matrix_kernel.
template <bool FILL>
__global__ __launch_bounds__(256) void msda_taps_kernel(const float* __restrict__ offaw, int64_t ld_offaw, const float* __restrict__ ref,
                                                        const int* __restrict__ shapes, const int* __restrict__ starts,
                                                        int* __restrict__ cnt, int2* __restrict__ rec, int cap, int B, int Lq, int Lin,
                                                        int M, int L, int P) {
  const int LP = L * P;
  const int64_t total = (int64_t)B * Lq * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i % M);
    const int64_t bq = i / M;
    const int q = (int)(bq % Lq);
    const int b = (int)(bq / Lq);
    const float* orow = offaw + bq * ld_offaw;
    const float* lg = orow + (int64_t)M * LP * 2 + m * LP;
    float w[MAX_LP];
    float mx = -1e30f;
    for (int j = 0; j < LP; ++j) {
      w[j] = lg[j];
      mx = fmaxf(mx, w[j]);
    }
    float den = 0.f;
    for (int j = 0; j < LP; ++j) {
      w[j] = __expf(w[j] - mx);
      den += w[j];
    }
    const float inv = 1.0f / den;
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    int* const cbase = cnt + (int64_t)(b * M + m) * Lin;
    int2* const rbase = rec + (int64_t)(b * M + m) * cap;
    for (int l = 0; l < L; ++l) {
      const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1], s0 = starts[l];
      for (int p = 0; p < P; ++p) {
        const int j = l * P + p;
        const float ox = orow[(m * LP + j) * 2], oy = orow[(m * LP + j) * 2 + 1];
        const float lx = rx + ox / (float)Wl, ly = ry + oy / (float)Hl;
        const float px = lx * (float)Wl - 0.5f, py = ly * (float)Hl - 0.5f;
        const float fx0 = floorf(px), fy0 = floorf(py);
        const float ax = px - fx0, ay = py - fy0;
        const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)Wl + 1.f);
        const int y0 = (int)fminf(fmaxf(fy0, -2.f), (float)Hl + 1.f);
        const float aw = w[j] * inv;
        // the four corners' atomics are issued together and their returned positions consumed afterwards: four round trips in
        // flight per point instead of one after the other
        int pos[4];
        bool ok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
          ok[t] = (unsigned)xx < (unsigned)Wl && (unsigned)yy < (unsigned)Hl;
          pos[t] = ok[t] ? atomicAdd(cbase + (s0 + yy * Wl + xx), 1) : 0;      // FILL: cursors start at the bucket offsets
        }
        if (FILL) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (ok[t]) {
              const float wv = aw * ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
              rbase[pos[t]] = make_int2(q, __builtin_bit_cast(int, wv));
            }
        }
      }
    }
  }
}
// one workgroup per (b, m): counts[Lin] -> offs[Lin + 1] (exclusive scan; offs[Lin] = the bucket total) and the fill cursors
__global__ __launch_bounds__(256) void msda_scan_kernel(int* __restrict__ cnt, int* __restrict__ offs, int Lin) {
  __shared__ int part[256];
  int* c = cnt + (int64_t)blockIdx.x * Lin;
  int* o = offs + (int64_t)blockIdx.x * (Lin + 1);
  const int per = (Lin + 255) / 256;
  const int i0 = threadIdx.x * per, i1 = min(i0 + per, Lin);
  int s = 0;
  for (int i = i0; i < i1; ++i) s += c[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = part[t];
      part[t] = run;
      run += v;
    }
    o[Lin] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = i0; i < i1; ++i) {
    const int v = c[i];
    o[i] = run;
    c[i] = run;          // the counts become the fill cursors
    run += v;
  }
}
// one wave per (b, m, pix): fixed-point sum of w * d out over the bucket's records
template <typename T>
__global__ __launch_bounds__(256) void msda_vgrad_kernel(const int* __restrict__ offs, const int2* __restrict__ rec, int cap,
                                                         const T* __restrict__ dout16, const float* __restrict__ amax,
                                                         float* __restrict__ dvalue, int B, int Lq, int Lin, int M, int Dh) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= (int64_t)B * M * Lin) return;
  const int pix = (int)(wv % Lin);
  const int bm = (int)(wv / Lin);
  const int m = bm % M, b = bm / M;
  const int* o = offs + (int64_t)bm * (Lin + 1);
  const int e0 = o[pix], e1 = o[pix + 1];
  const int2* r = rec + (int64_t)bm * cap;
  const int D = M * Dh;
  // fixed-point grid: |w| <= 1 and |d out| <= amax < 2^(e + 1), so |w d| 2^(29 - e) < 2^30
  int e = (int)((__builtin_bit_cast(uint32_t, *amax) >> 23) & 0xFF) - 127;
  e = e < -100 ? -100 : (e > 90 ? 90 : e);
  const float sc = __builtin_bit_cast(float, (uint32_t)(127 + 29 - e) << 23), isc = __builtin_bit_cast(float, (uint32_t)(127 - 29 + e) << 23);
  for (int d0 = 2 * lane; d0 < Dh; d0 += 128) {     // Dh <= 128: one trip; wider heads: lanes loop
    long long a0 = 0, a1 = 0;
    // eight records at a time: their gathers are independent loads in flight together (one record per trip left every wave
    // waiting a full memory round trip per record: the kernel ran no faster than the dense form it replaces)
    for (int k = e0; k < e1; k += 8) {
      int2 en[8];
      uint32_t xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        en[u] = r[k + u < e1 ? k + u : e1 - 1];
        if (k + u >= e1) en[u].y = 0;        // weight 0.0f
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const uint32_t*>(dout16 + ((int64_t)b * Lq + en[u].x) * D + m * Dh + d0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float w = __builtin_bit_cast(float, en[u].y);
        float x0, x1;
        unpack2<T>(xv[u], x0, x1);
        a0 += (long long)__float2int_rn(w * x0 * sc);
        a1 += (long long)__float2int_rn(w * x1 * sc);
      }
    }
    *reinterpret_cast<float2*>(dvalue + ((int64_t)b * Lin + pix) * D + m * Dh + d0) = make_float2((float)a0 * isc, (float)a1 * isc);
  }
}

// ---- DWConv 3x3 + GELU backward ---------------------------------------------------------------------------------------
__device__ __forceinline__ void locate(int tok, const int* shapes, const int* starts, int L, int& l, int& y, int& x, int& H,
                                       int& W, int& s0) {
  l = 0;
  for (int k = 1; k < L; ++k)
    if (tok >= starts[k]) l = k;
  H = shapes[2 * l];
  W = shapes[2 * l + 1];
  s0 = starts[l];
  const int r = tok - s0;
  y = r / W;
  x = r - y * W;
}

// g[b,tok,c] = dy * gelu'(dwconv(x)[tok] + bias);  partial[blk][10][C]: rows 0..8 = d w9[tap] = sum g * x[tok+tap], row 9 = d bias.
// Block = 256 threads = (C/4 channel chunks) x (rows): a thread keeps its channel chunk (C/4 must divide 256).
__global__ __launch_bounds__(256) void dwconv_gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                              const float* __restrict__ bias, const int* __restrict__ shapes,
                                                              const int* __restrict__ starts, int L,
                                                              const float* __restrict__ dy, float* __restrict__ g,
                                                              float* __restrict__ partial, int B, int Ntok, int C,
                                                              int rows_per_block) {
  __shared__ float red[256 * 4];
  const int cpt = C >> 2;
  const int rw = 256 / cpt;
  const int cx = threadIdx.x % cpt, ry = threadIdx.x / cpt;
  const int64_t rows = (int64_t)B * Ntok;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float4 wt[9], acc[10];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = reinterpret_cast<const float4*>(w9 + (int64_t)t * C)[cx];
  const float4 bs = reinterpret_cast<const float4*>(bias)[cx];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = r0 + ry; row < r1; row += rw) {
    const int b = (int)(row / Ntok), tok = (int)(row - (int64_t)b * Ntok);
    int l, y, xx, H, W, s0;
    locate(tok, shapes, starts, L, l, y, xx, H, W, s0);
    const float* xb = x + ((int64_t)b * Ntok + s0) * C;
    float4 pre = bs, xv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, x2 = xx + t % 3 - 1;
      xv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)yy < (unsigned)H && (unsigned)x2 < (unsigned)W) {
        xv[t] = reinterpret_cast<const float4*>(xb + ((int64_t)yy * W + x2) * C)[cx];
        pre.x += wt[t].x * xv[t].x; pre.y += wt[t].y * xv[t].y; pre.z += wt[t].z * xv[t].z; pre.w += wt[t].w * xv[t].w;
      }
    }
    const float4 d = reinterpret_cast<const float4*>(dy + row * C)[cx];
    float4 gg;
    gg.x = d.x * gelu_erf_grad(pre.x); gg.y = d.y * gelu_erf_grad(pre.y);
    gg.z = d.z * gelu_erf_grad(pre.z); gg.w = d.w * gelu_erf_grad(pre.w);
    reinterpret_cast<float4*>(g + row * C)[cx] = gg;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t].x += gg.x * xv[t].x; acc[t].y += gg.y * xv[t].y; acc[t].z += gg.z * xv[t].z; acc[t].w += gg.w * xv[t].w;
    }
    acc[9].x += gg.x; acc[9].y += gg.y; acc[9].z += gg.z; acc[9].w += gg.w;
  }
  // fold the rw row lanes of each channel chunk, one accumulator at a time
  for (int t = 0; t < 10; ++t) {
    reinterpret_cast<float4*>(red)[threadIdx.x] = acc[t];
    __syncthreads();
    if (ry == 0) {
      float4 s = acc[t];
      for (int k = 1; k < rw; ++k) {
        const float4 o = reinterpret_cast<const float4*>(red)[k * cpt + cx];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 10 + t) * C)[cx] = s;
    }
    __syncthreads();
  }
}

// dx[tok] = sum_taps w9[tap] * g[tok - tap offset]  (depthwise transposed conv = correlation with the mirrored taps) -> 16-bit
template <typename T>
__global__ __launch_bounds__(256) void dwconv_t_kernel(const float* __restrict__ g, const float* __restrict__ w9,
                                                       const int* __restrict__ shapes, const int* __restrict__ starts, int L,
                                                       T* __restrict__ out, int B, int Ntok, int C) {
  const int cpt = C >> 2;
  const int64_t total = (int64_t)B * Ntok * cpt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpt);
    const int64_t row = i / cpt;
    const int b = (int)(row / Ntok), tok = (int)(row - (int64_t)b * Ntok);
    int l, y, xx, H, W, s0;
    locate(tok, shapes, starts, L, l, y, xx, H, W, s0);
    const float* gb = g + ((int64_t)b * Ntok + s0) * C;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // forward: pre[y'] += w[t] * x[y' + dt]; so x[y] receives w[t] * g[y - dt]
      const int yy = y - (t / 3 - 1), x2 = xx - (t % 3 - 1);
      if ((unsigned)yy < (unsigned)H && (unsigned)x2 < (unsigned)W) {
        const float4 gv = reinterpret_cast<const float4*>(gb + ((int64_t)yy * W + x2) * C)[c];
        const float4 wv = reinterpret_cast<const float4*>(w9 + (int64_t)t * C)[c];
        a.x += wv.x * gv.x; a.y += wv.y * gv.y; a.z += wv.z * gv.z; a.w += wv.w * gv.w;
      }
    }
    uint2 o;
    o.x = pack2<T>(a.x, a.y);
    o.y = pack2<T>(a.z, a.w);
    reinterpret_cast<uint2*>(out)[i] = o;
  }
}

}  // namespace

#define DT_OK(dtype, name) ASIS_REQUIRE(dtype == ASIS_F16 || dtype == ASIS_BF16, name ": bad dtype %d", dtype)

extern "C" int asis_msda_bwd(void* stream, int dtype, const void* value, const float* offaw, int64_t ld_offaw,
                             const float* ref, const int32_t* shapes, const int32_t* starts, const float* dout, float* dvalue,
                             float* doffaw, int B, int Lq, int Lin, int M, int L, int P, int Dh) {
  ASIS_REQUIRE(value && offaw && ref && shapes && starts && dout && doffaw, "asis_msda_bwd: null pointer");
  DT_OK(dtype, "asis_msda_bwd");
  ASIS_REQUIRE(Dh % 8 == 0 && M >= 1 && M <= MAX_M && L * P >= 1 && L * P <= MAX_LP && M * Dh / 8 <= 256,
               "asis_msda_bwd: need Dh %% 8 == 0, M <= %d, L*P <= %d, M*Dh <= 2048", MAX_M, MAX_LP);
  ASIS_REQUIRE(ld_offaw >= (int64_t)M * L * P * 3, "asis_msda_bwd: ld_offaw too small");
  ASIS_REQUIRE(asis_aligned16(value) && asis_aligned16(dout), "asis_msda_bwd: value / dout must be 16-byte aligned");
  const int cpq = M * Dh / 8;
  int threads = (cpq + 63) / 64 * 64;
  if (threads < 64) threads = 64;
  int64_t grid = (int64_t)B * Lq;
  if (grid > 65535 * 8) grid = 65535 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // d value: LDS-tiled when CH channels of all Lin pixels fit in 128 KiB, else the global-atomics form
  int CH = 0;
  for (int c : {8, 4, 2})
    if (Dh % c == 0 && (int64_t)Lin * c * 4 <= 128 * 1024) { CH = c; break; }
  const bool skip_dvalue = dvalue == nullptr;  // the caller takes d value from asis_msda_sampling_matrix + a GEMM
  const bool tiled = skip_dvalue || (CH != 0 && (int64_t)B * (M * Dh / (CH ? CH : 1)) <= 0x7fffffff);
  if (dtype == ASIS_F16) {
    if (tiled) hipLaunchKernelGGL((msda_bwd_kernel<f16, false>), dim3((unsigned)grid), dim3(threads), 0, s, reinterpret_cast<const f16*>(value), offaw, ld_offaw, ref, shapes, starts, dout, dvalue, doffaw, B, Lq, Lin, M, L, P, Dh);
    else hipLaunchKernelGGL((msda_bwd_kernel<f16, true>), dim3((unsigned)grid), dim3(threads), 0, s, reinterpret_cast<const f16*>(value), offaw, ld_offaw, ref, shapes, starts, dout, dvalue, doffaw, B, Lq, Lin, M, L, P, Dh);
  } else {
    if (tiled) hipLaunchKernelGGL((msda_bwd_kernel<bf16, false>), dim3((unsigned)grid), dim3(threads), 0, s, reinterpret_cast<const bf16*>(value), offaw, ld_offaw, ref, shapes, starts, dout, dvalue, doffaw, B, Lq, Lin, M, L, P, Dh);
    else hipLaunchKernelGGL((msda_bwd_kernel<bf16, true>), dim3((unsigned)grid), dim3(threads), 0, s, reinterpret_cast<const bf16*>(value), offaw, ld_offaw, ref, shapes, starts, dout, dvalue, doffaw, B, Lq, Lin, M, L, P, Dh);
  }
  if (tiled && !skip_dvalue) {
    const unsigned nb = (unsigned)(B * (M * Dh / CH));
    const size_t lds = (size_t)Lin * CH * 4;
    if (CH == 8) {
      static bool set8 = false;
      if (!set8) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_bwd_dvalue_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set8 = true; }
      hipLaunchKernelGGL((msda_bwd_dvalue_kernel<8>), dim3(nb), dim3(256), lds, s, offaw, ld_offaw, ref, shapes, starts, dout, dvalue, B, Lq, Lin, M, L, P, Dh);
    } else if (CH == 4) {
      static bool set4 = false;
      if (!set4) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_bwd_dvalue_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set4 = true; }
      hipLaunchKernelGGL((msda_bwd_dvalue_kernel<4>), dim3(nb), dim3(256), lds, s, offaw, ld_offaw, ref, shapes, starts, dout, dvalue, B, Lq, Lin, M, L, P, Dh);
    } else {
      static bool set2 = false;
      if (!set2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_bwd_dvalue_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set2 = true; }
      hipLaunchKernelGGL((msda_bwd_dvalue_kernel<2>), dim3(nb), dim3(256), lds, s, offaw, ld_offaw, ref, shapes, starts, dout, dvalue, B, Lq, Lin, M, L, P, Dh);
    }
  }
  ASIS_CHECK_LAUNCH("asis_msda_bwd");
  return ASIS_OK;
}

extern "C" int asis_dwconv_bwd_nblk(int64_t rows) {
  int64_t n = (rows + 63) / 64;
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int asis_dwconv_gelu_bwd(void* stream, int dtype, const float* x, const float* w9, const float* bias,
                                    const int32_t* shapes, const int32_t* starts, int L, const float* dy, float* g,
                                    float* partial, void* dx, int B, int Ntok, int C) {
  ASIS_REQUIRE(x && w9 && bias && shapes && starts && dy && g && partial && dx, "asis_dwconv_gelu_bwd: null pointer");
  DT_OK(dtype, "asis_dwconv_gelu_bwd");
  ASIS_REQUIRE(C % 4 == 0 && C >= 4 && C / 4 <= 256 && 256 % (C / 4) == 0,
               "asis_dwconv_gelu_bwd: C=%d must be 4*2^k <= 1024", C);
  ASIS_REQUIRE(L >= 1 && B > 0 && Ntok > 0, "asis_dwconv_gelu_bwd: bad shape");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * Ntok;
  const int nblk = asis_dwconv_bwd_nblk(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  hipLaunchKernelGGL(dwconv_gelu_bwd_kernel, dim3(nblk), dim3(256), 0, s, x, w9, bias, shapes, starts, L, dy, g, partial, B, Ntok,
                     C, rpb);
  int64_t gsz = (rows * (C / 4) + 255) / 256;
  if (gsz > 65535 * 4) gsz = 65535 * 4;
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((dwconv_t_kernel<f16>), dim3((unsigned)gsz), dim3(256), 0, s, g, w9, shapes, starts, L, reinterpret_cast<f16*>(dx),
                       B, Ntok, C);
  else
    hipLaunchKernelGGL((dwconv_t_kernel<bf16>), dim3((unsigned)gsz), dim3(256), 0, s, g, w9, shapes, starts, L,
                       reinterpret_cast<bf16*>(dx), B, Ntok, C);
  ASIS_CHECK_LAUNCH("asis_dwconv_gelu_bwd");
  return ASIS_OK;
}

extern "C" int asis_msda_vgrad_cap(int Lq, int L, int P) { return Lq * L * P * 4; }

extern "C" int asis_msda_value_grad(void* stream, int dtype, const float* offaw, int64_t ld_offaw, const float* ref,
                                    const int32_t* shapes, const int32_t* starts, const void* dout16, const float* amax,
                                    int32_t* cnt, int32_t* offs, void* rec, float* dvalue, int B, int Lq, int Lin, int M, int L,
                                    int P, int Dh) {
  ASIS_REQUIRE(offaw && ref && shapes && starts && dout16 && amax && cnt && offs && rec && dvalue, "asis_msda_value_grad: null pointer");
  DT_OK(dtype, "asis_msda_value_grad");
  ASIS_REQUIRE(M >= 1 && L * P >= 1 && L * P <= MAX_LP && Dh >= 2 && Dh % 2 == 0, "asis_msda_value_grad: bad shape (L*P <= %d, Dh even)", MAX_LP);
  ASIS_REQUIRE(ld_offaw >= (int64_t)M * L * P * 3, "asis_msda_value_grad: ld_offaw too small");
  ASIS_REQUIRE((int64_t)B * M * Lin < (1ll << 31) && (int64_t)Lq * L * P * 4 < (1ll << 31), "asis_msda_value_grad: shape too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int cap = asis_msda_vgrad_cap(Lq, L, P);
  ASIS_REQUIRE(hipMemsetAsync(cnt, 0, (size_t)B * M * Lin * sizeof(int32_t), s) == hipSuccess, "asis_msda_value_grad: memset failed");
  int64_t g = ((int64_t)B * Lq * M + 255) / 256;
  if (g > 65535 * 8) g = 65535 * 8;
  hipLaunchKernelGGL((msda_taps_kernel<false>), dim3((unsigned)g), dim3(256), 0, s, offaw, ld_offaw, ref, shapes, starts,
                     reinterpret_cast<int*>(cnt), reinterpret_cast<int2*>(rec), cap, B, Lq, Lin, M, L, P);
  hipLaunchKernelGGL(msda_scan_kernel, dim3((unsigned)(B * M)), dim3(256), 0, s, reinterpret_cast<int*>(cnt), reinterpret_cast<int*>(offs), Lin);
  hipLaunchKernelGGL((msda_taps_kernel<true>), dim3((unsigned)g), dim3(256), 0, s, offaw, ld_offaw, ref, shapes, starts,
                     reinterpret_cast<int*>(cnt), reinterpret_cast<int2*>(rec), cap, B, Lq, Lin, M, L, P);
  const int64_t waves = (int64_t)B * M * Lin;
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((msda_vgrad_kernel<f16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, reinterpret_cast<const int*>(offs),
                       reinterpret_cast<const int2*>(rec), cap, reinterpret_cast<const f16*>(dout16), amax, dvalue, B, Lq, Lin, M, Dh);
  else
    hipLaunchKernelGGL((msda_vgrad_kernel<bf16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, reinterpret_cast<const int*>(offs),
                       reinterpret_cast<const int2*>(rec), cap, reinterpret_cast<const bf16*>(dout16), amax, dvalue, B, Lq, Lin, M, Dh);
  ASIS_CHECK_LAUNCH("asis_msda_value_grad");
  return ASIS_OK;
}

extern "C" int asis_msda_sampling_matrix(void* stream, int dtype, const float* offaw, int64_t ld_offaw, const float* ref,
                                         const int32_t* shapes, const int32_t* starts, void* ST, int64_t ldt, int B, int Lq,
                                         int Lin, int M, int L, int P) {
  ASIS_REQUIRE(offaw && ref && shapes && starts && ST, "asis_msda_sampling_matrix: null pointer");
  DT_OK(dtype, "asis_msda_sampling_matrix");
  ASIS_REQUIRE(M >= 1 && L * P >= 1 && L * P <= MAX_LP && ldt >= Lq, "asis_msda_sampling_matrix: bad shape (L*P <= %d, ldt >= Lq)", MAX_LP);
  ASIS_REQUIRE(ld_offaw >= (int64_t)M * L * P * 3, "asis_msda_sampling_matrix: ld_offaw too small");
  int64_t g = ((int64_t)B * Lq * M + 255) / 256;
  if (g > 65535 * 8) g = 65535 * 8;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ASIS_F16)
    hipLaunchKernelGGL((msda_sampling_matrix_kernel<f16>), dim3((unsigned)g), dim3(256), 0, s, offaw, ld_offaw, ref, shapes, starts,
                       reinterpret_cast<f16*>(ST), ldt, B, Lq, Lin, M, L, P);
  else
    hipLaunchKernelGGL((msda_sampling_matrix_kernel<bf16>), dim3((unsigned)g), dim3(256), 0, s, offaw, ld_offaw, ref, shapes, starts,
                       reinterpret_cast<bf16*>(ST), ldt, B, Lq, Lin, M, L, P);
  ASIS_CHECK_LAUNCH("asis_msda_sampling_matrix");
  return ASIS_OK;
}
