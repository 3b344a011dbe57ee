// Lab: operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950) with fp8 e4m3 operands, found from the hardware itself
// (the ISA tables are not in this image).  What the GEMM kernels rely on, and what this program checks:
//   (1) A operand: all 32 bytes of lane l belong to row l & 15; B operand: to column l & 15;
//   (2) C/D layout = the 16x16 f32 map of every other 16x16 MFMA (col = l & 15, row = 4 (l >> 4) + reg);
//   (3) the K pairing is the identity on (lane >> 4, byte index): byte j of an A lane with l >> 4 == q multiplies byte j of the
//       B lanes with l >> 4 == q — so ANY assignment of K elements to (q, j) slots is valid as long as A and B use the same one;
//   (4) scale operands: byte 0 of the scale VGPR (op_sel 0) is an E8M0 exponent, result scaled by 2^(sa-127) * 2^(sb-127);
//   (5) v_cvt_pk_fp8_f32: round to nearest even, saturation behaviour at |x| > 448, byte order of the packed pair.
//     hipcc --offload-arch=gfx950 -O2 scripts/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void mx_kernel(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  c[l] = acc;
}

__global__ void cvt_kernel(const float* x, int* o, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], w, false);
    o[i] = w;
  }
}

static float e4m3_to_f(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = ldexpf((float)m / 8.f, -6);
  else if (e == 15 && m == 7) f = NAN;
  else f = ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -f : f;
}
static uint8_t f_to_e4m3(float x) {  // exact search (test values are exactly representable)
  for (int v = 0; v < 256; ++v)
    if (e4m3_to_f((uint8_t)v) == x && !(x == 0.f && v == 0x80)) return (uint8_t)v;
  fprintf(stderr, "value %g not representable\n", x);
  exit(2);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  uint8_t *da, *db; float* dc; int *dsa, *dsb;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dc, 64 * 16)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
  std::vector<uint8_t> ha(64 * 32), hb(64 * 32);
  std::vector<float> hc(64 * 4);
  std::vector<int> hsa(64, 127), hsb(64, 127);
  auto run = [&]() -> int {
    CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (v4f*)dc, dsa, dsb);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc.data(), dc, 64 * 16, hipMemcpyDeviceToHost));
    return 0;
  };
  auto C = [&](int row, int col) { return hc[((row >> 2) * 16 + col) * 4 + (row & 3)]; };  // assumed C map (2)
  const uint8_t ONE = f_to_e4m3(1.f);
  int bad = 0;

  // (1a)+(2): A one-hot at (lane, byte), B all ones -> exactly row (lane & 15) of C is 1 in every column
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; j += 5) {
      memset(ha.data(), 0, ha.size()); memset(hb.data(), ONE, hb.size());
      ha[l * 32 + j] = ONE;
      if (run()) return 1;
      for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c)
          if (C(r, c) != (r == (l & 15) ? 1.f : 0.f)) { if (++bad < 5) printf("A row map: lane %d byte %d -> C[%d][%d] = %g\n", l, j, r, c, C(r, c)); }
    }
  printf("(1a) A bytes of lane l all belong to row l&15 (and C map as assumed): %s\n", bad ? "NO" : "yes");
  // (1b): B one-hot, A all ones -> column (lane & 15)
  int bad_b = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; j += 5) {
      memset(hb.data(), 0, hb.size()); memset(ha.data(), ONE, ha.size());
      hb[l * 32 + j] = ONE;
      if (run()) return 1;
      for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c)
          if (C(r, c) != (c == (l & 15) ? 1.f : 0.f)) { if (++bad_b < 5) printf("B col map: lane %d byte %d -> C[%d][%d] = %g\n", l, j, r, c, C(r, c)); }
    }
  printf("(1b) B bytes of lane l all belong to column l&15: %s\n", bad_b ? "NO" : "yes");
  // (3): K pairing.  A one-hot at slot p = (q, j) of row 0; B column c (c = 0..6) carries bit c of its own slot index p' = 32 q' + j'
  int bad_k = 0;
  for (int p = 0; p < 128; ++p) {
    memset(ha.data(), 0, ha.size()); memset(hb.data(), 0, hb.size());
    ha[((p >> 5) * 16 + 0) * 32 + (p & 31)] = ONE;
    for (int pp = 0; pp < 128; ++pp)
      for (int c = 0; c < 7; ++c)
        if ((pp >> c) & 1) hb[((pp >> 5) * 16 + c) * 32 + (pp & 31)] = ONE;
    if (run()) return 1;
    int dec = 0;
    for (int c = 0; c < 7; ++c) dec |= (C(0, c) != 0.f) << c;
    if (dec != p) { if (++bad_k < 8) printf("K pairing: A slot %d (q %d, j %d) pairs with B slot %d (q %d, j %d)\n", p, p >> 5, p & 31, dec, dec >> 5, dec & 31); }
  }
  printf("(3) K pairing is the identity on (lane>>4, byte): %s\n", bad_k ? "NO" : "yes");
  // full random check with small exact values under the hypothesis k = 32 q + j
  {
    const float vals[] = {0.f, 0.5f, 1.f, -1.f, 2.f, -0.5f, 1.5f, -2.f, 0.25f, 3.f};
    float A[16][128], B[128][16];
    srand(3);
    for (int r = 0; r < 16; ++r) for (int k = 0; k < 128; ++k) { A[r][k] = vals[rand() % 10]; ha[((k >> 5) * 16 + r) * 32 + (k & 31)] = f_to_e4m3(A[r][k]); }
    for (int c = 0; c < 16; ++c) for (int k = 0; k < 128; ++k) { B[k][c] = vals[rand() % 10]; hb[((k >> 5) * 16 + c) * 32 + (k & 31)] = f_to_e4m3(B[k][c]); }
    for (int s = 0; s < 3; ++s) {
      const int sa = s == 1 ? 130 : 127, sb = s == 2 ? 120 : 127;
      for (int l = 0; l < 64; ++l) { hsa[l] = sa | 0x55000000; hsb[l] = sb | 0x00330000; }  // junk in the other bytes: op_sel 0 must take byte 0
      if (run()) return 1;
      int badr = 0;
      for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
        double ref = 0; for (int k = 0; k < 128; ++k) ref += (double)A[r][k] * B[k][c];
        ref *= ldexp(1.0, sa - 127) * ldexp(1.0, sb - 127);
        if (fabs(C(r, c) - ref) > 1e-6 * fabs(ref) + 1e-9) { if (++badr < 4) printf("random: C[%d][%d] = %g ref %g (sa %d sb %d)\n", r, c, C(r, c), ref, sa, sb); }
      }
      printf("(4) random 16x128x16 product, scale bytes sa=%d sb=%d (E8M0, byte 0): %s\n", sa, sb, badr ? "MISMATCH" : "exact");
    }
    // per-lane scales: lane l's scale applies to its own 32 K elements of its own row?
    for (int l = 0; l < 64; ++l) { hsa[l] = 127 + (l >> 4); hsb[l] = 127; }
    if (run()) return 1;
    int badl = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      double ref = 0; for (int k = 0; k < 128; ++k) ref += (double)A[r][k] * B[k][c] * ldexp(1.0, k >> 5);
      if (fabs(C(r, c) - ref) > 1e-6 * fabs(ref) + 1e-9) ++badl;
    }
    printf("(4b) per-lane A scale = scale of (row l&15, K block l>>4): %s\n", badl ? "NO" : "yes");
  }
  // (5) conversion
  {
    const float xs[] = {1.0f, -1.0f, 1.0625f, 1.1875f, 448.f, 449.f, 1000.f, -1e6f, 1e-3f, 0.0019f, 0.015625f, 0.017f, INFINITY, 0.3f, 240.f, 465.f};
    float* dx; int* dout;
    CK(hipMalloc(&dx, sizeof(xs))); CK(hipMalloc(&dout, 64));
    CK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dx, dout, 8);
    int ho[8];
    CK(hipMemcpy(ho, dout, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i)
      printf("(5) cvt_pk_fp8_f32(%g, %g) -> bytes %02x %02x = %g, %g\n", xs[2 * i], xs[2 * i + 1], ho[i] & 255, (ho[i] >> 8) & 255,
             e4m3_to_f(ho[i] & 255), e4m3_to_f((ho[i] >> 8) & 255));
  }
  return 0;
}
