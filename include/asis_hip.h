/*
 * asis_hip.h — C ABI of libasis_hip.so: the MI355X (gfx950 / CDNA4) kernels behind
 * adaptersis_amd, the drop-in for the ViT-adapter segmentation training step of
 * weimengmeng1999/AdapterSIS.
 *
 * Conventions
 *  - plain pointers and sizes only (device pointers unless a name says host); no framework
 *    types.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - the caller owns every buffer; kernels are launched asynchronously on `stream`; nothing
 *    here allocates, frees or synchronises (graph-capture safe).
 *  - return 0 on success, a negative ASIS_E* code on failure; asis_last_error() returns a
 *    thread-local message (Python raises ValueError/RuntimeError from it, mirroring the
 *    reference's AssertionError/ValueError conventions).
 *  - `dtype` selects the 16-bit MFMA operand type of activations/weights: ASIS_F16 (default:
 *    the reference's own autocast dtype, dinov2/configs/ssl_default_config.yaml:9, and the
 *    only one that meets the 1e-3 logits tolerance) or ASIS_BF16.
 *    Accumulation, the residual stream, LayerNorm/BatchNorm statistics, softmax and losses are
 *    always fp32.
 *  - "tokens" tensors are row-major [rows, channels]; image tensors handed between kernels are
 *    NHWC (= tokens of a (b, h, w) grid), so the reference's rearranges (train.py:389-406) are
 *    free.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 */
#ifndef ASIS_HIP_H
#define ASIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASIS_OK 0
#define ASIS_EINVAL (-1)  /* bad argument (shape / alignment / unsupported size) */
#define ASIS_ELAUNCH (-2) /* HIP launch or runtime error */

#define ASIS_F16 0
#define ASIS_BF16 1

#define ASIS_ACT_NONE 0
#define ASIS_ACT_GELU 1 /* exact erf GELU: dinov2/layers/mlp.py:35 (nn.GELU default) */
#define ASIS_ACT_RELU 2
#define ASIS_ACT_SILU_MUL 3 /* reserved: SwiGLU dinov2/layers/swiglu_ffn.py:30-34 (separate kernel) */

const char* asis_last_error(void);
int asis_version(void);
/* Number of HIP devices visible, or <0 on error (does not create a context). */
int asis_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM  C[b] = epilogue( A[b] (MxK) * B[b]^T (NxK) )  on MFMA 32x32x16, fp32 accumulate.
 * Replaces every nn.Linear / 1x1 / 3x3 conv on the path:
 *   attention.py:58,67 (qkv, proj)  mlp.py:35,38 (fc1, fc2)  swiglu_ffn.py:31,34
 *   ms_deform_attn.py:152,157,158,184   adapter_blocks.py:94,99   encoders.py:44-47
 *   patch_embed.py:75 (after asis_im2col_patch)   decoders.py:110-135 (conv=1, implicit GEMM)
 * A, B: 16-bit (dtype), K contiguous; lda/ldb in elements, multiples of 8; K multiple of 8.
 * Epilogue, per element (m, n):
 *     v = acc + bias_n[n] + bias_m[m]           (either may be NULL)
 *     v = act(v)
 *     v = v * scale_n[n]                        (NULL = 1; LayerScale layer_scale.py:27)
 *     v = v + res[b][m*ldr + n]                 (NULL = 0; fp32 residual block.py:112-113)
 *     C = out_f32 ? (float)v : (dtype)v
 * conv != 0: A is an NHWC activation [B, H, W, Cin] read as an implicit im2col matrix with
 *     M = B*OH*OW rows and K = KH*KW*Cin columns (k = (kh*KW + kw)*Cin + ci, zero padding);
 *     B must be laid out [N, KH, KW, Cin] (asis_pack_conv_weight).  batch must be 1.
 * ------------------------------------------------------------------------------------------- */
typedef struct asis_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  int64_t lda, ldb, ldc;             /* elements */
  int64_t strideA, strideB, strideC; /* elements per batch step (0 = shared) */
  int32_t batch;
  int32_t M, N, K;
  const float* bias_n;
  const float* bias_m;
  const float* scale_n;
  const float* res;
  int64_t ldr, strideR;
  int32_t act;
  int32_t out_f32;
  int32_t dtype;
  /* implicit-GEMM convolution */
  int32_t conv; /* 0 = dense A */
  int32_t B_, H, W, Cin, OH, OW, KH, KW, stride, pad;
  /* optional fp32 per-column partial statistics of the fp32 output (BatchNorm train mode):
   * stats[(tile_m * 2 + {0,1}) * N + n] = sum / sum of squares over the tile's valid rows. */
  float* stats;
} asis_gemm_desc;
int asis_gemm(void* stream, const asis_gemm_desc* d);
/* number of M tiles asis_gemm uses for M rows (size of the stats buffer = tiles*2*N floats) */
int asis_gemm_tiles_m(int M);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim, fp32 statistics, eps inside sqrt, biased variance
 * (nn.LayerNorm(eps=1e-6): vision_transformer.py:89, block.py:63,75, adapter_blocks.py:114,162).
 * x fp32 [rows, D] (row stride ldx); y = out_f32 ? fp32 : dtype, row stride ldy.
 * D multiple of 4, D <= 2048.
 * ------------------------------------------------------------------------------------------- */
int asis_layernorm(void* stream, int dtype, const float* x, int64_t ldx, const float* w, const float* b, float eps,
                   void* y, int64_t ldy, int out_f32, int64_t rows, int D);

/* ---------------------------------------------------------------------------------------------
 * Fused softmax attention forward, head dim 64 (all DINOv2 archs), no mask, no dropout:
 *   O = softmax(scale * Q K^T) V          (attention.py:60-66; MemEffAttention :84 is the same maths)
 * q, k: [B*N, ld] tokens-major with head h at columns h*64.. (q and k may live in one buffer);
 * vt:  V transposed per image: [B, H*64, ldvt] with keys contiguous (written by asis_gemm with
 *      A = W_v, B = x: a GEMM whose output is already V^T);  ldvt multiple of 8, >= N.
 * o:   [B*N, ldo] 16-bit, head h at columns h*64.
 * ------------------------------------------------------------------------------------------- */
int asis_attention_fwd(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                       int64_t ldvt, void* o, int64_t ldo, int B, int H, int N, float scale);

/* ---------------------------------------------------------------------------------------------
 * Patch-embed im2col (patch_embed.py:75: Conv2d k=s=P) : img fp32 NCHW [B,3,Himg,Wimg] ->
 * A 16-bit [B*(Himg/P)*(Wimg/P), ldk], k = c*P*P + i*P + j, columns >= 3*P*P zero-filled.
 * ------------------------------------------------------------------------------------------- */
int asis_im2col_patch(void* stream, int dtype, const float* img, int B, int Himg, int Wimg, int P, void* out,
                      int64_t ldk);

/* fp32 -> 16-bit cast with optional zero-padded columns: src [rows, cols] (ld_src) -> dst [rows, ld_dst],
 * columns cols..ld_dst-1 are written as zero.  Used to pack weights once. */
int asis_cast_pad(void* stream, int dtype, const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows,
                  int cols);

/* tokens_A[b, 0, :] = cls + pos[0];  tokens_A[b, 1+t, :] = x[b, t, :] + pos[1+t]
 * (vision_transformer.py:196-197 with the interpolated pos-embed cached per (H,W)). fp32. */
int asis_add_cls_pos(void* stream, const float* x, const float* cls, const float* pos, float* out, int B, int N,
                     int D);

#ifdef __cplusplus
}
#endif
#endif /* ASIS_HIP_H */
