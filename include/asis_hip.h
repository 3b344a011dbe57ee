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
#define ASIS_F32 2 /* only where a function says it accepts an fp32 output */

#define ASIS_ACT_NONE 0
#define ASIS_ACT_GELU 1 /* exact erf GELU: dinov2/layers/mlp.py:35 (nn.GELU default) */
#define ASIS_ACT_RELU 2
#define ASIS_ACT_SILU_MUL 3 /* SwiGLU gate in the epilogue, dinov2/layers/swiglu_ffn.py:30-34: B = w12 with its rows INTERLEAVED in   \
                               groups of 16 (rows 32g .. 32g+15 = x1 rows 16g .., rows 32g+16 .. 32g+31 = x2 rows 16g ..; bias_n    \
                               likewise), C = 16-bit [M, N / 2] = silu(x1 + b1) * (x2 + b2).  Dense launches on the 8-phase          \
                               one-tile-per-workgroup form only (plain or MX split operands): K % 64 == 0, M >= 256, N >= 256,       \
                               N % 32 == 0, ldc % 8 == 0, no residual / scale / statistics / second plane; else ASIS_EINVAL          \
                               (the caller then runs asis_swiglu on the fp32 pre-activation) */
#define ASIS_ACT_GELU_GRAD 4 /* backward of GELU in an input-gradient GEMM: C = (A B^T) * gelu'(aux), aux = 16-bit pre-activation */

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
  /* optional split-precision halves: A ~= A + A_lo, B ~= B + B_lo (rounding residuals, see asis_cast_pad part=1).
   * Both given: the kernel accumulates A*B + A_lo*B + A*B_lo in ONE pass over a virtual 3K-long reduction (same
   * layouts/strides as A and B).  Only one given (dense GEMMs: the weight operand of a linear layer, whose rounding
   * error is common to all rows): A*B + that one correction, a 2K-long reduction.  Needs the large-tile path:
   * K % 64 == 0, M >= 256, N >= 32 (conv: both halves, Cin % 64 == 0, fp32 output); otherwise ASIS_EINVAL — callers
   * then run accumulate passes. */
  const void* A_lo;
  const void* B_lo;
  /* act == ASIS_ACT_GELU_GRAD: 16-bit [M, N] pre-activation (row stride ld_aux, elements), large-tile path only
   * (M >= 256, N >= 128, K % 32 == 0, N % 4 == 0); otherwise ASIS_EINVAL and the caller runs asis_gelu16 separately */
  const void* aux;
  int64_t ld_aux;
  int32_t ksplit;                    /* > 1 (conv = 1 on the large-tile kernel, batch == 1, KH*KW divisible by it): the taps
                                        are cut into `ksplit` groups run side by side (gridDim.y); group p writes its fp32
                                        partial map at C + p * strideC, group 0 adds bias_n; `stats` and `res` must be null.
                                        Sum the parts with asis_reduce_rows, take BatchNorm statistics with asis_colstats.
                                        For layers whose tile count fills only part of the machine (EINVAL elsewhere). */
  /* ---- LayerNorm folded into the linear layers around it (block.py:89-114: x + ls * f(LN(x)); norm1 / norm2) -------------
   * The residual stream between two linear layers travels as TWO 16-bit planes, x ~= hi + lo (the same 4 bytes per element as
   * fp32), with per-row (mean, rstd); the layer that consumes LN(x) takes the hi plane as its A operand directly, with
   * diag(ln_weight) folded into its weight, and undoes the normalisation in its epilogue:
   *     LN(x) W^T + b  =  rstd * (x W'^T - mean * cs) + b',   W' = W diag(w_ln),  cs[n] = sum_k W'[n,k],  b' = b + W b_ln
   * so no LayerNorm kernel (and no extra HBM pass) runs between them.  All fields optional, dense 8-phase kernels only (the
   * persistent form and the K >= 1024 one-tile-per-workgroup form; ASIS_EINVAL elsewhere, callers then run asis_layernorm):
   *   C_lo      16-bit output (out_f32 == 0) as two planes: C = (dtype)v, C_lo = (dtype)(v - (float)C); same ldc
   *   rowstats  fp32 [M, ceil(N / 64), 2]: per row and 64-column group (sum, sum of squares) of the fp32 result v, the input of
   *             asis_ln_stats_finalize
   *   res16 / res16_lo / ldr16   the residual as two 16-bit planes instead of fp32 `res` (v += (float)hi + (float)lo)
   *   ln_mr     fp32 [M, 2] (mean, rstd) per A row (ln_cols != 0: per B row, i.e. per output COLUMN, for the swapped V^T GEMM)
   *   ln_cs     fp32 [N] cs (ln_cols != 0: [M], per output row); applied BEFORE bias_n / bias_m and the activation */
  /* ---- MX correction operands of a split convolution (conv != 0, A_lo and B_lo given, Cin % 64 == 0) or of a dense split
   * GEMM (conv == 0: K % 64 == 0, N >= 256, batch == 1; the linear layers of config.precise_level 2) -------------------------
   * mx_amax_a, mx_amax_b != NULL: device floats = the absolute maxima of the A and the B tensor; A_lo / B_lo then hold the MX form
   * of the rounding residuals (csrc/asis_common.h: two fp8 e4m3 bytes per element — activations (hi8, lo8), weights (lo8, hi8);
   * asis_bn_relu_upsample_mx / asis_decoder_input_mx / asis_pack_conv_weight_mx write them) and the reduction runs over TWO K
   * parts: A B on the 16-bit MFMA + one block-scaled fp8 MFMA pass (v_mfma_scale_f32_16x16x128_f8f6f4, twice the 16-bit rate
   * per byte) that yields A_hi B_lo + A_lo B_hi at ~4 significant bits — 2/3 of the MFMA time of the three 16-bit parts. */
  const float* mx_amax_a;
  const float* mx_amax_b;
  void* C_lo;
  float* rowstats;
  const void* res16;
  const void* res16_lo;
  int64_t ldr16;
  const float* ln_mr;
  const float* ln_cs;
  int32_t ln_cols;
} asis_gemm_desc;
/* (sum, sum of squares) partials [rows, groups, 2] of asis_gemm's rowstats -> mr [rows, 2] = (mean, rstd) of nn.LayerNorm
 * (biased variance, eps inside the sqrt; vision_transformer.py:89 eps = 1e-6); D = the row length the partials cover, each
 * group 64 columns (the last one D - 64 (groups - 1)); partials are combined as (count, mean, M2) triples. */
int asis_ln_stats_finalize(void* stream, const float* rowstats, int64_t rows, int groups, int D, float eps, float* mr);
/* fp32 [rows, D] (row stride ldx) -> 16-bit planes hi, lo (row stride ld16) + mr [rows, 2]: the entry into the folded
 * LayerNorm chain for a tensor that no GEMM epilogue produced (patch embedding output, adapter stage outputs). */
int asis_split_stats(void* stream, int dtype, const float* x, int64_t ldx, void* hi, void* lo, int64_t ld16, float* mr,
                     int64_t rows, int D, float eps);
int asis_gemm(void* stream, const asis_gemm_desc* d);
/* number of M tiles asis_gemm uses for M rows (size of the stats buffer = tiles*2*N floats) */
int asis_gemm_tiles_m(int M);
/* |x| maximum of an fp32 [rows, cols] tensor (row stride ld, cols % 4 == 0) into *amax (device float; reset != 0 zeroes it
 * first, reset == 0 accumulates over several tensors); asis_bn_relu_absmax: the maximum of relu(x * scale[c] + shift[c])
 * (relu == 0: of |x * scale + shift|) over x fp32 [P, C] — the tensor a BatchNorm + ReLU (+ bilinear upsampling) kernel is
 * about to write.  They feed the per-tensor power-of-two scales of the MX operands above. */
int asis_absmax_f32(void* stream, const float* x, int64_t rows, int cols, int64_t ld, float* amax, int reset);
int asis_bn_relu_absmax(void* stream, const float* x, const float* scale, const float* shift, int64_t P, int C, int relu, float* amax);
/* |x| maximum of a 16-bit [rows, cols] tensor (row stride ld, cols % 8 == 0) into *amax (zeroed first), and the MX plane of a
 * split-precision operand from its stored (hi, lo) 16-bit planes (same shape / leading dimension ld_in; out_mx leading dimension
 * ld_out): two e4m3 bytes per element, activations (hi8, lo8), wside != 0 (the weight operand) (lo8, hi8); amax = the
 * absolute maximum of the hi plane.  The A_lo / B_lo + mx_amax_a / mx_amax_b operands of a dense asis_gemm
 * (`dinov2/layers/block.py:89-114` linear layers at fp32 in the reference; config.precise_level 2 here). */
/* 3x3, stride 1, pad 1 convolution for NARROW outputs (Cout = 64 or 128, Cin % 64 == 0) on the operand planes of an MX split
 * convolution — x_hi / x_mx NHWC [B, H, W, Cin] (asis_bn_relu_upsample_mx), w_hi / w_mx [Cout, 9 Cin] (asis_pack_conv_weight(_mx) mode 0),
 * amax_a / amax_b the tensors' maxima — as a halo-tile kernel: one workgroup per 16 x 16 pixel tile and all output channels, the
 * 18 x 18 halo of a 64-channel chunk staged once per plane and read by the nine taps at shifted addresses (the implicit-GEMM form
 * re-stages every input element nine times per plane; with 64 / 128 output columns that fill is what bounds it).
 * out fp32 NHWC [B, H, W, Cout] (+ bias); stats != NULL: [asis_conv3x3_halo_mx_tiles(B, H, W)][2][Cout] per-tile (sum, sum of squares)
 * of the outputs, the BatchNorm partial sums asis_reduce_partials takes.  float16 only.
 * Replaces `backbones/decoders.py:109-135` conv3x3 of the last two FeatureDecoder stages (256 -> 128, 128 -> 64). */
int asis_conv3x3_halo_mx_tiles(int B, int H, int W);
int asis_conv3x3_halo_mx(void* stream, int dtype, const void* x_hi, const void* x_mx, const void* w_hi, const void* w_mx, const float* bias,
                         const float* amax_a, const float* amax_b, float* out, float* stats, int B, int H, int W, int Cin, int Cout);
int asis_absmax_16(void* stream, int dtype, const void* x, int64_t rows, int cols, int64_t ld, float* amax);
int asis_mx_from_pair(void* stream, int dtype, const void* hi, const void* lo, int64_t ld_in, void* out_mx, int64_t ld_out, int64_t rows,
                      int cols, const float* amax, int wside);
/* run-time dispatch switches of asis_gemm (same meaning as the environment variable read at first use):
 *   "p8" (ASIS_GEMM_P8): 1 = dense launches with at least one 256x256 tile per CU run on the persistent 8-phase kernel
 *   (csrc/gemm_p8.h) when K <= 2048, 2 = any K and from 16 tiles on, 3 = any K, 0 = never (one workgroup per tile,
 *   csrc/gemm_big.h); "noepi" (ASIS_GEMM_NOEPI, lab): 1 = main loops only, results are wrong.  Unknown name: ASIS_EINVAL. */
int asis_gemm_set_option(const char* name, int value);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim, fp32 statistics, eps inside sqrt, biased variance
 * (nn.LayerNorm(eps=1e-6): vision_transformer.py:89, block.py:63,75, adapter_blocks.py:114,162).
 * x fp32 [rows, D] (row stride ldx); y = out_f32 ? fp32 : dtype, row stride ldy.
 * D multiple of 4, D <= 2048.
 * ------------------------------------------------------------------------------------------- */
int asis_layernorm(void* stream, int dtype, const float* x, int64_t ldx, const float* w, const float* b, float eps,
                   void* y, int64_t ldy, int out_f32, int64_t rows, int D);
/* the same writing the 16-bit output AND its MX plane (asis_gemm_desc.mx_amax_a: two e4m3 bytes per element, (hi8, lo8)) in one
 * pass — the A operand pair of a split-precision linear layer behind a LayerNorm (`block.py:89-114` norm1 -> qkv, norm2 -> fc1 /
 * w12 at fp32 in the reference; config.precise_level 2).  amax: device float, an UPPER BOUND of |y| (e.g. sqrt(D) max|w| + max|b|:
 * the scales are powers of two and e4m3 spans 17 binades, a bound a few binades high costs nothing where it matters). */
int asis_layernorm_mx(void* stream, int dtype, const float* x, int64_t ldx, const float* w, const float* b, float eps, void* y,
                      void* y_mx, int64_t ldy, const float* amax, int64_t rows, int D);

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
/* two token batches stacked along the rows in ONE launch (images 0..B1-1 have N1 tokens, the next B2 have N2: the
 * cls + pos-embed pass and the raw patch-token pass of train.py:287,300-302): q, k, o rows are image-major in that
 * order, vt is [B1+B2, H*64, ldvt].  24 x 16 x 14 workgroups fill the chip in 10.5 rounds instead of 2 x 5.25 -> 2 x 6.
 * B2 = 0: one batch (then lse2 may be given, see below). */
int asis_attention_fwd_seg(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt, int64_t ldvt,
                           void* o, int64_t ldo, int B1, int N1, int B2, int N2, int H, float scale, float* lse2);
/* same, with an optional second output o_lo (NULL = none; layout of o): the rounding residual of the 16-bit output,
 * o_exact ~= o + o_lo with ~22 significant bits.  The projection GEMM (attention.py:67) takes it as asis_gemm_desc.A_lo:
 * rounding the attention output to 16 bits is white noise on the residual stream of every block, which spatially
 * sensitive decode heads amplify (tests/precision_probe.py: the largest single error term on the UNet / MLA logits). */
int asis_attention_fwd_split(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt, int64_t ldvt,
                             void* o, void* o_lo, int64_t ldo, int B1, int N1, int B2, int N2, int H, float scale, float* lse2);
/* same as asis_attention_fwd_split for a q that already carries scale * log2(e) (folded into the q rows of the qkv
 * projection weight and bias before their 16-bit rounding, dinov2/layers/attention.py:58-60: q' = x (c W_q)^T + c b_q):
 * the scores leave the MFMA in exp2 units and the running maximum enters the score chain as its initial accumulator, so
 * P = exp2(S') needs no per-element multiply-add (-5 % kernel time).  lse2 as below. */
int asis_attention_fwd_prescaled(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt, int64_t ldvt,
                                 void* o, void* o_lo, int64_t ldo, int B1, int N1, int B2, int N2, int H, float* lse2);
/* the same attention with V row-major: q, k, v are three column blocks of ONE projection output [tokens, ld]
 * (attention.py:58 `qkv = self.qkv(x)`), no transposed copy of V; prescaled != 0: q already carries scale * log2(e) (scale
 * ignored), as in asis_attention_fwd_prescaled.  o_lo / lse2 as above. */
int asis_attention_fwd_qkv(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ld, void* o, void* o_lo,
                           int64_t ldo, int B1, int N1, int B2, int N2, int H, float scale, int prescaled, float* lse2);
/* asis_attention_fwd_qkv with the second output plane in the MX form (two fp8 bytes per element, activation side; the A_lo + MX
 * operand of the projection GEMM, asis_gemm_desc.mx_amax_a): amax = device float, an upper bound of |o| — the attention output
 * is a convex combination of V rows, so max |v| (asis_absmax_16 over the v columns) is one.  Saves the absmax + conversion passes
 * over (o, o_lo) of the precise_level-2 blocks (round 5). */
int asis_attention_fwd_qkv_mx(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ld, void* o, void* o_mx,
                              int64_t ldo, int B1, int N1, int B2, int N2, int H, float scale, int prescaled, float* lse2,
                              const float* amax);
/* same, also writing lse2[B,H,N] = log2 sum_k exp2(log2(e) * scale * q.k) per query (what asis_attention_bwd_rows
 * needs to rebuild the probabilities); lse2 NULL = asis_attention_fwd */
int asis_attention_fwd_lse(void* stream, int dtype, const void* q, const void* k, int64_t ldqk, const void* vt,
                           int64_t ldvt, void* o, int64_t ldo, int B, int H, int N, float scale, float* lse2);
/* 16-bit [B, N, ld] (first C columns) -> [B, C, ldt], token index contiguous, columns N..ldt-1 zero
 * (the V^T / K^T / Q^T / dO^T operand layout of the attention kernels).  C % 64 == 0, ldt % 64 == 0, ldt >= N. */
int asis_transpose_tokens(void* stream, int dtype, const void* src, int64_t ld, void* dst, int64_t ldt, int B, int N,
                          int C);
/* Backward of the fused attention (the autograd transpose of attention.py:60-66):
 *   dV = P^T dO, dS = scale * P * (dO V^T - D), D = rowsum(dO * O), dQ = dS K, dK = dS^T Q,
 * scores recomputed from q, k and lse2.  q, k, v: 16-bit [tokens, >= H*64] row-major, row stride ld; o, dO: forward output and
 * its gradient; dq, dk, dv: 16-bit outputs, row stride lddq (e.g. the three column blocks of one [tokens, 3*H*64] buffer, ready
 * for the qkv weight-gradient / input-gradient GEMMs).  Row-major operands only: the products that reduce over the token index
 * read the row-major K / Q / dO tiles through transposing LDS reads (rounds 1-4 took transposed images of q, k, dO).  Two
 * stacked token batches per launch, laid out as in asis_attention_fwd_split: B1 images of N1 tokens followed by B2 images of N2 tokens (B2 = 0: one batch) in
 * every [tokens, *] operand; lse2 and the scratch D are [B1, H, N1] followed by [B2, H, N2] (D receives -scale * rowsum(dO * O),
 * the initial accumulator of the dP chains).  Pipelined LDS-DMA kernels (csrc/attn_bwd_pipe.hip); bit-reproducible. */
int asis_attention_bwd_rows(void* stream, int dtype, const void* q, const void* k, const void* v, int64_t ld, const void* o,
                            int64_t ldo, const void* dO, int64_t lddo, const float* lse2, float* D, void* dq, void* dk,
                            void* dv, int64_t lddq, int B1, int N1, int B2, int N2, int H, float scale);

/* ---------------------------------------------------------------------------------------------
 * Patch-embed im2col (patch_embed.py:75: Conv2d k=s=P) : img fp32 NCHW [B,3,Himg,Wimg] ->
 * A 16-bit [B*(Himg/P)*(Wimg/P), ldk], k = c*P*P + i*P + j, columns >= 3*P*P zero-filled.
 * ------------------------------------------------------------------------------------------- */
int asis_im2col_patch(void* stream, int dtype, const float* img, int B, int Himg, int Wimg, int P, void* out,
                      int64_t ldk);
/* same, also writing the 16-bit rounding residuals (img ~= out + out_lo) for the split-precision patch embedding: the
 * rounding of pixels and patch weights to 16 bits is the largest single error term of the whole step on the features
 * (2.7e-4 of 3.3e-4, tests/precision_probe.py) and the conv is 0.07 % of its FLOPs; out_lo NULL = asis_im2col_patch */
int asis_im2col_patch_split(void* stream, int dtype, const float* img, int B, int Himg, int Wimg, int P, void* out,
                            void* out_lo, int64_t ldk);

/* fp32 -> 16-bit cast (x scale) with optional zero-padded columns: src [rows, cols] (ld_src) ->
 * dst [rows, ld_dst], columns cols..ld_dst-1 are written as zero.  Used to pack weights once.
 * part = 0: hi = (dtype)v;  part = 1: lo = (dtype)(v - (float)hi)  — the two halves of a split-precision
 * operand: v ~= hi + lo to ~22 bits.  The conv chains (encoder, decoder forward) run
 * A_hi W_hi + A_lo W_hi + A_hi W_lo as three asis_gemm passes (res = previous pass) because five 16-bit
 * conv layers in a row exceed the 1e-3 logits tolerance on their own (Numerics). */
int asis_cast_pad(void* stream, int dtype, const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows,
                  int cols, float scale, int part);

/* tokens_A[b, 0, :] = cls + pos[0];  tokens_A[b, 1+t, :] = x[b, t, :] + pos[1+t]
 * (vision_transformer.py:196-197 with the interpolated pos-embed cached per (H,W)). fp32. */
int asis_add_cls_pos(void* stream, const float* x, const float* cls, const float* pos, float* out, int B, int N,
                     int D);

/* ---------------------------------------------------------------------------------------------
 * Row / elementwise backward kernels of the DINOv2 block (dinov2/layers/block.py:89-114).
 * asis_rowblock_nblk(rows): rows of partial sums the row kernels below write.
 * LayerNorm backward (nn.LayerNorm, eps inside sqrt, biased variance — statistics recomputed from x):
 *   dx[r] = (res ? res[r] : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w      (fp32, D <= 2048, D % 4 == 0)
 *   partial[nblk][2][D]: [0] = sum_r dy * xhat (d weight), [1] = sum_r dy (d bias); sum with asis_reduce_rows.
 * asis_gelu16: erf-GELU on 16-bit operands (mlp.py:35): dpost NULL -> out = gelu(pre); else out = dpost * gelu'(pre).
 * asis_colsum: partial[nblk][C] column sums of a 16-bit or fp32 (dtype ASIS_F32) [rows, C] matrix (bias gradients).
 * asis_ls_linear_finish: for out = x + gamma * (A W^T + b) (LayerScale o Linear, block.py:112-113) given
 *   G = dout^T A (asis_wgrad of the UNSCALED dout, fp32 [N,K]) and cs = colsum(dout):
 *   dW = gs*gamma*G, db = gs*gamma*cs, dgamma = gs*(rowsum(W*G) + b*cs); gamma NULL: dW = gs*G, db = gs*cs.
 * ------------------------------------------------------------------------------------------- */
int asis_rowblock_nblk(int64_t rows);
int asis_layernorm_bwd(void* stream, const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w, float eps,
                       const float* res, int64_t ldr, float* dx, int64_t lddx, float* partial, int64_t rows, int D);
int asis_gelu16(void* stream, int dtype, const void* pre, const void* dpost, void* out, int64_t n);
/* erf-GELU of an fp32 pre-activation into a split-precision operand pair: hi = 16-bit gelu(x), lo (optional) = the rounding
 * residual of hi (the MaskTransformer head's blocks run split precision end to end, backbones/masktrans_block.py). n % 4 == 0 */
int asis_gelu_split(void* stream, int dtype, const float* x, void* hi, void* lo, int64_t n);
/* SwiGLU gate backward (swiglu_ffn.py:30-34): x12 fp32 [R, 2*Hd] = [x1 | x2], dh 16-bit [R, Hd] = d(silu(x1) * x2)
 * -> dx12 16-bit [R, 2*Hd] = [d x1 | d x2] */
int asis_swiglu_bwd(void* stream, int dtype, const float* x12, const void* dh, void* dx12, int64_t R, int Hd);
int asis_colsum(void* stream, int dtype, const void* x, int64_t ld, float* partial, int64_t rows, int C);
/* fused: out (16-bit, row stride ldo) = scale * x (fp32 [rows, D], D <= 2048) and partial[asis_rowblock_nblk(rows)][D] =
 * column sums of x — the GEMM operand of a residual-stream gradient plus its bias / LayerScale sums in one pass */
int asis_cast_colsum(void* stream, int dtype, const float* x, int64_t ldx, void* out, int64_t ldo, float scale,
                     float* partial, int64_t rows, int D);
int asis_ls_linear_finish(void* stream, const float* G, const float* W, const float* bias, const float* gamma,
                          const float* cs, float grad_scale, float* dW, float* db, float* dgamma, int N, int K);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention core (backbones/ops/modules/ms_deform_attn.py:33-54 and the
 * location / softmax arithmetic of MSDeformAttn.forward :155-166), forward.
 * value: 16-bit [B, Lin, M*Dh] (output of value_proj; head m at columns m*Dh..)
 * offaw: fp32 [B*Lq, ld_offaw]: columns [0, M*L*P*2) = sampling_offsets(query) in the reference's
 *        (m, l, p, xy) order, columns [M*L*P*2, M*L*P*3) = attention_weights(query) logits (m, l, p)
 * ref:   fp32 [Lq, 2] reference points (x, y) in [0,1] (one reference level, broadcast over L:
 *        adapter_blocks.py:9-22);  shapes int32 [L,2] = (H_l, W_l);  starts int32 [L]
 * out:   16-bit [B*Lq, M*Dh]  (input of output_proj).   Dh % 8 == 0, L*P <= 16.
 * ------------------------------------------------------------------------------------------- */
int asis_msda_fwd(void* stream, int dtype, const void* value, const float* offaw, int64_t ld_offaw, const float* ref,
                  const int32_t* shapes, const int32_t* starts, void* out, int B, int Lq, int Lin, int M, int L, int P,
                  int Dh);
/* same, also writing the 16-bit rounding residuals of the sampled rows (out_lo, laid out like out; NULL = asis_msda_fwd): the split
 * A operand of output_proj for checkpoints / operand types whose adapters need 16 significant bits (bf16 at 1e-3) */
int asis_msda_fwd_split(void* stream, int dtype, const void* value, const float* offaw, int64_t ld_offaw, const float* ref,
                        const int32_t* shapes, const int32_t* starts, void* out, void* out_lo, int B, int Lq, int Lin, int M,
                        int L, int P, int Dh);

/* DWConv 3x3 depthwise (pad 1, bias) over the token grids of each pyramid level + erf GELU
 * (backbones/adapter_blocks.py:67-80,95-97).  x fp32 [B, Ntok, C]; w9 fp32 [9][C]
 * (w9[kh*3+kw][c] = weight[c,0,kh,kw]); shapes/starts describe the L grids; out 16-bit. */
int asis_dwconv_gelu(void* stream, int dtype, const float* x, const float* w9, const float* bias,
                     const int32_t* shapes, const int32_t* starts, int L, void* out, int B, int Ntok, int C);

/* ---------------------------------------------------------------------------------------------
 * Adapter backward (`train_adapters` mode; the autograd transposes of asis_msda_fwd and asis_dwconv_gelu).
 * asis_msda_bwd: dout fp32 [B*Lq, M*Dh] = d(sampled output) -> dvalue fp32 [B, Lin, M*Dh] (scatter form: an LDS tile
 *   per channel chunk, or fp32 global atomics into a caller-zeroed buffer when it does not fit; summation order not
 *   fixed — the GEMM form below is the deterministic, faster default of the Python layer) and doffaw fp32
 *   [B*Lq, ld_offaw] in the layout of offaw (d offsets, then d logits with the softmax over L*P already applied).
 *   M <= 32, L*P <= 16, M*Dh <= 2048.
 * asis_dwconv_gelu_bwd: x (input of the depthwise conv, fp32 [B, Ntok, C]), dy fp32 = d(GELU output) ->
 *   g fp32 scratch [B, Ntok, C] (= d pre-activation), partial[asis_dwconv_bwd_nblk(B*Ntok)][10][C] (rows 0..8: d w9[tap],
 *   row 9: d bias; sum with asis_reduce_rows) and dx 16-bit [B, Ntok, C] = d x.  C = 4*2^k <= 1024.
 * ------------------------------------------------------------------------------------------- */
int asis_msda_bwd(void* stream, int dtype, const void* value, const float* offaw, int64_t ld_offaw, const float* ref,
                  const int32_t* shapes, const int32_t* starts, const float* dout, float* dvalue, float* doffaw, int B,
                  int Lq, int Lin, int M, int L, int P, int Dh);
/* d value as a GEMM: ST[b, m, pix, q] (16-bit, q contiguous with row stride ldt >= Lq, zeroed by the caller) = the
 * attention-weighted bilinear sampling weight of query q on pixel pix for head m; then
 *   d value[b, :, m*Dh:(m+1)*Dh] = ST[b, m] (Lin x Lq) . d out[b, :, m*Dh:(m+1)*Dh] (Lq x Dh)      (asis_gemm, batch B per head).
 * asis_msda_bwd with dvalue == NULL then only produces doffaw. */
/* d value as a gather over taps bucketed by destination pixel (csrc/adapter_bwd.hip; replaces the dense sampling matrix, its
 * memset and the batched GEMMs): dout16 = 16-bit copy of d out [B*Lq, M*Dh], amax = device float >= max |d out| (asis_absmax_f32);
 * workspaces cnt int32 [B*M*Lin], offs int32 [B*M*(Lin+1)], rec 8-byte records [B*M*asis_msda_vgrad_cap(Lq, L, P)];
 * dvalue fp32 [B, Lin, M*Dh], every element written.  Bitwise reproducible (64-bit fixed-point sums). */
int asis_msda_vgrad_cap(int Lq, int L, int P);
int asis_msda_value_grad(void* stream, int dtype, const float* offaw, int64_t ld_offaw, const float* ref, const int32_t* shapes,
                         const int32_t* starts, const void* dout16, const float* amax, int32_t* cnt, int32_t* offs, void* rec,
                         float* dvalue, int B, int Lq, int Lin, int M, int L, int P, int Dh);
int asis_msda_sampling_matrix(void* stream, int dtype, const float* offaw, int64_t ld_offaw, const float* ref,
                              const int32_t* shapes, const int32_t* starts, void* ST, int64_t ldt, int B, int Lq, int Lin,
                              int M, int L, int P);
int asis_dwconv_bwd_nblk(int64_t rows);
int asis_dwconv_gelu_bwd(void* stream, int dtype, const float* x, const float* w9, const float* bias, const int32_t* shapes,
                         const int32_t* starts, int L, const float* dy, float* g, float* partial, void* dx, int B, int Ntok,
                         int C);

/* ---------------------------------------------------------------------------------------------
 * CNN encoder / decoder companions (backbones/encoders.py:9-47, backbones/decoders.py:109-135).
 * BatchNorm is in TRAIN mode everywhere on this path (batch statistics):
 *   conv (asis_gemm conv=1, fp32 out, per-tile stats) -> asis_reduce_partials -> [all-reduce of
 *   the 2C sums across ranks = SyncBatchNorm] -> asis_bn_finalize -> fused apply kernel.
 * ------------------------------------------------------------------------------------------- */
/* stem conv, Cin = 3: img fp32 NCHW [B,3,H,W], w fp32 [Cout,3,3,3] -> out fp32 NHWC [B,OH,OW,Cout] */
int asis_conv3x3_c3(void* stream, const float* img, const float* w, float* out, int B, int H, int W, int Cout,
                    int stride, int pad);
/* Direct fp32 3x3 conv (stride 1, pad 1) for layers with <= 16 output channels (the decode heads' final
 * classifier conv, decoders.py:135 / :80): x = x_hi (+ x_lo, optional split half) 16-bit NHWC [B,H,W,Cin],
 * w fp32 [Cout,Cin,3,3] (the parameter itself), out fp32 NHWC [B,H,W,Cout].  Cin in {8,16,32,64}. */
int asis_conv3x3_smallcout_fwd(void* stream, int dtype, const void* x_hi, const void* x_lo, const float* w,
                               const float* bias, float* out, int B, int H, int W, int Cin, int Cout);
/* its input gradient: dy = dy_hi (+ dy_lo) 16-bit [B,H,W,CoP] (CoP >= 8, first Cout channels valid, Cout <= 8)
 * -> dx fp32 NHWC [B,H,W,Cin] */
int asis_conv3x3_smallcout_dgrad(void* stream, int dtype, const void* dy_hi, const void* dy_lo, int CoP, const float* w,
                                 float* dx, int B, int H, int W, int Cin, int Cout);

/* The same classifier conv and its weight gradient with the preceding BatchNorm + ReLU + bilinear x2 upsampling (align_corners = True)
 * evaluated ON LOAD (`decoders.py:131-135`: BatchNorm2d, ReLU, Upsample(2), Conv2d(64, classes, 3, padding = 1)): raw = the previous
 * stage's fp32 conv output NHWC [B, H, W, 64], scale / shift = its BatchNorm affine (asis_bn_finalize); the convolution runs on the
 * [2H, 2W] map, whose halo tiles are computed from `raw` while they are staged — the upsampled operand planes (4x the bytes of
 * `raw`, written by asis_bn_relu_upsample and read back twice) never exist.  Same arithmetic as asis_bn_relu_upsample followed by
 * asis_conv3x3_smallcout_fwd / _wgrad.  out fp32 [B, 2H, 2W, Cout]; dy 16-bit [B, 2H, 2W, CoP = 8]. */
int asis_conv3x3_smallcout_fwd_up(void* stream, int dtype, const float* raw, const float* scale, const float* shift, const float* w,
                                  const float* bias, float* out, int B, int H, int W, int Cin, int Cout);
int asis_conv3x3_smallcout_wgrad_up(void* stream, int dtype, const void* dy, int CoP, const float* raw, const float* scale,
                                    const float* shift, float* slabs, int nblk, int B, int H, int W, int Cin, int Cout);

/* Weight gradient of the same 3x3 / stride 1 / pad 1 classifier conv (`backbones/decoders.py:135` under
 * `loss.backward()`, `train.py:432`): dy 16-bit [B,H,W,CoP], x 16-bit [B,H,W,Cin] -> `nblk` fp32 slab rows of
 * [Cout,Cin,3,3] partial sums (one per workgroup; sum them with asis_reduce_rows).  Cin in {8,16,32,64}. */
int asis_conv3x3_smallcout_wgrad(void* stream, int dtype, const void* dy, int CoP, const void* x, float* slabs, int nblk,
                                 int B, int H, int W, int Cin, int Cout);
/* column sums / sums of squares of fp32 [R, C] -> partial[nparts][2][C], nparts = asis_colstats_nparts(R) */
int asis_colstats_nparts(int64_t R);
int asis_colstats(void* stream, const float* x, int64_t R, int C, float* partial);
/* partial[nparts][2][C] (fp32) -> sums[2][C] (double) */
int asis_reduce_partials(void* stream, const float* partial, int nparts, int C, double* sums);
/* scale = gamma*invstd, shift = beta - mean*scale; optional running-stat update (momentum, unbiased var)
 * and num_batches_tracked += 1 (nn.BatchNorm2d / SyncBatchNorm train-mode semantics). */
int asis_bn_finalize(void* stream, const double* sums, double count, int C, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* scale, float* shift, float* mean_out, float* invstd_out);
/* eval-mode BatchNorm: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale (train.py:451) */
int asis_bn_eval_affine(void* stream, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, int C, float* scale, float* shift);
/* out(16-bit) = [relu](x*scale + shift), x fp32 [R, C].  Every fused apply kernel below optionally
 * also writes out_lo (NULL = skip): the rounding residual of out, second half of a split operand. */
int asis_bn_act(void* stream, int dtype, const float* x, const float* scale, const float* shift, int relu, void* out,
                void* out_lo, int64_t R, int C);
/* the same with the lo output in the MX form (asis_gemm_desc.mx_amax_a / asis_conv3x3_halo_mx); amax = device float, the tensor's
 * absolute maximum (asis_bn_relu_absmax) */
int asis_bn_act_mx(void* stream, int dtype, const float* x, const float* scale, const float* shift, int relu, void* out, void* out_mx,
                   const float* amax, int64_t R, int C);
/* BN + ReLU + MaxPool2d(3, stride 2, pad 1): x fp32 NHWC [B,H,W,C] -> 16-bit [B,OH,OW,C] (encoders.py:19) */
int asis_bn_relu_maxpool(void* stream, int dtype, const float* x, const float* scale, const float* shift, void* out,
                         void* out_lo, int B, int H, int W, int C);
/* BN + ReLU + bilinear upsample xfactor, align_corners=True (decoders.py:112-113; MLAHead :38-45 with factor 4) */
int asis_bn_relu_upsample(void* stream, int dtype, const float* x, const float* scale, const float* shift, void* out,
                          void* out_lo, int B, int H, int W, int C, int factor);
/* the same with the lo output in the MX form (asis_gemm_desc.mx_amax_a / mx_amax_b); amax = device float, the tensor's absolute maximum */
int asis_bn_relu_upsample_mx(void* stream, int dtype, const float* x, const float* scale, const float* shift, void* out,
                             void* out_mx, const float* amax, int B, int H, int W, int C, int factor);
/* conv weight fp32 [Cout,Cin,KH,KW] -> 16-bit GEMM operand.
 * mode 0 (forward): out[co][(kh*KW+kw)*Cin+ci], rows Cout.
 * mode 1 (dgrad):   out[ci][((KH-1-kh)*KW+(KW-1-kw))*CoP+co], rows Cin, CoP = Cout rounded up to 8. */
int asis_pack_conv_weight(void* stream, int dtype, const float* w, void* out, int Cout, int Cin, int KH, int KW,
                          int mode, int64_t ldo, int part);
/* the same with the lo output in the MX form (asis_gemm_desc.mx_amax_a / mx_amax_b); amax = device float, the tensor's absolute maximum */
int asis_pack_conv_weight_mx(void* stream, int dtype, const float* w, void* out_mx, int Cout, int Cin, int KH, int KW,
                             int mode, int64_t ldo, const float* amax);
/* both operands of a split convolution in ONE pass over the fp32 weight (round 5): out_hi = asis_pack_conv_weight(part 0),
 * out_lo = its rounding residual (amax NULL) or its MX form (amax = the weight's absolute maximum, device float) */
int asis_pack_conv_weight_pair(void* stream, int dtype, const float* w, void* out_hi, void* out_lo, int Cout, int Cin, int KH,
                               int KW, int mode, int64_t ldo, const float* amax);
/* decoder input (train.py:389-406): [xs | zero-padded c4 | vit] fp32 tokens -> 16-bit NHWC [B,h,w,3D];
 * every source has its own batch stride (elements) so token slices need no copies */
int asis_decoder_input(void* stream, int dtype, const float* xs, int64_t xs_bstride, const float* c4,
                       int64_t c4_bstride, const float* vit, int64_t vit_bstride, void* out, void* out_lo, int B, int h,
                       int w, int h4, int w4, int D);
/* the same with the lo output in the MX form (asis_gemm_desc.mx_amax_a / mx_amax_b); amax = device float, the tensor's absolute maximum */
int asis_decoder_input_mx(void* stream, int dtype, const float* xs, int64_t xs_bstride, const float* c4, int64_t c4_bstride,
                          const float* vit, int64_t vit_bstride, void* out, void* out_mx, const float* amax, int B, int h,
                          int w, int h4, int w4, int D);
/* SwiGLU gate (dinov2/layers/swiglu_ffn.py:30-34): x12 fp32 [R, 2*Hd] = [x1 | x2] -> out 16-bit [R, Hd] = silu(x1)*x2 */
int asis_swiglu(void* stream, int dtype, const float* x12, void* out, int64_t R, int Hd);
/* same with an optional second output out_lo (NULL = none): the rounding residual of the 16-bit result, so that (out, out_lo) is
 * a split-precision A operand of the w3 GEMM (config.precise_level 2) */
int asis_swiglu_split(void* stream, int dtype, const float* x12, void* out, void* out_lo, int64_t R, int Hd);
/* strided row copy in bytes (channel concat / split of NHWC tensors: torch.cat(dim=1), decoders.py:47) */
int asis_copy_channels(void* stream, const void* src, int64_t src_ld_bytes, void* dst, int64_t dst_ld_bytes, int64_t rows,
                       int64_t row_bytes);
/* out[b] = a[b] + b_[b] over n floats per batch element, each operand with its own batch stride
 * (fp32; train.py:320,343,365,387 residual adds with the cls-stripped pass-A features) */
int asis_add_f32(void* stream, const float* a, const float* b, float* out, int64_t n, int batch, int64_t stride_a,
                 int64_t stride_b, int64_t stride_out);

/* ---------------------------------------------------------------------------------------------
 * UNet decode head data movement (backbones/unet_parts.py:26-104; the 3x3 convs, BatchNorm and ReLU of DoubleConv are
 * asis_gemm(conv) + asis_bn_act, the 1x1 OutConv and the ConvTranspose2d products are asis_gemm).
 * MaxPool2d(2) (unet_parts.py:31-33), floor mode, on split-precision NHWC maps: x/x_lo 16-bit [B,H,W,C] (x_lo NULL =
 * single precision) -> out/out_lo [B,H/2,W/2,C] and idx uint8 [B,H/2,W/2,C] (argmax 0..3 = di*2+dj, first maximum
 * wins like ATen; optional).  Backward ACCUMULATES: dx[b,2i+di,2j+dj,c] += dy[b,i,j,c] at the argmax — dx fp32
 * [B,H,W,C] already holds the tensor's other gradient path (the skip connection) or zeros.  C % 8 == 0.
 * ------------------------------------------------------------------------------------------- */
int asis_maxpool2_fwd(void* stream, int dtype, const void* x, const void* x_lo, void* out, void* out_lo, uint8_t* idx,
                      int B, int H, int W, int C);
int asis_maxpool2_bwd(void* stream, const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C);
/* Tail of the MaskTransformer decode head (reference eval/eval_dinov2_masktrans.py:452-462): L2-normalised patch and class
 * features, their cosines, LayerNorm over the C classes (mask_norm).  Stacked token layout: batch b owns rows b*(N+C) ..
 * +N-1 (patches) and the C rows behind them (class tokens) of an fp32 [B*(N+C), D] matrix.  C <= 16, D % 4 == 0.
 *   asis_cls_l2norm      x (class rows of the stacked matrix) -> chat fp32 [B,C,D] = x / ||x||, inv_c fp32 [B*C]
 *   asis_cls_l2norm_bwd  dchat, chat, inv_c -> the class rows of dx (stacked layout; patch rows untouched)
 *   asis_mask_logits_fwd P (stacked; patch rows read), chat, mask_norm weight / bias / eps -> logits fp32 [B*N, C],
 *                        cosm fp32 [B*N, C] (the cosines), inv_p fp32 [B*N] = 1 / ||patch row||
 *   asis_mask_logits_bwd dlogits fp32 [B*N, C] -> dP (patch rows of the stacked layout), dcos fp32 [B*N, C], and
 *                        part fp32 [asis_mask_logits_nblk(B, N), 2, C]: per-workgroup sums of (dlogits * xhat | dlogits) =
 *                        mask_norm weight / bias gradients after a column sum
 *   asis_mask_dchat      dchat fp32 [B,C,D] = sum_n dcos[b,n,:]^T (P[b,n,:] inv_p[b,n])   (zeroed inside, row slices by atomics) */
int asis_cls_l2norm(void* stream, const float* x, float* chat, float* inv_c, int B, int N, int C, int D);
int asis_cls_l2norm_bwd(void* stream, const float* dchat, const float* chat, const float* inv_c, float* dx, int B, int N, int C,
                        int D);
int asis_mask_logits_fwd(void* stream, const float* P, const float* chat, const float* gamma, const float* beta, float eps,
                         float* logits, float* cosm, float* inv_p, int B, int N, int C, int D);
int asis_mask_logits_nblk(int B, int N);
int asis_mask_logits_bwd(void* stream, const float* dlogits, const float* cosm, const float* inv_p, const float* P,
                         const float* chat, const float* gamma, float eps, float* dP, float* dcos, float* part, int B, int N, int C,
                         int D);
int asis_mask_dchat(void* stream, const float* dcos, const float* P, const float* inv_p, float* dchat, int B, int N, int C, int D);
/* FCUUp + FusionModel of the OR-UNet fuse head (reference eval/eval_dinov2_or_unet_fuse.py:502-530, used at :448-464):
 *   x <- relu(x + F.interpolate(r, size=(H, W)))           (default mode 'nearest')
 * asis_nearest_add_relu: in place on the 16-bit map x (+ x_lo) [B,H,W,C]; r (+ r_lo) 16-bit [B,h,w,C]; ys int32 [H] / xs
 *   int32 [W] = source row / column of every destination row / column (ATen: min(floor(dst * float(in) / out), in - 1),
 *   computed by the caller in that float arithmetic).  C % 8 == 0.
 * asis_nearest_sum: the transpose, dr fp32 [B,h,w,C] = sum of g fp32 [B,H,W,C] over each source pixel's destination
 *   rectangle rows y0[sy] .. y0[sy+1]-1, columns x0[sx] .. x0[sx+1]-1 (y0 int32 [h+1], x0 int32 [w+1]).  C % 4 == 0.
 *   (The ReLU of the sum needs no mask of its own in the backward: both addends are post-ReLU, so wherever the sum is 0 the
 *   producers' own ReLU masks are 0 too.) */
int asis_nearest_add_relu(void* stream, int dtype, void* x, void* x_lo, const void* r, const void* r_lo, const int* ys,
                          const int* xs, int B, int H, int W, int h, int w, int C);
int asis_nearest_sum(void* stream, const float* g, float* dr, const int* y0, const int* x0, int B, int H, int W, int h, int w,
                     int C);
/* nn.ConvTranspose2d(Cin, Cout, kernel_size=2, stride=2) (unet_parts.py:50,77) as one GEMM + a pixel shuffle:
 *   G[p, co*4 + di*2 + dj] = bias[co] + sum_ci x[p, ci] * w[ci, co, di, dj],   p = (b, i, j)
 * scatter: G fp32 [B*H*W, 4*Cout] -> dst/dst_lo 16-bit [B,H2,W2,Ctot], channels [coff, coff+Cout), pixel
 *   (padT + 2i + di, padL + 2j + dj) — i.e. straight into the F.pad + torch.cat([x2, x1], dim=1) buffer of
 *   Up.forward (unet_parts.py:54-63); the caller zero-fills the pad border.
 * gather: the transpose, d cat fp32 [B,H2,W2,Ctot] -> dG/dG_lo 16-bit [B*H*W, 4*Cout].
 * bias_grad: partial[asis_convt2x2_bias_nblk(B*2H*2W)][Cout] column sums of the same slice (sum rows -> d bias).
 * Cout, Ctot, coff multiples of 8. */
int asis_convt2x2_scatter(void* stream, int dtype, const float* G, void* dst, void* dst_lo, int B, int H, int W, int Cout,
                          int H2, int W2, int Ctot, int coff, int padT, int padL);
int asis_convt2x2_gather(void* stream, int dtype, const float* dcat, void* dG, void* dG_lo, int B, int H, int W, int Cout,
                         int H2, int W2, int Ctot, int coff, int padT, int padL);
int asis_convt2x2_bias_nblk(int64_t rows);
/* input gradient of the UNet's 1x1 classifier (OutConv, `backbones/unet_parts.py:95-104` under autograd): dU fp32 [M, Cq] =
 * (d_hi + d_lo)[M, :C] w, d_hi / d_lo = the 16-bit halves of the logits' gradient (row stride ldd elements, >= C; d_lo may be
 * NULL), w fp32 [C, Cq] = the conv weight as stored, C <= 8 classes, Cq % 4 == 0, Cq <= 1024.  One pass over dU (round 5: as a
 * K = 8 GEMM in three split passes this was 1.6 % of the config-2 step). */
int asis_conv1x1_dgrad_small(void* stream, int dtype, const void* d_hi, const void* d_lo, int64_t ldd, const float* w, float* dU,
                             int64_t M, int Cq, int C);
int asis_convt2x2_bias_grad(void* stream, const float* dcat, float* partial, int B, int H, int W, int Cout, int H2, int W2,
                            int Ctot, int coff, int padT, int padL);

/* ---------------------------------------------------------------------------------------------
 * Segmentation losses, fused with the bilinear resize (h,w)->(H,W) of the logits (align_corners=False).
 * With x0 = resized logits, x1 = softmax_C(x0), x2 = softmax_C(x1):
 *   region term on q = x_{n_region} from the per-(b,c) sums I = sum q t, Sp = sum q, St = sum t:
 *     mode 0  Dice      1 - mean 2I/(Sp+St+eps)                      segloss/dice.py:22-33 (eps 1e-19)
 *     mode 1  soft IoU  mean [1 - (I+eps)/(Sp+St-I+eps)]             segloss/iou_multi.py:9-49 (eps = smooth 1e-6)
 *     mode 2  SoftDice  -mean (2I+eps)/(Sp+St+eps)                   segloss/dice_loss.py:255-291,31-81 (smooth 1)
 *     mode 3  Tversky   -mean (I+eps)/(I + .3 fp + .7 fn + eps)      segloss/dice_loss.py:333-372
 *     mode 4  none
 *   CE term (n_ce = 1 or 2; 0 = none): weighted-mean nll of log softmax(x_{n_ce-1})
 *     (segloss/ND_Crossentropy.py:11-32, nn.CrossEntropyLoss of eval/eval_dinov2_unet.py:291, train.py:616-617);
 *     ce_weight fp32 [C] or NULL.  loss = region + CE (DC_and_CE_loss, segloss/dice_loss.py:445-459).
 * Examples: train.py:422-428 = (n_region 2, mode 0); train_mla.py:385-389 = (2, mode 1);
 *   eval_dinov2_unet.py:291-297 CE + DC(2) on the raw resized logits = (n_region 1, mode 0, n_ce 1);
 *   DC_and_CE_loss()(softmaxed output) = (n_region 1, mode 2, n_ce 2).
 * logits fp32 NHWC [B,h,w,C], C <= 16, B*C <= 256; target int64 [B,H,W];
 * partial: asis_dice_nblk(H,W)*B*(C*3+2) floats; sums (optional) [B,C,3] = I, Sp, St; loss 1 float;
 * coef B*C*2+1 floats feeds the backward (already multiplied by grad_scale = the static loss scale).
 * ------------------------------------------------------------------------------------------- */
int asis_dice_nblk(int H, int W);
int asis_seg_loss_fwd(void* stream, const float* logits, const int64_t* target, const float* ce_weight, int B, int h,
                      int w, int H, int W, int C, int n_region, int mode, float eps, int n_ce, float grad_scale,
                      float* partial, float* sums, float* loss, float* coef);
/* dz fp32 [B,H,W,C] = d loss / d resized-logits (asis_resize_bilinear_bwd carries it back to (h,w)) */
int asis_seg_loss_bwd(void* stream, const float* logits, const int64_t* target, const float* coef, const float* ce_weight,
                      int B, int h, int w, int H, int W, int C, int n_region, int mode, int n_ce, float* dz);
/* region-only shorthands (mode 0 / 1, n_region = n_softmax); same buffer sizes as above */
int asis_dice_fwd(void* stream, const float* logits, const int64_t* target, int B, int h, int w, int H, int W, int C,
                  int n_softmax, float eps, int mode, float grad_scale, float* partial, float* sums, float* loss,
                  float* coef);
/* Validation metrics (train.py:616-617,642) fused with the resize: partial[asis_ce_acc_nblk(B*H*W)][3] =
 * {sum w[t]*nll, sum w[t], #(argmax == t)}; weight NULL = 1.  CE = col0/col1, accuracy = col2/(B*H*W). */
int asis_ce_acc_nblk(int64_t total_pixels);
int asis_ce_acc(void* stream, const float* logits, const int64_t* target, const float* weight, int B, int h, int w,
                int H, int W, int C, float* partial);
/* same pass + per-class pixel counts for `ch_iou` / `isi_iou` (segloss/iou_multi.py:51-88, called on the argmax of
 * the batch at train_multi_class.py:582-589): counts int32 [C][3] = #(target == c), #(argmax == c), #(both);
 * zeroed by the call; argmax ties go to the lowest class like torch.max */
int asis_ce_acc_counts(void* stream, const float* logits, const int64_t* target, const float* weight, int B, int h, int w,
                       int H, int W, int C, float* partial, int32_t* counts);
int asis_dice_bwd(void* stream, const float* logits, const int64_t* target, const float* coef, int B, int h, int w,
                  int H, int W, int C, int n_softmax, float* dz);
/* F.interpolate(x, size=(H,W), mode="bilinear") (align_corners=False): fp32 NHWC [B,h,w,C] -> [B,H,W,C], C <= 16
 * (decoders.py:88; the training loss fuses this into asis_dice_fwd instead) */
int asis_resize_bilinear_fwd(void* stream, const float* x, int B, int h, int w, int H, int W, int C, float* out);
/* transpose of F.interpolate(bilinear, align_corners=False): dz [B,H,W,C] -> 16-bit [B,h,w,CP] (CP = C
 * rounded up to 8, pad channels zero; dtype ASIS_F32: fp32 output, any CP >= C)
 * + partial[asis_resize_bwd_nblk(B*h*w)][C] column sums */
int asis_resize_bwd_nblk(int64_t total_pixels);
int asis_resize_bilinear_bwd(void* stream, int dtype, const float* dz, int B, int H, int W, int h, int w, int C, int CP,
                             void* out, void* out_lo, float* partial);
/* out[k] = scale * sum_n partial[n][k], summed in double in a fixed order */
int asis_reduce_rows(void* stream, const float* partial, int n, int K, float scale, float* out);

/* ---------------------------------------------------------------------------------------------
 * Backward of conv -> BN(train) -> ReLU -> upsample stages, weight gradients, optimizer.
 * ------------------------------------------------------------------------------------------- */
/* grid size the element-wise backward kernels use for `total_chunks` float4 chunks (partials rows) */
int asis_ew_blocks(int64_t total_chunks);
/* rows of partial sums the two BatchNorm-backward kernels below write for `rows` pixels of C channels (C % 4 == 0) */
int asis_bn_bwd_nblk(int64_t rows, int C);
/* g = relu'(bn(x)) * upsample^T(dU); partial[asis_bn_bwd_nblk(B*H*W, C)][2][C] = sum g, sum g*xhat */
int asis_upsample_bn_relu_bwd(void* stream, const float* dU, const float* x, const float* scale, const float* shift,
                              const float* mean, const float* invstd, float* g, float* partial, int B, int H, int W,
                              int C, int factor);
/* the same for the stem's BN + ReLU + MaxPool2d(3, 2, 1) (encoders.py:17-18): dy fp32 [B,OH,OW,C], x = raw conv output
 * [B,H,W,C] -> g [B,H,W,C] (gradient at the first maximum of every window, as ATen) + the same partial sums */
int asis_maxpool_bn_relu_bwd(void* stream, const float* dy, const float* x, const float* scale, const float* shift,
                             const float* mean, const float* invstd, float* g, float* partial, int B, int H, int W, int C);
/* zero-insertion for the input gradient of a stride-2 conv: out[b,2i,2j,:] = in[b,i,j,:], rest 0; 16-bit NHWC (+lo) */
int asis_dilate2(void* stream, int dtype, const void* in, const void* in_lo, void* out, void* out_lo, int B, int OH, int OW,
                 int Hd, int Wd, int C);
/* dx(16-bit) = gamma*invstd*(g - dbeta/n - xhat*dgamma/n); partial[asis_bn_bwd_nblk(R, C)][C] = sum dx.
 * out_lo (optional) = rounding residual of dx: the dgrad GEMM chain runs split-precision because the
 * mean subtraction of the next BatchNorm backward amplifies 16-bit rounding noise (Numerics). */
int asis_bn_bwd_apply(void* stream, int dtype, const float* g, const float* x, const float* mean, const float* invstd,
                      const float* gamma, const float* dgamma, const float* dbeta, double count, void* out,
                      void* out_lo, float* partial, int64_t R, int C);
/* the same with dx's lo output in the MX form (asis_gemm_desc.mx_amax_a: the input-gradient convolution then runs its two correction
 * terms as one block-scaled fp8 pass); amax = max |dx| from asis_bn_bwd_absmax (the same expression, nothing written) */
int asis_bn_bwd_apply_mx(void* stream, int dtype, const float* g, const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* dgamma, const float* dbeta, double count, void* out, void* out_mx,
                         const float* amax, float* partial, int64_t R, int C);
int asis_bn_bwd_absmax(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* gamma,
                       const float* dgamma, const float* dbeta, double count, float* amax, int64_t R, int C);

/* Weight gradient dW[Cout,Cin,KH,KW] = sum_p dy[p,co] * x[b, oh*s+kh-pad, ow*s+kw-pad, ci]
 * (conv2d; KH=KW=1 gives the nn.Linear weight grad).  dy 16-bit [P, ld_dy] with CoP (multiple of 8)
 * valid-or-zero channels; x 16-bit NHWC [B,H,W,Cin].  out: `splits` fp32 slabs of Cout*Cin*KH*KW,
 * to be summed with asis_reduce_rows(out, splits, Cout*Cin*KH*KW, 1/loss_scale, grad). */
typedef struct asis_wgrad_desc {
  const void* dy;
  const void* x;
  float* out;
  int64_t ld_dy;
  int64_t P;
  int32_t dtype;
  int32_t Cout, CoP, Cin;
  int32_t B_, H, W, OH, OW, KH, KW, stride, pad;
  int32_t splits;
  int64_t k_per_split; /* filled by asis_wgrad */
} asis_wgrad_desc;
int asis_wgrad_splits(int64_t P, int Cout, int Ntot);
int asis_wgrad(void* stream, const asis_wgrad_desc* d);
/* Weight gradient of a 3x3 / stride 1 / pad 1 convolution with Cout % 64 == 0 and Cin % 128 == 0 on halo tiles (round 5,
 * csrc/convwgrad.hip; the narrow decoder stages of backbones/decoders.py:109-135, whose 64 / 128 output channels leave the
 * implicit-GEMM form of asis_wgrad at 0.12 / 0.19 matrix-pipe busy): dy 16-bit [B,H,W,ld_dy] (first Cout channels), x 16-bit
 * [B,H,W,Cin] -> slabs fp32 [nblk, Cout*Cin*9], each row a partial dW in the parameter's [Cout][Cin][3][3] layout, to be summed
 * by asis_reduce_rows (fixed order: deterministic).  nblk = asis_conv3x3_wgrad_halo_nblk(...) workgroup columns (x (Cin / 128)
 * (Cout / 64) channel-block combinations = one workgroup per CU). */
int asis_conv3x3_wgrad_halo_nblk(int B, int H, int W, int Cin, int Cout);
int asis_conv3x3_wgrad_halo(void* stream, int dtype, const void* dy, int64_t ld_dy, const void* x, float* slabs, int nblk, int B,
                            int H, int W, int Cin, int Cout);

/* torch.optim.SGD step (train.py:178-191: momentum, weight decay, dampening 0, no Nesterov) on a flat
 * fp32 parameter buffer; g is multiplied by inv_scale (1/loss_scale) first. */
int asis_sgd_momentum(void* stream, float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                      float weight_decay, float inv_scale, int first_step);
/* Training-time augmentation on the device (train.py:139-163 runs albumentations on uint8 images in the DataLoader
 * workers, tools/dataset.py:150-161 converts to float / 255): img uint8 [B,S,S,3], mask uint8 [B,S,S] ->
 * out fp32 [B,3,S,S] in [0,1], mask_out int64 [B,S,S].  Per sample: crop + resize back to S x S (OpenCV 8-bit INTER_LINEAR
 * fixed-point arithmetic; mask INTER_NEAREST), horizontal flip, np.rot90 by rotk, one 256-entry look-up table
 * (brightness/contrast then gamma).  Tables are built by the host (adaptersis_amd/tools/augment.py):
 *   geo int32 [B,4] = {flip, rotk, identity (no crop), 0}; xofs, yofs int32 [B,S] left / top source index;
 *   xa, ya int16 [B,S,2] the two 11-bit coefficients; mx, my int32 [B,S] nearest source index; lut uint8 [B,256]. */
int asis_augment(void* stream, const uint8_t* img, const uint8_t* mask, const int32_t* geo, const int32_t* xofs,
                 const int32_t* yofs, const int16_t* xa, const int16_t* ya, const int32_t* mx, const int32_t* my,
                 const uint8_t* lut, float* out, int64_t* mask_out, int B, int S);
/* CLAHE stage of the same pipeline (train.py:161 A.CLAHE(p=0.8); csrc/augment.hip).  A batch with CLAHE samples runs
 *   asis_augment_geo_u8: the geometric stage of asis_augment alone -> uint8 RGB [B,S,S,3] + the final int64 mask;
 *   asis_clahe: clahe int32 [B,2] = (apply flag, integer clip limit of CLAHE_Impl::apply) per sample; the five look-up tables
 *     of OpenCV's initLabTabs (device: gamma u16[256], cbrt u16[3072], l2yf u16[256*2], ab2xz i32[36864], invgamma u8[4096])
 *     and the two 3x3 12-bit matrices of RGB2Lab_b / Lab2RGBinteger (HOST int32[9] each); luts = workspace u8 [B,tiles,tiles,256];
 *     lut u8 [B,256] = brightness/contrast + gamma table of asis_augment; out fp32 [B,3,S,S] in [0,1].  RGB -> Lab, tiled
 *     contrast-limited equalisation of L (tiles x tiles grid, reflect-101 padding), Lab -> RGB, lut, / 255 — integer / table
 *     arithmetic, bit-identical to the numpy restatement oracle/augment_ref.py. */
int asis_augment_geo_u8(void* stream, const uint8_t* img, const uint8_t* mask, const int32_t* geo, const int32_t* xofs,
                        const int32_t* yofs, const int16_t* xa, const int16_t* ya, const int32_t* mx, const int32_t* my,
                        uint8_t* out_u8, int64_t* mask_out, int B, int S);
int asis_clahe(void* stream, const uint8_t* rgb, const int32_t* clahe, const uint16_t* tab_gamma, const uint16_t* tab_cbrt,
               const uint16_t* tab_l2yf, const int32_t* tab_ab2xz, const uint8_t* tab_invgamma, const int32_t* coef_fwd,
               const int32_t* coef_inv, uint8_t* luts, const uint8_t* lut, float* out, int B, int S, int tiles);

/* ---------------------------------------------------------------------------------------------
 * Dropout of the MaskTransformer decode head (backbones/masktrans_block.py:11-89: nn.Dropout(p) on the attention probabilities,
 * the projection output, behind GELU and behind fc2; eval_dinov2_masktrans.py:136-139 builds it with p = 0.1).  csrc/dropout.hip.
 * Counter-based masks: keep(seed, site, i) = Philox4x32-10(key = seed, counter = (i / 4, site))[i % 4] >= p * 2^32 — a pure
 * function of (seed, dropout-layer number, element index); forward, backward and the mask export regenerate it.
 *   asis_dropout_f32:  out = (res ? res : 0) + (alpha * x + bias_n[col]) * keep / (1 - p), fp32 [n / ncols, ncols] contiguous
 *                      (bias_n NULL: no affine term beyond alpha); n % 4 == 0.
 *   asis_dropout_t16:  16-bit in place: x *= keep (rescale != 0: also / (1 - p)); x_lo (optional rounding-residual half of a
 *                      split operand, rescale must be 0) is zeroed where x is.
 *   asis_dropout_mask: uint8 [n] keep flags (what the oracle replays).
 *   asis_softmax_dropout_fwd: scores S fp32 [rows, ld] (columns >= N padding) -> p16 = softmax(scale * S), pd16 = p16 * keep /
 *                      (1 - p), 16-bit [rows, ld], padding 0; mask index = row * ld + column.
 *   asis_softmax_dropout_bwd: ds16 = scale * P * (keep / (1 - p) * dPd - sum_k pd16 dPd), the gradient of S.
 * ------------------------------------------------------------------------------------------- */
int asis_dropout_f32(void* stream, const float* x, const float* res, float* out, int64_t n, uint64_t seed, int site, float p,
                     float alpha, const float* bias_n, int ncols);
int asis_dropout_t16(void* stream, int dtype, void* x, void* x_lo, int64_t n, uint64_t seed, int site, float p, int rescale);
int asis_dropout_mask(void* stream, uint8_t* out, int64_t n, uint64_t seed, int site, float p);
int asis_softmax_dropout_fwd(void* stream, int dtype, const float* S, void* p16, void* pd16, int64_t rows, int N, int ld, float scale,
                             uint64_t seed, int site, float p);
int asis_softmax_dropout_bwd(void* stream, int dtype, const void* p16, const void* pd16, const float* dPd, void* ds16, int64_t rows,
                             int N, int ld, float scale, uint64_t seed, int site, float p);

/* Overflow guard for the static loss scale of the 16-bit gradient tensors (the reference trains in fp32 and has no
 * counterpart; torch.cuda.amp.GradScaler.step has the same skip semantics).  guard = int32[2] in device memory:
 *   asis_grad_guard: guard[0] |= (any element of g is inf / NaN); reset != 0 clears guard[0] first (call once per step
 *   with reset = 1 on the first bucket, reset = 0 on the others);
 *   asis_sgd_momentum_guarded: asis_sgd_momentum, except that a step with guard[0] != 0 changes nothing and, when
 *   count_skip != 0 (set it for ONE bucket of the step), adds 1 to guard[1] = the count of skipped steps, read by the host
 *   whenever it likes. */
int asis_grad_guard(void* stream, const float* g, int64_t n, int32_t* guard, int reset);
int asis_sgd_momentum_guarded(void* stream, float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                              float weight_decay, float inv_scale, int first_step, int32_t* guard, int count_skip);
int asis_scale_f32(void* stream, float* x, int64_t n, float a);
/* zero `bytes` bytes at p on `stream` (hipMemsetAsync: the DMA fill, ~6 TB/s; the step's few accumulate-into buffers) */
int asis_zero(void* stream, void* p, int64_t bytes);
/* 16-bit transport form of a gradient range for the data-parallel all-reduce (replaces nothing in the reference: DDP's
 * bf16 compression hook `torch.distributed.algorithms.ddp_comm_hooks.default_hooks.bf16_compress_hook` is the torch-side
 * counterpart; train.py:84-116 wraps its modules in plain fp32 DDP).  g fp32 [n] <-> out bf16 [n], n % 4 == 0, RNE. */
int asis_grad_pack_bf16(void* stream, const float* g, int64_t n, void* out);
int asis_grad_unpack_bf16(void* stream, const void* in, int64_t n, float* g);

#ifdef __cplusplus
}
#endif
#endif /* ASIS_HIP_H */
