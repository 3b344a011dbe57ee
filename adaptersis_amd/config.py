"""Process-wide numerics configuration.

``operand_dtype`` is the 16-bit MFMA operand type of activations and weights.  float16 is the
default: it is the reference's own GPU autocast dtype (`dinov2/configs/ssl_default_config.yaml:9`,
`dinov2/eval/setup.py:52-59`), it runs at the same MFMA rate as bfloat16 on gfx950, and it is the
only 16-bit type that keeps 48 frozen block evaluations within the 1e-3 logits tolerance
(bf16 operand rounding measures 3-5e-3).  ``ASIS_OPERAND=bf16`` selects
bfloat16.  Accumulators, the residual stream, normalisation statistics, softmax and losses are
fp32 in both modes.

``loss_scale`` multiplies dL/dlogits in the backward pass of 16-bit gradient tensors and is
divided out exactly (power of two) in the optimizer kernel; Dice gradients are ~1e-7 per pixel,
below fp16's normal range.
"""
import os

import torch

operand_dtype = torch.bfloat16 if os.environ.get("ASIS_OPERAND", "f16").lower() in ("bf16", "bfloat16") else torch.float16
loss_scale = float(os.environ.get("ASIS_LOSS_SCALE", 65536.0 if operand_dtype == torch.float16 else 1.0))


def set_operand_dtype(dt: torch.dtype) -> None:
    global operand_dtype
    if dt not in (torch.float16, torch.bfloat16):
        raise ValueError("operand dtype must be float16 or bfloat16")
    operand_dtype = dt

# Split-precision (hi + lo 16-bit halves, three MFMA passes) for the forward convolutions of the
# CNN encoder and the decode head: five plain 16-bit conv+BN layers in a row put ~1.2-1.7e-3 on the
# logits by themselves (measured), above north_star's 1e-3; split operands remove that term for
# ~7 % more FLOPs per step.  ASIS_SPLIT_CONV=0 disables it.
split_conv = os.environ.get("ASIS_SPLIT_CONV", "1") != "0"

# Opt-in precision mode for the attention branch of the ViT blocks (ASIS_PRECISE=1): the qkv and proj WEIGHTS enter their
# GEMMs as hi + lo 16-bit halves (one extra MFMA pass over the weight correction).  tests/precision_probe.py: rounding the
# weights to 16 bits is the largest error term on the adapter stream — it is the same for every token, so it adds up
# coherently through the blocks — and qkv + proj carry 55 % of its energy.  Off by default: +33 % linear-layer FLOPs.
precise_attention = os.environ.get("ASIS_PRECISE", "0") not in ("0", "")
# forward-only attention (frozen trunk, `Attention.attend_rows`): softmax scale * log2(e) folded into the q rows of the qkv
# projection weight / bias before their 16-bit rounding, attention through asis_attention_fwd_prescaled (ASIS_FOLD_SCALE=0:
# the unfolded kernel).  The training forward / backward keep the unfolded q (the backward rebuilds P from q, k, scale).
fold_attn_scale = os.environ.get("ASIS_FOLD_SCALE", "1") not in ("0", "")
# forward-only attention: q | k | v from ONE projection GEMM and V read row-major by the attention kernel (transposing LDS
# reads, asis_attention_fwd_qkv) instead of the batched V^T GEMMs.  OFF by default: with every launch on one stream it saves
# 1.0 ms per step (V^T GEMMs -3.5 ms, wider projection +1.9, attention +3 %), but the V^T GEMMs already hide almost completely
# on their side stream -- same-box A/B of the default command: 201.7 -> 187.5 img/s.  ASIS_FUSED_QKV=1 for single-stream use.
fused_qkv = os.environ.get("ASIS_FUSED_QKV", "0") not in ("0", "")

# The CNN encoder of the frozen-backbone step runs on a side HIP stream, overlapped with the ViT block loop
# (engines.SegEngine._encoder_on_side_stream).  ASIS_ENC_STREAM=0: everything on the compute stream.
encoder_stream = os.environ.get("ASIS_ENC_STREAM", "1") not in ("0", "")

# The V^T GEMMs of a stacked attention (two batched, ragged launches of 672 tiles each = 1.3 rounds of the 512 tile slots) run
# on a side HIP stream next to the q|k GEMM (8-phase form: 5.19 rounds of 256 tiles, i.e. a last round that leaves 81 % of the
# CUs idle): the three launches fill each other's partial rounds (blocks.Attention.attend_rows).  ASIS_VT_STREAM=0: in order.
vt_stream = os.environ.get("ASIS_VT_STREAM", "1") not in ("0", "")

# Conv weight gradients of the decode heads run on a side HIP stream (parallel.grad_side_stream): nothing on the backward's
# critical path reads them — only the gradient all-reduce and the optimizer, which wait for that stream
# (StageReducer.stage_done / finish) — so they overlap the dgrad GEMMs and the memory-bound BatchNorm / upsample transposes of
# the stages below.  ASIS_WGRAD_STREAM=0: in order on the compute stream.
wgrad_stream = os.environ.get("ASIS_WGRAD_STREAM", "1") not in ("0", "")

# Layers whose forward conv runs on plain 16-bit operands although split_conv is on (comma-separated stage keys: d1..d4 =
# FeatureDecoder stages, stem3 / stem6 / conv2 / conv3 / conv4 = encoder): the lab switch behind DESIGN.md's per-layer table.
unsplit_layers = set(filter(None, os.environ.get("ASIS_UNSPLIT", "").split(",")))

# The two ViT passes of the frozen-backbone step (cls + pos-embed tokens / raw patch tokens: `train.py:287,300-302`) as two
# independent launch streams instead of one row-stacked stream (engines.SegEngine._trunk_dual): every dense GEMM fills the
# chip in a non-integral number of tile rounds (q|k 5.19, proj / fc2 2.59, fc1 10.4 at 12 images), and the CUs that the tail
# round of one stream's kernel leaves idle start the other stream's next kernel: +1.2 % img/s (same-box A/B, three rounds:
# 194.0-194.5 -> 196.5-196.9).  The first step of an engine runs in order (it fills the per-module operand caches on one
# stream).  ASIS_DUAL_STREAM=0: one stacked stream.
dual_stream = os.environ.get("ASIS_DUAL_STREAM", "1") not in ("0", "")

# The attention output o = softmax(QK^T)V can leave the fused attention kernel as hi + lo 16-bit halves and enter the projection
# GEMM as a split A operand (one extra K-long part on the persistent 8-phase GEMM: +8 % block FLOPs, -3.7 % img/s on the
# ViT-L step).  tests/precision_probe.py (ViT-B/14 12 blocks + UNet, stress weights): rounding o to 16 bits puts 9.0e-4 on the
# logits by itself — white noise injected into the residual stream of every block, which the decode heads amplify 8x more than
# the smooth error of rounded weights — against 1.28e-3 for all sites together.
# ASIS_SPLIT_O = 1 always / 0 never / "auto" (default): SegEngine turns it on where the measured head-room needs it — the
# UNet and MLA heads (they amplify a stream error 3.9x / 3.2x, the FeatureDecoder 2x: full-depth stress goldens 1.29e-3 ->
# 9.1e-4 for config 2) and the unfrozen backbone (config 4: 9.9e-4 -> 7.3e-4) — and leaves the frozen ViT-L + FeatureDecoder
# step (stress golden 9.0e-4) on single 16-bit operands.
# Training forward / backward of the attention on row-major operands only (round 5): V straight out of the qkv GEMM through
# transposing LDS reads in the forward (both stacked passes in one launch), asis_attention_bwd_rows in the backward — no
# transpose_tokens pass anywhere.  ASIS_ATTN_ROWS=0 keeps the V^T forward (the backward is the row-major form either way).
attn_rows = os.environ.get("ASIS_ATTN_ROWS", "1") != "0"

_so = os.environ.get("ASIS_SPLIT_O", "auto").lower()
split_attn_out_policy = None if _so == "auto" else (_so not in ("0", ""))   # None = per engine (SegEngine.split_attn_out)
split_attn_out = bool(split_attn_out_policy)     # what the blocks read; SegEngine sets it at the top of every step

# Precision level of the ViT blocks (ASIS_PRECISE_LEVEL, default 0).  2 = EVERY linear layer of a block on split-precision
# operands: LayerNorm / GELU / SwiGLU / attention outputs as hi + lo halves (A_lo) and the weights as hi + lo halves (B_lo) —
# three K parts per GEMM on the persistent 8-phase kernel, ~2.5x the GEMM time; only q, k, v, P inside the fused attention stay
# single 16-bit.  It exists for checkpoints / heads whose conditioning exceeds what single 16-bit operands can hold at 1e-3: the
# full-depth stress golden of config 5 (ViT-g/14, 2 x 40 block evaluations, MLA head amplifying 3.2x) stands at 1.30e-3 on the
# default policy because every remaining error term is an operand of a big GEMM (tests/precision_probe.py: LayerNorm outputs
# 7.5e-4, SwiGLU hidden 5.2e-4, weights 8.8e-4 on the MLA output); tests/test_gpu_fulldepth.py runs that case on level 2 as well.
# ASIS_PRECISE_LEVEL = "auto" (default): per engine — SegEngine runs level 2 where the measured head-room needs it (the MLA head
# on a >= 40-block backbone, i.e. BASELINE config 5: its full-depth stress golden holds 1e-3 only there) and level 0 elsewhere;
# a number forces that level for every engine.  Like split_attn_out, SegEngine sets ``precise_level`` at the top of every step.
_pl = os.environ.get("ASIS_PRECISE_LEVEL", "auto").lower() or "auto"
precise_level_policy = None if _pl == "auto" else int(_pl)
precise_level = int(precise_level_policy or 0)
# Which linear layers of a block run on split operands at precise_level 2 (ASIS_PRECISE_PARTS, comma separated, of qkv, proj,
# fc1 / w12, fc2 / w3; "auto" (default) = per engine).  On the MX form a split layer costs 1.5 passes whether or not the weight's
# residual takes part (the block-scaled pass pairs both planes), so the saving left is to run a whole layer single.  Measured
# on config 5 (ViT-g/14 x 40 + MLA: full-depth stress golden at batch 1 / bench.py --config 5 at batch 12, round 5, one box,
# profiles/r05_c5_parts_ab.txt):   all four 3.1e-4 / 48.2 img/s;  qkv,proj,fc1 6.8e-4 / 54.1;  proj,fc1,fc2 5.7e-4 / 51.9;
# proj,fc1 8.3e-4 / 58.9;  qkv,proj 1.12e-3 (fails) / 62.1;  proj alone (= level 0 + split_attn_out) 1.21e-3 (fails) / 68.8.
# SegEngine's automatic policy for that geometry is therefore {proj, fc1}: the attention output and the LayerNorm output in
# front of the widest GEMM — the two largest terms of tests/precision_probe.py (1.22e-3, 7.5e-4) — and nothing else.
# The adapters' MSDA linear layers on split operands (weights hi + lo, sampled rows hi + lo): what bfloat16 needs on top of
# precise_level 2 to hold 1e-3 on the stress golden (tests/precision_probe.py vit_large 588 bf16 --fdec: of the 1.49e-3 left at
# level 2 the MSDA group carries 1.43e-3 — weights 9.2e-4, sampled rows 6.2e-4).  ASIS_PRECISE_ADAPTERS = 1 / 0 / "auto"
# (default): on for bfloat16 operands at precise_level >= 2, off otherwise (float16 measures 2.9e-4 for the whole group).
_pa = os.environ.get("ASIS_PRECISE_ADAPTERS", "auto").lower() or "auto"


def precise_adapters_on() -> bool:
    if _pa != "auto":
        return _pa not in ("0", "")
    return operand_dtype == torch.bfloat16 and precise_level >= 2


_pp = os.environ.get("ASIS_PRECISE_PARTS", "auto").lower() or "auto"
precise_parts_policy = None if _pp == "auto" else frozenset(filter(None, _pp.replace("w12", "fc1").replace("w3", "fc2").split(",")))
precise_parts = precise_parts_policy if precise_parts_policy is not None else frozenset(("qkv", "proj", "fc1", "fc2"))
trunk_streams = int(os.environ.get("ASIS_TRUNK_STREAMS", "2") or 2)     # 4: each ViT pass as two image groups (lab)

# The LAST adapter stage's CACNN (`train.py:372-386`) writes a pyramid-token tensor that nothing reads: the decoder input takes
# c4 from the ENCODER output (`train.py:395`), so its result cannot reach the logits, the loss or any gradient.  The engine
# skips that one call (identical results, ~0.8 % of the step's FLOPs); ASIS_ELIDE_DEAD_CACNN=0 runs it anyway (bench.py reports
# the setting as config.dead_cacnn_elided; A/B in DESIGN.md §6).
elide_dead_cacnn = os.environ.get("ASIS_ELIDE_DEAD_CACNN", "1") not in ("0", "")
# Two more calls of the reference step whose results nothing reads or that repeat an identical computation:
#   * the encoder's `fc1` 1x1 conv at 147^2 (`backbones/encoders.py:44,55,68` -> c1): `train.py:279` takes c2, c3, c4 only.
#     ASIS_ELIDE_C1=0 runs it (34 GF + 1.06 GB of fp32 writes at 12 images);
#   * the patch embedding: the reference evaluates `model.patch_embed(inp)` once inside `get_intermediate_layers` (pass A,
#     `train.py:287`) and once for pass B (`:300`) on the same input with the same frozen weights; the engine computes it once.
#     ASIS_SHARE_PATCH_EMBED=0 computes it twice.
# bench.py reports both settings (config.c1_elided, config.patch_embed_shared) and times the step with EVERY reference call
# executed as ``secondary.all_reference_calls``.
elide_c1 = os.environ.get("ASIS_ELIDE_C1", "1") not in ("0", "")
share_patch_embed = os.environ.get("ASIS_SHARE_PATCH_EMBED", "1") not in ("0", "")

# LayerNorm folded into the linear layers around it (include/asis_hip.h: asis_gemm_desc.C_lo / rowstats / res16 / ln_mr): inside the
# frozen trunk the residual stream travels between GEMM epilogues as two 16-bit planes (hi + lo = the 4 bytes per element of the
# fp32 tensor it replaces, ~22 significant bits) with per-row (mean, rstd) from the producing epilogue's partial sums; the
# consumer takes the hi plane as its A operand, diag(ln.weight) folded into its weight, and undoes the normalisation in its
# epilogue: rstd * (x W'^T - mean * colsum(W')) + b'.  The standalone LayerNorm launches (one read of the fp32 stream + one
# 16-bit write each, 4.9 % of the headline step) disappear with no extra HBM traffic.  float16 operands, Mlp blocks,
# precise_level 0, shapes on the 8-phase kernels (>= 256 tiles: the headline batch); everything else runs asis_layernorm as
# before.  ASIS_LN_FOLD=0 disables it.
ln_fold = os.environ.get("ASIS_LN_FOLD", "1") not in ("0", "")

# MX correction operands of the split forward convolutions (include/asis_hip.h: asis_gemm_desc.mx_amax_a / mx_amax_b): the
# rounding residuals of activations and weights travel as two fp8 (e4m3) bytes per element with ONE power-of-two scale per
# tensor, and the two correction terms of a split convolution (x_lo w_hi + x_hi w_lo, which need ~3 significant bits) run as
# ONE block-scaled fp8 MFMA pass (v_mfma_scale_f32_16x16x128_f8f6f4) instead of two 16-bit passes: 2/3 of the MFMA time of a
# split convolution.  Applies where the producing kernel can write the MX form (FeatureDecoder stages; Cin % 64 == 0).
# ASIS_MX_CONV=0: the three 16-bit parts as before.
# float16 operands only by default: a bfloat16 residual is 2^-9 of its value and three fp8 mantissa bits leave 2^-13 of it —
# eight times the figure of the float16 case (2^-16); ASIS_MX_CONV=2 forces it for bfloat16 as well.
_mx = os.environ.get("ASIS_MX_CONV", "1")
mx_conv_all = _mx == "2"
mx_conv = _mx not in ("0", "")


def mx_conv_on() -> bool:
    return mx_conv and (mx_conv_all or operand_dtype == torch.float16)


# The same correction pass for the split-precision LINEAR layers of precise_level 2 (every nn.Linear of a ViT block on hi + lo
# operands): hi x hi on the 16-bit MFMA + ONE block-scaled fp8 pass over the MX planes of activations and weights instead of two
# more 16-bit passes (1.5 instead of 3 pass-equivalents).  float16 operands only, like the convolutions.  ASIS_MX_DENSE=0: three
# 16-bit parts.
mx_dense = os.environ.get("ASIS_MX_DENSE", "1") not in ("0", "")


def mx_dense_on() -> bool:
    return mx_dense and operand_dtype == torch.float16
