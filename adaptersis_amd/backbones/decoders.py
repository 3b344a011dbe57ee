"""HIP-backed decode heads — API / ``state_dict`` mirror of `backbones/decoders.py`.

``FeatureDecoder`` (`decoders.py:92-164`, the head `train.py` trains):
    4 x [conv3x3(+bias) -> BatchNorm2d (TRAIN mode) -> ReLU -> bilinear x2 align_corners=True] -> conv3x3

Forward per stage: implicit-GEMM MFMA conv on NHWC 16-bit input, fp32 output with BatchNorm
statistics from the GEMM epilogue, then one fused BN+ReLU+upsample kernel that writes the next
conv's 16-bit operand.  Backward per stage (the only gradients the reference step produces,
SURVEY.md fact 1): upsample^T + ReLU mask + BN partial sums, BN backward apply, conv dgrad
(implicit GEMM with flipped weights) and conv wgrad (transposed-read split-K GEMM).

The same functional core serves two front-ends:
  * ``forward(x)`` — reference-shaped ``nn.Module`` call on an NCHW fp32 tensor, differentiable
    through a ``torch.autograd.Function`` whose backward runs the HIP backward;
  * ``SegEngine`` — the fused training step, which feeds NHWC 16-bit input and writes gradients
    straight into the flat gradient bucket.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import config, ops, parallel
from ..dinov2.layers.blocks import _Packed, _pack
from . import _bn


# ASIS_DGRAD_MX=0: the decoder's input-gradient convolutions on three 16-bit parts instead of 16-bit + one MX pass
_DGRAD_MX = __import__("os").environ.get("ASIS_DGRAD_MX", "1") not in ("0", "")


class _Stage:
    """Saved tensors of one conv -> BN -> ReLU -> upsample stage."""
    __slots__ = ("x16", "raw", "scale", "shift", "mean", "invstd", "count", "factor", "stride", "pad", "pool")


def _conv_weights(owner: _Packed, key: str, conv: nn.Conv2d, split: bool):
    dt = config.operand_dtype
    if split:   # hi + residual in one pass over the fp32 weight
        w_hi, w_lo, _ = _pack(owner._cache, key + ".wpair", conv.weight, lambda p: ops.pack_conv_weight_pair(p.float().contiguous(), 0, dt))
        return w_hi, w_lo
    return _pack(owner._cache, key + ".w", conv.weight, lambda p: ops.pack_conv_weight(p.float().contiguous(), 0, dt)), None


_KSPLIT = os.environ.get("ASIS_CONV_KSPLIT", "1") != "0"
_CONV_8P = os.environ.get("ASIS_CONV_8P", "1") != "0"   # mirrors the dispatcher's switch (the K-part cost model depends on the form)


def _conv_ksplit(P: int, Cout: int, Cin: int, split: bool) -> int:
    """3 = run the three kernel rows of a 3x3 conv as side-by-side K parts (`asis_gemm_desc.ksplit`).  Cost model per CU
    (256 of them, at most two resident 256 x 128 tiles each): a CU that gets c tiles needs about
    1.7 * (c // 2) + (c % 2) single-tile times (two co-resident tiles take ~1.7x one), and the launch lasts as long as its
    busiest CU.  decoder_1 (332 tiles: some CUs hold two) goes from 1.7 to 3.4 / 3 = 1.13 (measured 3.07 -> 2.13 ms),
    decoder_2 (662) from 2.7 to 2.27; a layer with 252 tiles already has one tile on every CU and would only pay for the
    partial maps (measured: slower), one with 120 leaves half the CUs idle and gains.  Taken when the model promises >= 12 %
    after an 8 % allowance for summing the parts and taking the BatchNorm statistics in a separate pass."""
    if not _KSPLIT or Cin % 64 or P < 256 or Cout < 32 or Cout % 4:
        return 1
    cus = 256
    if _CONV_8P and Cout >= 256 and (Cout % 256 == 0 or Cout >= 1024):
        # the 8-phase 256x256 form (csrc/gemm.hip ASIS_CONV_8P): one workgroup per CU, so a launch lasts ceil(tiles / 256)
        # tile times: decoder_1 (83 x 2 = 166 tiles: a third of the CUs idle) and decoder_2 (331: 1.3 rounds) are cut
        tiles = ((P + 255) // 256) * ((Cout + 255) // 256)
        now, cut = -(-tiles // cus), -(-3 * tiles // cus) / 3.0 * 1.08
        return 3 if cut < 0.88 * now else 1
    tiles = ((P + 255) // 256) * ((Cout + 127) // 128 if Cout > 64 else (Cout + 63) // 64)

    def cost(t):
        c = -(-t // cus)
        return 1.7 * (c // 2) + (c % 2)

    now, cut = cost(tiles), cost(3 * tiles) / 3.0 * 1.08
    r = 3 if cut < 0.88 * now else 1
    if r == 3 and os.environ.get("ASIS_CONV_KSPLIT_DEBUG"):
        print(f"[ksplit] P={P} Cout={Cout} Cin={Cin} tiles={tiles} now={now:.2f} cut={cut:.2f}", flush=True)
    return r


def _conv_weights_mx(owner: _Packed, key: str, conv: nn.Conv2d):
    """(16-bit weight, its MX lo operand (weight side), absolute maximum) — config.mx_conv; one absmax pass + one pack pass"""
    dt = config.operand_dtype
    amax = _w_amax(owner, key, conv)
    return _pack(owner._cache, key + ".wpairmx", conv.weight,
                 lambda p: ops.pack_conv_weight_pair(p.float().contiguous(), 0, dt, mx=True, amax=amax))


def _w_amax(owner: _Packed, key: str, conv: nn.Conv2d):
    """the weight's absolute maximum, shared by the forward and the input-gradient MX packs of one step"""
    return _pack(owner._cache, key + ".wamax", conv.weight, lambda p: ops.conv_weight_absmax(p.float().contiguous()))


def conv_bn_relu_up_forward(owner: _Packed, key: str, x16, x_lo, conv: nn.Conv2d, bn: nn.BatchNorm2d, factor: int,
                            sync_bn: bool, save: bool, training: bool = True, stride: int = 1, pad: int = 1,
                            pool: bool = False, mx_out: bool = False, defer_up: bool = False):
    """(x16, x_lo|None) NHWC -> ((up_hi, up_lo|None), saved stage).  training=False: BatchNorm uses its running
    statistics (``seg_decoder.eval()`` in validate_network, train.py:451) and nothing is saved.  ``stride`` / ``pad``:
    the 3x3 conv's geometry (the CNN encoder's stride-2 stages); ``pool``: MaxPool2d(3, 2, 1) after the ReLU (stem).
    ``defer_up``: do not write the BatchNorm + ReLU + upsampled operand pair — the consumer evaluates it on load from the raw map
    (the classifier conv: ops.conv3x3_smallcout_fwd_up); returns ``(("deferred", raw, scale, shift), stage)``."""
    dt = config.operand_dtype
    split_out = x_lo is not None                   # the next stage's operand pair keeps the caller's precision mode
    if key in config.unsplit_layers:
        x_lo = None                                # this layer's conv on plain 16-bit operands (lab switch)
    split = x_lo is not None
    B, H, W, _ = x16.shape
    OH, OW = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    # ``x_lo`` tagged with an absolute maximum: it is in the MX form (two fp8 bytes per element; config.mx_conv) and the
    # weight's lo operand is built likewise — the two correction terms run as one block-scaled fp8 MFMA pass
    mx_in = ops.mx_amax_of(x_lo) if split else None
    if mx_in is not None:
        w_hi, w_lo, w_amax = _conv_weights_mx(owner, key, conv)
    else:
        w_hi, w_lo = _conv_weights(owner, key, conv, split)
    bias = owner._f32(key + ".b", conv.bias)
    ks = _conv_ksplit(B * OH * OW, conv.out_channels, x16.shape[3], split)
    stats = torch.empty((ops.gemm_tiles_m(B * OH * OW), 2, conv.out_channels), device=x16.device,
                        dtype=torch.float32) if (training and ks == 1) else None
    if mx_in is not None and ks == 1 and ops.conv_halo_ok(x16, conv.out_channels, stride, pad):
        # narrow outputs (64 / 128 channels): the halo-tile kernel stages every input element once per plane instead of nine times
        raw = ops.conv3x3_halo_mx(x16, x_lo, w_hi, w_lo, (mx_in, w_amax), bias_n=bias, want_stats=training)
        if training:
            raw, stats = raw
    elif mx_in is not None:
        raw = ops.conv_gemm_split(x16, x_lo, w_hi, w_lo, 3, 3, stride, pad, bias_n=bias, stats=stats, ksplit=ks, mx=(mx_in, w_amax))
    elif split:
        raw = ops.conv_gemm_split(x16, x_lo, w_hi, w_lo, 3, 3, stride, pad, bias_n=bias, stats=stats, ksplit=ks)
    else:
        raw = ops.conv_gemm(x16, w_hi, 3, 3, stride, pad, bias_n=bias, stats=stats, ksplit=ks)
    if training and ks > 1:
        stats = ops.colstats(raw)
    if training:
        scale, shift, mean, invstd, count = _bn.finalize(stats, B * OH * OW, bn, sync_bn)
    else:
        scale, shift = ops.bn_eval_affine(bn)
        mean = invstd = count = None
        save = False
    if defer_up:
        up = ("deferred", raw, scale, shift)
    elif pool:
        up = ops.bn_relu_maxpool(raw, scale, shift, dt, split_out)
    elif factor > 1 and mx_out and split_out:
        # the consumer is a split convolution that takes MX lo operands: the tensor's maximum first (BatchNorm + ReLU of the
        # low-resolution map; the bilinear upsampling is a convex combination), then the fused kernel writes hi + MX
        up = ops.bn_relu_upsample(raw, scale, shift, factor, dt, True, mx_amax=ops.bn_relu_absmax(raw, scale, shift))
    elif factor > 1:
        up = ops.bn_relu_upsample(raw, scale, shift, factor, dt, split_out)
    elif mx_out and split_out:
        # factor 1 (UNet DoubleConv: conv -> BN -> ReLU -> conv): the same for the plain BatchNorm + ReLU
        up = ops.bn_act(raw, scale, shift, True, dt, True, mx_amax=ops.bn_relu_absmax(raw, scale, shift))
    else:
        up = ops.bn_act(raw, scale, shift, True, dt, split_out)
    if not split_out and not defer_up:
        up = (up, None)
    st = None
    if save:
        st = _Stage()
        st.x16, st.raw, st.scale, st.shift, st.mean, st.invstd, st.count, st.factor = \
            x16, raw, scale, shift, mean, invstd, count, factor
        st.stride, st.pad, st.pool = stride, pad, pool
    return up, st


def conv_bn_relu_up_backward(owner: _Packed, key: str, st: _Stage, dU, conv: nn.Conv2d, bn: nn.BatchNorm2d,
                             inv_scale: float, grads: Dict[str, torch.Tensor], prefix: str, need_dx: bool,
                             sync_bn: bool, conv_name: Optional[str] = None, bn_name: Optional[str] = None):
    """dU fp32 [B, fH, fW, C] (scaled by the loss scale) -> grads[...] (unscaled) and dX fp32 or None.
    Parameter names default to ``prefix.0`` (conv) / ``prefix.1`` (BatchNorm)."""
    conv_name = conv_name or prefix + ".0"
    bn_name = bn_name or prefix + ".1"
    import torch.distributed as dist
    dt = config.operand_dtype
    C = conv.out_channels
    stride, pad = getattr(st, "stride", 1) or 1, getattr(st, "pad", 1)
    if getattr(st, "pool", False):
        g, partial = ops.maxpool_bn_relu_bwd(dU, st.raw, st.scale, st.shift, st.mean, st.invstd)
    else:
        g, partial = ops.upsample_bn_relu_bwd(dU, st.raw, st.scale, st.shift, st.mean, st.invstd, st.factor)
    red = ops.reduce_rows(partial.view(partial.shape[0], 2 * C))  # [2C]: sum g | sum g*xhat (scaled)
    local = red
    if sync_bn and parallel.bn_collectives_on():
        # SyncBatchNorm backward: the 2C global sums feed dx only; gamma / beta gradients stay LOCAL sums (the bucket
        # all-reduce averages them over ranks like every other parameter gradient — torch's SyncBatchNorm does the same)
        local = red.clone()
        dist.all_reduce(red)
    dbeta_s, dgamma_s = red[:C], red[C:]
    split = config.split_conv and need_dx
    # the input-gradient convolution of a stride-1 stage takes dx's lo half in the MX form (its two correction terms as one fp8 pass)
    P_ = st.raw.numel() // C
    mx_bwd = bool(split and stride == 1 and _DGRAD_MX and config.mx_conv_on() and ops.mx_conv_ok(P_, C, conv.in_channels))
    r = ops.bn_bwd_apply(g, st.raw, st.mean, st.invstd, owner._f32(key + ".g", bn.weight), dgamma_s, dbeta_s,
                         st.count, dt, split, mx=mx_bwd)
    dx16, dx_lo, bpart = r if split else (r[0], None, r[1])
    ops.reduce_rows(local[:C].view(1, C), inv_scale, grads[bn_name + ".bias"])      # unscale (n = 1 row)
    ops.reduce_rows(local[C:].view(1, C), inv_scale, grads[bn_name + ".weight"])
    if conv.bias is not None:
        ops.reduce_rows(bpart, inv_scale, grads[conv_name + ".bias"])
    wout = grads[conv_name + ".weight"]
    parallel.wgrad_on_side_stream(lambda: ops.wgrad(dx16, st.x16, C, 3, 3, stride, pad, inv_scale, out=wout), dx16, st.x16)
    if not need_dx:
        return None
    if stride == 1:
        return _dgrad(owner, key, conv, dx16, dx_lo)
    # stride 2: conv_transpose2d = stride-1 correlation of the zero-inserted gradient with the mirrored weights, padded by
    # K - 1 - pad; the dilated map is sized so that the result has the input's H x W (output_padding included)
    H, W = st.x16.shape[1:3]
    pp = 2 - pad
    d_hi, d_lo = ops.dilate2(dx16, dx_lo, H + 2 - 2 * pp, W + 2 - 2 * pp)
    return _dgrad(owner, key, conv, d_hi, d_lo, pad=pp)


def _dgrad(owner: _Packed, key: str, conv: nn.Conv2d, d16, d_lo, pad: int = 1):
    """dX = conv_transpose(dY): implicit GEMM on dY with flipped weights; split precision when d_lo is given
    (the next stage's BatchNorm backward subtracts means: 16-bit rounding noise would be amplified)."""
    dt = config.operand_dtype
    if d_lo is None:
        wd = _pack(owner._cache, key + ".wd", conv.weight, lambda p: ops.pack_conv_weight(p.float().contiguous(), 1, dt))
        return ops.conv_gemm(d16, wd, 3, 3, 1, pad)
    mx_in = ops.mx_amax_of(d_lo)
    if mx_in is not None:
        amax = _w_amax(owner, key, conv)
        wd, wd_mx, w_amax = _pack(owner._cache, key + ".wdpairmx", conv.weight,
                                  lambda p: ops.pack_conv_weight_pair(p.float().contiguous(), 1, dt, mx=True, amax=amax))
        return ops.conv_gemm_split(d16, d_lo, wd, wd_mx, 3, 3, 1, pad, mx=(mx_in, w_amax))
    wd, wd_lo, _ = _pack(owner._cache, key + ".wdpair", conv.weight, lambda p: ops.pack_conv_weight_pair(p.float().contiguous(), 1, dt))
    return ops.conv_gemm_split(d16, d_lo, wd, wd_lo, 3, 3, 1, pad)


class _DecoderFn(torch.autograd.Function):
    """autograd bridge: forward/backward are the HIP pipelines of the owning module."""

    @staticmethod
    def forward(ctx, module, x, *params):
        logits, saved = module._forward_core(*module._to_nhwc16(x), save=True)
        ctx.module, ctx.saved = module, saved
        ctx.names = [n for n, _ in module.named_parameters()]
        return logits.permute(0, 3, 1, 2)  # NCHW view of the NHWC buffer

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.module
        dt = config.operand_dtype
        S = config.loss_scale
        B, C, h, w = dlogits.shape
        d = dlogits.permute(0, 2, 3, 1).contiguous().float().view(B * h * w, C)
        CP = (C + 7) // 8 * 8
        d16 = ops.cast_pad(d, CP, dt, scale=S).view(B, h, w, CP)
        d_lo = ops.cast_pad(d, CP, dt, scale=S, part=1).view(B, h, w, CP) if config.split_conv else None
        grads = {n: torch.empty_like(p) for n, p in m.named_parameters()}
        m._backward_core(ctx.saved, d16, None, 1.0 / S, grads, dlogits_f32=d, d_lo=d_lo)
        ctx.saved = None
        if d16.is_cuda:
            parallel.join_grad_streams()     # weight gradients computed on the side stream (config.wgrad_stream)
        return (None, None) + tuple(grads[n] for n in ctx.names)


class FeatureDecoder(_Packed):
    def __init__(self, img_size=588, inplanes=64, embed_dim=1024, num_classes=2, features=[1024, 512, 256, 128, 64]):
        super().__init__()
        self.img_size, self.features, self.inplanes = img_size, list(features), inplanes
        self.embed_dim, self.num_classes = embed_dim, num_classes
        chans = [features[0] * 3, features[1], features[2], features[3], features[4]]
        for c in chans:
            if c % 8:
                raise ValueError("FeatureDecoder channel counts must be multiples of 8")
        for i in range(4):
            setattr(self, f"decoder_{i + 1}", nn.Sequential(
                nn.Conv2d(chans[i], chans[i + 1], 3, padding=1), nn.BatchNorm2d(chans[i + 1]), nn.ReLU(inplace=True),
                nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)))
        self.final_out = nn.Conv2d(chans[4], num_classes, 3, padding=1)
        self.sync_bn = False  # plain nn.BatchNorm2d in the reference decoder (per-GPU statistics)

    GRAD_ORDER = ("final_out", "decoder_4", "decoder_3", "decoder_2", "decoder_1")  # gradient-ready order of the backward

    # ---- functional core ---------------------------------------------------------------------------
    def _to_nhwc16(self, x):
        """NCHW fp32 (reference call convention) -> NHWC 16-bit operand."""
        B, C, H, W = x.shape
        x2 = x.detach().permute(0, 2, 3, 1).contiguous().float().view(B * H * W, C)
        hi = ops.cast_pad(x2, C, config.operand_dtype).view(B, H, W, C)
        lo = ops.cast_pad(x2, C, config.operand_dtype, part=1).view(B, H, W, C) if config.split_conv else None
        return hi, lo

    def _forward_core(self, x16, x_lo, save: bool, training: Optional[bool] = None):
        """(x16, x_lo|None): NHWC 16-bit decoder input (x_lo = rounding residual for split precision)."""
        training = self.training if training is None else training
        saved: List = []
        a = (x16, x_lo)
        for i in range(1, 5):
            seq = getattr(self, f"decoder_{i}")
            # stages 1..3 feed a split 3x3 convolution (MX lo operands where config.mx_conv and the shapes allow); stage 4 feeds
            # the classifier conv, which takes the 16-bit residuals
            nxt = getattr(self, f"decoder_{i + 1}")[0] if i < 4 else None
            Bq, Hq, Wq, _ = a[0].shape
            mx_out = bool(config.mx_conv_on() and a[1] is not None and nxt is not None and f"d{i + 1}" not in config.unsplit_layers and
                          ops.mx_conv_ok(Bq * 4 * Hq * Wq, nxt.in_channels, nxt.out_channels))
            # stage 4 -> classifier: with 64 channels, <= 8 classes and split operands the upsampled pair (4x the bytes of the raw map)
            # is never written: the classifier conv and its weight gradient evaluate BatchNorm + ReLU + upsampling on load
            fo = self.final_out
            defer = bool(i == 4 and a[1] is not None and ops.FUSE_CLS_UP and seq[0].out_channels == 64 and fo.in_channels == 64 and
                         fo.out_channels <= 8 and Hq >= 4 and Wq >= 8)
            a, st = conv_bn_relu_up_forward(self, f"d{i}", a[0], a[1], seq[0], seq[1], 2, self.sync_bn, save, training, mx_out=mx_out,
                                            defer_up=defer)
            saved.append(st)
        if isinstance(a[0], str):                     # ("deferred", raw, scale, shift)
            logits = ops.conv3x3_smallcout_fwd_up(a[1], a[2], a[3], self._f32("final.wf", self.final_out.weight),
                                                  self._f32("final.b", self.final_out.bias), config.operand_dtype)
            saved.append(None)                        # no classifier input tensor: _final_backward recomputes it from saved[3]
            return logits, saved
        logits = self._final_forward(a)
        saved.append(a[0] if save else None)
        return logits, saved

    def _final_forward(self, a):
        """`decoders.py:135` classifier conv on the (hi, lo) NHWC operand pair."""
        bias = self._f32("final.b", self.final_out.bias)
        fo = self.final_out
        if fo.out_channels <= 16 and fo.in_channels in (8, 16, 32, 64):  # few classes: direct fp32 kernel, no MFMA tile waste
            return ops.conv3x3_smallcout_fwd(a[0], a[1], self._f32("final.wf", fo.weight), bias)
        w_hi, w_lo = _conv_weights(self, "final", self.final_out, a[1] is not None)
        if a[1] is not None:
            return ops.conv_gemm_split(a[0], a[1], w_hi, w_lo, 3, 3, 1, 1, bias_n=bias)
        return ops.conv_gemm(a[0], w_hi, 3, 3, 1, 1, bias_n=bias)

    def _final_backward(self, x5, d16, d_lo, bias_partial, inv_scale, grads, dlogits_f32, st4=None):
        """Gradients of the classifier conv; returns loss_scale * dL/d(its input), fp32 NHWC."""
        C = self.num_classes
        if bias_partial is not None:
            ops.reduce_rows(bias_partial, inv_scale, grads["final_out.bias"])
        else:  # compatibility path: column sums of the fp32 dlogits [P, C]
            ops.reduce_rows(dlogits_f32, 1.0, grads["final_out.bias"])
        wout = grads["final_out.weight"]
        if x5 is None:      # the classifier's input was never written (upsample on load): st4 = the stage that would have produced it
            parallel.wgrad_on_side_stream(lambda: ops.conv3x3_smallcout_wgrad_up(d16, st4.raw, st4.scale, st4.shift, C, inv_scale, out=wout),
                                          d16, st4.raw)
        else:
            parallel.wgrad_on_side_stream(lambda: ops.wgrad(d16, x5, C, 3, 3, 1, 1, inv_scale, out=wout), d16, x5)
        fo = self.final_out
        if fo.out_channels <= 8 and fo.in_channels <= 224:
            return ops.conv3x3_smallcout_dgrad(d16, d_lo, self._f32("final.wf", fo.weight))
        return _dgrad(self, "final", fo, d16, d_lo)

    def _backward_core(self, saved, d16, bias_partial, inv_scale, grads, dlogits_f32=None, stage_done=None, d_lo=None,
                       need_input_grad: bool = False):
        """d16: 16-bit [B,h,w,CP] = loss_scale * dL/dlogits (pad channels zero).  ``stage_done()`` is
        called after the final conv and after each decoder stage (4,3,2,1) once its gradients are
        enqueued — the engine launches that stage's gradient all-reduce from it.  ``need_input_grad``: also return
        loss_scale * dL/d(input) as fp32 NHWC (end-to-end training of the backbone)."""
        dU = self._final_backward(saved[4], d16, d_lo, bias_partial, inv_scale, grads, dlogits_f32, st4=saved[3])
        if stage_done is not None:
            stage_done()
        for i in range(4, 0, -1):
            seq = getattr(self, f"decoder_{i}")
            dU = conv_bn_relu_up_backward(self, f"d{i}", saved[i - 1], dU, seq[0], seq[1], inv_scale, grads,
                                          f"decoder_{i}", need_dx=(i > 1 or need_input_grad), sync_bn=self.sync_bn)
            if stage_done is not None:
                stage_done()
        return dU

    # ---- reference-shaped entry point ----------------------------------------------------------------
    def forward(self, x):
        """`decoders.py:137-164`: (B, 3*embed, h, w) fp32 -> logits (B, classes, 16h, 16w) fp32."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _DecoderFn.apply(self, x, *list(self.parameters()))
        logits, _ = self._forward_core(*self._to_nhwc16(x), save=False)
        return logits.permute(0, 3, 1, 2)


class DecoderSETR(FeatureDecoder):
    """`backbones/decoders.py:167-203`: the SETR-style progressive-upsampling head — the same four
    conv3x3+BN+ReLU+bilinear x2 stages and final conv3x3 as ``FeatureDecoder`` (identical ``state_dict`` keys), fed by
    ``in_channels`` instead of the 3*embed concat."""

    def __init__(self, in_channels, out_channels, features=[512, 256, 128, 64]):
        _Packed.__init__(self)
        chans = [in_channels] + list(features)
        for c in chans:
            if c % 8:
                raise ValueError("DecoderSETR channel counts must be multiples of 8")
        self.in_channels, self.out_channels, self.features = in_channels, out_channels, list(features)
        self.num_classes = out_channels
        for i in range(4):
            setattr(self, f"decoder_{i + 1}", nn.Sequential(
                nn.Conv2d(chans[i], chans[i + 1], 3, padding=1), nn.BatchNorm2d(chans[i + 1]), nn.ReLU(inplace=True),
                nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)))
        self.final_out = nn.Conv2d(chans[4], out_channels, 3, padding=1)
        self.sync_bn = False


class _SETRFFn(torch.autograd.Function):
    """autograd bridge of DecoderSETRF: x and the three skips in, logits out; gradients for the parameters, the skips and x."""

    @staticmethod
    def forward(ctx, module, x, c1, c2, c3, *params):
        logits, saved = module._forward_core(module._to_nhwc16(x), [module._to_nhwc16(c) for c in (c1, c2, c3)], save=True)
        ctx.module, ctx.saved = module, saved
        ctx.names = [n for n, _ in module.named_parameters()]
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.module
        dt = config.operand_dtype
        S = config.loss_scale
        B, C, h, w = dlogits.shape
        d = dlogits.permute(0, 2, 3, 1).contiguous().float().view(B * h * w, C)
        CP = (C + 7) // 8 * 8
        d16 = ops.cast_pad(d, CP, dt, scale=S).view(B, h, w, CP)
        d_lo = ops.cast_pad(d, CP, dt, scale=S, part=1).view(B, h, w, CP) if config.split_conv else None
        grads = {n: torch.empty_like(p) for n, p in m.named_parameters()}
        dx, dcs = m._backward_core(ctx.saved, d16, 1.0 / S, grads, d, d_lo, need_x=ctx.needs_input_grad[1])
        ctx.saved = None
        nchw = lambda t: None if t is None else (t * (1.0 / S)).permute(0, 3, 1, 2)
        dc = [nchw(t) if need else None for t, need in zip(dcs, ctx.needs_input_grad[2:5])]
        if d16.is_cuda:
            parallel.join_grad_streams()
        return (None, nchw(dx), dc[0], dc[1], dc[2]) + tuple(grads[n] for n in ctx.names)


class DecoderSETRF(FeatureDecoder):
    """`backbones/decoders.py:205-257`: DecoderSETR with the CNN pyramid fused in.  After stage 2 the map is
    zero-padded (centred, `:240-243`) to c3's size and concatenated with it along the channels; c2 joins after stage 3
    and c1 after stage 4, so decoder_3 / decoder_4 / final_out see twice the channels.  Same ``state_dict`` keys as the
    reference; the convs, BatchNorm(train), ReLU, bilinear x2 and all their gradients are the FeatureDecoder kernels,
    the pad + concat between them is plain tensor plumbing on the 16-bit NHWC operands."""

    def __init__(self, in_channels, out_channels, features=[512, 256, 128, 64]):
        _Packed.__init__(self)
        f = list(features)
        cin = [in_channels, f[0], 2 * f[1], 2 * f[2]]
        for c in cin + f + [2 * f[3]]:
            if c % 8:
                raise ValueError("DecoderSETRF channel counts must be multiples of 8")
        self.in_channels, self.out_channels, self.features = in_channels, out_channels, f
        self.num_classes = out_channels
        for i in range(4):
            setattr(self, f"decoder_{i + 1}", nn.Sequential(
                nn.Conv2d(cin[i], f[i], 3, padding=1), nn.BatchNorm2d(f[i]), nn.ReLU(inplace=True),
                nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)))
        self.final_out = nn.Conv2d(2 * f[3], out_channels, 3, padding=1)
        self.sync_bn = False

    @staticmethod
    def _fuse(a, c):
        """(hi, lo) NHWC stage output + (hi, lo) NHWC skip -> zero-padded concat and the crop window of the first part."""
        B, h, w, Cx = a[0].shape
        H, Wd = c[0].shape[1:3]
        dy, dx = H - h, Wd - w
        if dy < 0 or dx < 0:
            raise ValueError(f"DecoderSETRF: the skip ({H}x{Wd}) must not be smaller than the decoder map ({h}x{w})")
        top, left = dy // 2, dx // 2
        pad = (0, 0, left, dx - left, top, dy - top)
        def cat(u, v):
            if dy == 0 and dx == 0:     # equal maps (every stage at 588 and the other 16-divisible sizes): two strided row copies
                out = torch.empty((B, H, Wd, Cx + v.shape[3]), device=u.device, dtype=u.dtype)
                o2 = out.view(-1, out.shape[3])
                ops.copy_channels(u.reshape(-1, Cx), o2[:, :Cx])
                ops.copy_channels(v.reshape(-1, v.shape[3]), o2[:, Cx:])
                return out
            return torch.cat([F.pad(u, pad), v], dim=3).contiguous()   # odd sizes: zero border through ATen (copies only)
        hi = cat(a[0], c[0])
        lo = cat(a[1], c[1]) if a[1] is not None and c[1] is not None else None
        return (hi, lo), (top, left, h, w, Cx)

    @staticmethod
    def _unfuse(d, win):
        top, left, h, w, Cx = win
        return d[:, top:top + h, left:left + w, :Cx].contiguous(), d[..., Cx:]

    def _forward_core(self, x, skips, save: bool, training: Optional[bool] = None):
        training = self.training if training is None else training
        c1, c2, c3 = skips
        saved, wins = [], {}
        a = x
        for i in range(1, 5):
            if i == 3:
                a, wins[3] = self._fuse(a, c3)
            elif i == 4:
                a, wins[4] = self._fuse(a, c2)
            seq = getattr(self, f"decoder_{i}")
            a, st = conv_bn_relu_up_forward(self, f"d{i}", a[0], a[1], seq[0], seq[1], 2, self.sync_bn, save, training)
            saved.append(st)
        a, wins[5] = self._fuse(a, c1)
        logits = self._final_forward(a)
        saved.append(a[0] if save else None)
        saved.append(wins)
        return logits, saved

    def _backward_core(self, saved, d16, inv_scale, grads, dlogits_f32, d_lo, need_x: bool = False):
        """-> (loss_scale * dL/dx or None, [loss_scale * dL/dc1, dc2, dc3]) as fp32 NHWC."""
        wins = saved[5]
        dU = self._final_backward(saved[4], d16, d_lo, None, inv_scale, grads, dlogits_f32)
        dU, dc1 = self._unfuse(dU, wins[5])
        dcs = {1: dc1}
        for i in range(4, 0, -1):
            seq = getattr(self, f"decoder_{i}")
            dU = conv_bn_relu_up_backward(self, f"d{i}", saved[i - 1], dU, seq[0], seq[1], inv_scale, grads,
                                          f"decoder_{i}", need_dx=(i > 1 or need_x), sync_bn=self.sync_bn)
            if i == 4:
                dU, dcs[2] = self._unfuse(dU, wins[4])
            elif i == 3:
                dU, dcs[3] = self._unfuse(dU, wins[3])
        return dU, [dcs[1], dcs[2], dcs[3]]

    def forward(self, x, c1, c2, c3):
        """`decoders.py:238-257`: x (B, in_channels, h, w) and the pyramid c1 (finest) .. c3 -> logits, fp32 NCHW."""
        if self.training and torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                          or any(t.requires_grad for t in (x, c1, c2, c3))):
            return _SETRFFn.apply(self, x, c1, c2, c3, *list(self.parameters()))
        logits, _ = self._forward_core(self._to_nhwc16(x), [self._to_nhwc16(c) for c in (c1, c2, c3)], save=False)
        return logits.permute(0, 3, 1, 2)


class _MLAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, i0, i1, i2, i3, *params):
        ins = [module._to_nhwc16(t) for t in (i0, i1, i2, i3)]
        logits, saved = module._forward_core(ins, save=True)
        ctx.module, ctx.saved = module, saved
        ctx.names = [n for n, _ in module.named_parameters()]
        B, h, w, C = logits.shape
        out = ops.resize_bilinear_fwd(logits, module.img_size, module.img_size)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        m = ctx.module
        dt = config.operand_dtype
        S = config.loss_scale
        B, C, H, W = dout.shape
        dz = dout.permute(0, 2, 3, 1).contiguous().float()
        h = m._last_hw
        d32, bpart = ops.resize_bilinear_bwd(dz, h, h, torch.float32)
        d2 = d32.view(B * h * h, C)
        CP = (C + 7) // 8 * 8
        d16 = ops.cast_pad(d2, CP, dt, scale=S).view(B, h, h, CP)
        d_lo = ops.cast_pad(d2, CP, dt, scale=S, part=1).view(B, h, h, CP) if config.split_conv else None
        grads = {n: torch.empty_like(p) for n, p in m.named_parameters()}
        m._backward_core(ctx.saved, d16, None, 1.0 / S, grads, dlogits_f32=d2, d_lo=d_lo)
        ctx.saved = None
        if d16.is_cuda:
            parallel.join_grad_streams()
        return (None, None, None, None, None) + tuple(grads[n] for n in ctx.names)


class MLAHead(nn.Module):
    """`backbones/decoders.py:7-46` parameter container (4 x [conv-BN-ReLU, conv-BN-ReLU], no conv bias)."""

    def __init__(self, mla_channels=1024, mlahead_channels=128, norm_cfg=None):
        super().__init__()
        for h in ("head2", "head3", "head4", "head5"):
            setattr(self, h, nn.Sequential(
                nn.Conv2d(mla_channels, mlahead_channels, 3, padding=1, bias=False), nn.BatchNorm2d(mlahead_channels), nn.ReLU(),
                nn.Conv2d(mlahead_channels, mlahead_channels, 3, padding=1, bias=False), nn.BatchNorm2d(mlahead_channels),
                nn.ReLU()))


class DecoderMLA(_Packed):
    """`backbones/decoders.py:48-89` (the `train_mla.py` head).  Unlike the reference, ``num_classes`` is honoured
    (the reference hard-codes 2, `decoders.py:59`; BASELINE config 5 needs 11)."""

    def __init__(self, img_size=588, mla_channels=1024, mlahead_channels=128, norm_layer=nn.BatchNorm2d, num_classes=2,
                 norm_cfg=None):
        super().__init__()
        if mla_channels % 8 or mlahead_channels % 8:
            raise ValueError("DecoderMLA channel counts must be multiples of 8")
        self.img_size, self.norm_cfg, self.mla_channels = img_size, norm_cfg, mla_channels
        self.BatchNorm, self.mlahead_channels, self.num_classes = norm_layer, mlahead_channels, num_classes
        self.mlahead = MLAHead(mla_channels, mlahead_channels, norm_cfg)
        self.cls = nn.Sequential(nn.Conv2d(4 * mlahead_channels, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(inplace=True))
        self.cls_1 = nn.Sequential(nn.Conv2d(256, 128, 3, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True))
        self.cls_2 = nn.Sequential(nn.Conv2d(128, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.cls_3 = nn.Conv2d(64, num_classes, 3, padding=1)
        self.sync_bn = False
        self._last_hw = None

    _HEADS = ("head2", "head3", "head4", "head5")
    GRAD_ORDER = ("cls_3", "cls_2", "cls_1", "cls", "mlahead")

    def _to_nhwc16(self, x):
        B, C, H, W = x.shape
        x2 = x.detach().permute(0, 2, 3, 1).contiguous().float().view(B * H * W, C)
        hi = ops.cast_pad(x2, C, config.operand_dtype).view(B, H, W, C)
        lo = ops.cast_pad(x2, C, config.operand_dtype, part=1).view(B, H, W, C) if config.split_conv else None
        return hi, lo

    def _forward_core(self, ins, save: bool, training: Optional[bool] = None):
        """ins: 4 x (hi, lo|None) NHWC 16-bit maps [B,h,w,mla_channels] -> logits fp32 NHWC at 4h x 4w."""
        training = self.training if training is None else training
        dt = config.operand_dtype
        split = ins[0][1] is not None
        Cm = self.mlahead_channels
        B, h, w, _ = ins[0][0].shape
        cat_hi = torch.empty((B, 4 * h, 4 * w, 4 * Cm), device=ins[0][0].device, dtype=dt)
        cat_lo = torch.empty_like(cat_hi) if split else None
        saved = {"heads": []}
        for k, (hn, (xh, xl)) in enumerate(zip(self._HEADS, ins)):
            seq = getattr(self.mlahead, hn)
            a, s1 = conv_bn_relu_up_forward(self, hn + "a", xh, xl, seq[0], seq[1], 1, self.sync_bn, save, training)
            u, s2 = conv_bn_relu_up_forward(self, hn + "b", a[0], a[1], seq[3], seq[4], 4, self.sync_bn, save, training)
            ops.copy_channels(u[0].view(-1, Cm), cat_hi.view(-1, 4 * Cm)[:, k * Cm:(k + 1) * Cm])
            if split:
                ops.copy_channels(u[1].view(-1, Cm), cat_lo.view(-1, 4 * Cm)[:, k * Cm:(k + 1) * Cm])
            saved["heads"].append((s1, s2))
        a = (cat_hi, cat_lo)
        saved["cls"] = []
        for name in ("cls", "cls_1", "cls_2"):
            seq = getattr(self, name)
            a, st = conv_bn_relu_up_forward(self, name, a[0], a[1], seq[0], seq[1], 1, self.sync_bn, save, training)
            saved["cls"].append(st)
        fo = self.cls_3
        bias = self._f32("cls3.b", fo.bias)
        if fo.out_channels <= 16:
            logits = ops.conv3x3_smallcout_fwd(a[0], a[1], self._f32("cls3.wf", fo.weight), bias)
        else:
            w_hi, w_lo = _conv_weights(self, "cls3", fo, split)
            logits = ops.conv_gemm_split(a[0], a[1], w_hi, w_lo, 3, 3, 1, 1, bias_n=bias) if split else \
                ops.conv_gemm(a[0], w_hi, 3, 3, 1, 1, bias_n=bias)
        saved["x_last"] = a[0] if save else None
        self._last_hw = 4 * h
        return logits, saved

    def _backward_core(self, saved, d16, bias_partial, inv_scale, grads, dlogits_f32=None, stage_done=None, d_lo=None,
                       need_input_grad: bool = False):
        """``need_input_grad`` (train_adapters on the `train_mla.py` flow): also return the gradients of the four input
        maps, fp32 NHWC [B, h, w, mla_channels] each (still multiplied by the loss scale), in argument order."""
        fo = self.cls_3
        C = self.num_classes
        Cm = self.mlahead_channels
        dins = []
        if bias_partial is not None:
            ops.reduce_rows(bias_partial, inv_scale, grads["cls_3.bias"])
        else:
            ops.reduce_rows(dlogits_f32, 1.0, grads["cls_3.bias"])
        ops.wgrad(d16, saved["x_last"], C, 3, 3, 1, 1, inv_scale, out=grads["cls_3.weight"])
        if fo.out_channels <= 8:
            dU = ops.conv3x3_smallcout_dgrad(d16, d_lo, self._f32("cls3.wf", fo.weight))
        else:
            dU = _dgrad(self, "cls3", fo, d16, d_lo)
        if stage_done is not None:
            stage_done()
        for name, st in zip(("cls_2", "cls_1", "cls"), reversed(saved["cls"])):
            seq = getattr(self, name)
            dU = conv_bn_relu_up_backward(self, name, st, dU, seq[0], seq[1], inv_scale, grads, name, True, self.sync_bn)
            if stage_done is not None:
                stage_done()
        B, H, W, _ = dU.shape
        for k, hn in enumerate(self._HEADS):
            seq = getattr(self.mlahead, hn)
            s1, s2 = saved["heads"][k]
            dh = torch.empty((B, H, W, Cm), device=dU.device, dtype=torch.float32)
            ops.copy_channels(dU.view(-1, 4 * Cm)[:, k * Cm:(k + 1) * Cm], dh.view(-1, Cm))
            p = f"mlahead.{hn}"
            dx = conv_bn_relu_up_backward(self, hn + "b", s2, dh, seq[3], seq[4], inv_scale, grads, p, True, self.sync_bn,
                                          conv_name=p + ".3", bn_name=p + ".4")
            dins.append(conv_bn_relu_up_backward(self, hn + "a", s1, dx, seq[0], seq[1], inv_scale, grads, p, need_input_grad,
                                                 self.sync_bn, conv_name=p + ".0", bn_name=p + ".1"))
        if stage_done is not None:
            stage_done()
        return dins if need_input_grad else None

    def forward(self, input, input1, input2, input3):
        """`decoders.py:82-89`: four (B, C, h, w) maps -> logits resized to (B, classes, img_size, img_size)."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _MLAFn.apply(self, input, input1, input2, input3, *list(self.parameters()))
        logits, _ = self._forward_core([self._to_nhwc16(t) for t in (input, input1, input2, input3)], save=False)
        return ops.resize_bilinear_fwd(logits, self.img_size, self.img_size).permute(0, 3, 1, 2)
