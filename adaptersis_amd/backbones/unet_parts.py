"""HIP-backed UNet decode head — API / ``state_dict`` mirror of `backbones/unet_parts.py` (BASELINE config 2 head,
`eval/eval_dinov2_unet.py:152`).

    x3 = input (B, C, h, w)
    x4 = Down(C, 2C)(x3)           MaxPool2d(2) -> DoubleConv                       unet_parts.py:26-38
    x5 = Down(2C, 4C)(x4)
    y  = Up(4C, 2C)(x5, x4)        ConvTranspose2d(4C, 2C, 2, 2) -> pad -> cat([x4, up]) -> DoubleConv   :41-64
    y  = Up(2C, C)(y, x3)
    y  = Up_wc(C, C/2)(y)          ConvTranspose2d(C, C, 2, 2) -> DoubleConv        :66-92
    y  = Up_wc(C/2, C/4)(y)
    logits = OutConv(C/4, classes)(y)    1x1 conv                                    :95-101

The reference hard-wires C = 384 (ViT-S) and ignores ``n_channels``; here the widths follow ``n_channels`` (384 gives
the reference's exact module; BASELINE config 2 needs 768 for ViT-B).

DoubleConv = 2 x [implicit-GEMM conv3x3 -> BatchNorm (train-mode statistics from the GEMM epilogue) -> ReLU], the same
fused stage the other heads use (`decoders.conv_bn_relu_up_forward`, factor 1); ConvTranspose2d(k=2, s=2) is ONE MFMA
GEMM over N = 4*Cout columns plus a pixel-shuffle kernel that writes the 16-bit operand straight into the concat
buffer (`csrc/unet.hip`); MaxPool2d keeps an argmax byte per element for the backward.  Every parameter of the head
trains; the backward produces all of them (conv wgrad / dgrad, BatchNorm, ConvTranspose2d as dgrad GEMM + 1x1 wgrad).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .. import config, ops
from ..dinov2.layers.blocks import _Packed, _pack
from .decoders import _DecoderFn, conv_bn_relu_up_backward, conv_bn_relu_up_forward


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2 — parameter container (`unet_parts.py:6-23`)."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))


class Down(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))


class Up(nn.Module):
    def __init__(self, in_channels, out_channels, bilinear=False):
        super().__init__()
        if bilinear:
            raise NotImplementedError("UNet(bilinear=True) is never used by the reference scripts")
        self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels)


class Up_wc(nn.Module):
    def __init__(self, in_channels, out_channels, bilinear=False):
        super().__init__()
        if bilinear:
            raise NotImplementedError("UNet(bilinear=True) is never used by the reference scripts")
        self.up = nn.ConvTranspose2d(in_channels, in_channels, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels)


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)


class _Up:
    """Saved tensors of one ConvTranspose2d (+ concat) step."""
    __slots__ = ("x16", "B", "H", "W", "Cout", "coff", "padT", "padL")


class UNet(_Packed):
    def __init__(self, n_channels, n_classes, bilinear=False):
        super().__init__()
        if bilinear:
            raise NotImplementedError("UNet(bilinear=True) is never used by the reference scripts")
        C = int(n_channels)
        if C % 32:
            raise ValueError("UNet: n_channels must be a multiple of 32 (C/4 feeds 16-byte vector kernels)")
        self.n_channels, self.n_classes, self.bilinear = C, n_classes, bilinear
        self.num_classes = n_classes
        self.down3 = Down(C, 2 * C)
        self.down4 = Down(2 * C, 4 * C)
        self.up1 = Up(4 * C, 2 * C)
        self.up2 = Up(2 * C, C)
        self.up3 = Up_wc(C, C // 2)
        self.up4 = Up_wc(C // 2, C // 4)
        self.outc = OutConv(C // 4, n_classes)
        self.sync_bn = False

    GRAD_ORDER = ("outc", "up4", "up3", "up2", "up1", "down4", "down3")  # gradient-ready order of the backward

    # ---- building blocks ---------------------------------------------------------------------------------------
    def _dconv_fwd(self, key: str, dc: DoubleConv, a, save: bool, training: bool):
        seq = dc.double_conv
        # the first conv's BatchNorm + ReLU output feeds the second 3x3 conv only: its lo half in the MX form where that conv can take it
        Bq, Hq, Wq, _ = a[0].shape
        mx_mid = bool(config.mx_conv_on() and a[1] is not None and ops.mx_conv_ok(Bq * Hq * Wq, seq[3].in_channels, seq[3].out_channels))
        a, s1 = conv_bn_relu_up_forward(self, key + "a", a[0], a[1], seq[0], seq[1], 1, self.sync_bn, save, training, mx_out=mx_mid)
        a, s2 = conv_bn_relu_up_forward(self, key + "b", a[0], a[1], seq[3], seq[4], 1, self.sync_bn, save, training)
        return a, (s1, s2)

    def _dconv_bwd(self, key: str, dc: DoubleConv, st, dU, inv_scale, grads, prefix: str, need_dx: bool):
        seq = dc.double_conv
        p = prefix + ".double_conv"
        dU = conv_bn_relu_up_backward(self, key + "b", st[1], dU, seq[3], seq[4], inv_scale, grads, p, True, self.sync_bn,
                                      conv_name=p + ".3", bn_name=p + ".4")
        return conv_bn_relu_up_backward(self, key + "a", st[0], dU, seq[0], seq[1], inv_scale, grads, p, need_dx,
                                        self.sync_bn, conv_name=p + ".0", bn_name=p + ".1")

    def _convt_weights(self, key: str, up: nn.ConvTranspose2d, split: bool):
        dt = config.operand_dtype
        cin = up.in_channels

        def fwd(part):  # B operand of the forward GEMM: rows n = co*4 + di*2 + dj, K = Cin
            return lambda p: ops.cast_pad(p.float().reshape(cin, -1).t().contiguous(), dtype=dt, part=part)

        def bwd(part):  # B operand of the dgrad GEMM: rows ci, K = 4*Cout
            return lambda p: ops.cast_pad(p.float().reshape(cin, -1).contiguous(), dtype=dt, part=part)

        c = self._cache
        return (_pack(c, key + ".wf", up.weight, fwd(0)), _pack(c, key + ".wflo", up.weight, fwd(1)) if split else None,
                _pack(c, key + ".wb", up.weight, bwd(0)), _pack(c, key + ".wblo", up.weight, bwd(1)) if split else None)

    def _up_fwd(self, key: str, up: nn.ConvTranspose2d, x, skip, save: bool):
        """x (hi, lo) [B,H,W,Cin] -> concat buffer (hi, lo) [B,H2,W2,Cskip+Cout] with the skip in the leading channels
        (`unet_parts.py:54-63`) or, without a skip (Up_wc), the plain upsampled map."""
        dt = config.operand_dtype
        xh, xl = x
        split = xl is not None
        B, H, W, Cin = xh.shape
        Cout = up.out_channels
        wf, wflo, _, _ = self._convt_weights(key, up, split)
        bias4 = _pack(self._cache, key + ".b4", up.bias, lambda p: p.float().repeat_interleave(4).contiguous())
        G = torch.empty((B * H * W, 4 * Cout), device=xh.device, dtype=torch.float32)
        a2 = xh.view(B * H * W, Cin)
        if split:
            ops.gemm_split(a2, xl.view(B * H * W, Cin), wf, wflo, out=G, bias_n=bias4)
        else:
            ops.gemm(a2, wf, out=G, bias_n=bias4)
        if skip is None:
            H2, W2, coff, padT, padL = 2 * H, 2 * W, 0, 0, 0
            cat_hi = torch.empty((B, H2, W2, Cout), device=xh.device, dtype=dt)
        else:
            _, H2, W2, Cs = skip[0].shape
            dY, dX = H2 - 2 * H, W2 - 2 * W
            if dY < 0 or dX < 0:
                raise ValueError("UNet: the skip map is smaller than the upsampled map")
            coff, padT, padL = Cs, dY // 2, dX // 2
            alloc = torch.zeros if (dY or dX) else torch.empty
            cat_hi = alloc((B, H2, W2, Cs + Cout), device=xh.device, dtype=dt)
        cat_lo = (torch.zeros_like(cat_hi) if (skip is not None and (dY or dX)) else torch.empty_like(cat_hi)) if split else None
        if skip is not None:
            Ct = cat_hi.shape[-1]
            ops.copy_channels(skip[0].view(-1, coff), cat_hi.view(-1, Ct)[:, :coff])
            if split:
                ops.copy_channels(skip[1].view(-1, coff), cat_lo.view(-1, Ct)[:, :coff])
        ops.convt2x2_scatter(G, cat_hi, cat_lo, B, H, W, coff, padT, padL)
        st = None
        if save:
            st = _Up()
            st.x16, st.B, st.H, st.W, st.Cout, st.coff, st.padT, st.padL = xh, B, H, W, Cout, coff, padT, padL
        return (cat_hi, cat_lo), st

    def _up_bwd(self, key: str, up: nn.ConvTranspose2d, st: _Up, dcat, inv_scale, grads, prefix: str):
        """dcat fp32 [B,H2,W2,Ctot] (scaled) -> grads of the ConvTranspose2d and dX fp32 [B,H,W,Cin]."""
        dt = config.operand_dtype
        split = config.split_conv
        Cin = up.in_channels
        dG, dG_lo, bpart = ops.convt2x2_gather(dcat, st.B, st.H, st.W, st.Cout, st.coff, st.padT, st.padL, dt, split)
        ops.reduce_rows(bpart, inv_scale, grads[prefix + ".bias"])
        # dW[ci, (co, di, dj)] = sum_p x[p, ci] dG[p, n]: the 1x1 "weight gradient" with x in the dy role
        ops.wgrad(st.x16, dG.view(st.B, st.H, st.W, 4 * st.Cout), Cin, 1, 1, 1, 0, inv_scale,
                  out=grads[prefix + ".weight"].view(Cin, 4 * st.Cout, 1, 1))
        _, _, wb, wblo = self._convt_weights(key, up, split)
        dX = torch.empty((st.B * st.H * st.W, Cin), device=dcat.device, dtype=torch.float32)
        if split:
            ops.gemm_split(dG, dG_lo, wb, wblo, out=dX)
        else:
            ops.gemm(dG, wb, out=dX)
        return dX.view(st.B, st.H, st.W, Cin)

    # ---- OutConv (1x1) --------------------------------------------------------------------------------------------------
    def _outc_fwd(self, y):
        """y (hi, lo|None) [B,H,W,Cq] -> logits fp32 NHWC [B,H,W,classes]"""
        oc = self.outc.conv
        B, H, W, Cq = y[0].shape
        logits = torch.empty((B * H * W, oc.out_channels), device=y[0].device, dtype=torch.float32)
        w_hi = self._w16("outc.w", oc.weight)
        bias = self._f32("outc.b", oc.bias)
        if y[1] is not None:
            w_lo = _pack(self._cache, "outc.wlo", oc.weight,
                         lambda p: ops.cast_pad(p.reshape(p.shape[0], -1).contiguous().float(), dtype=config.operand_dtype, part=1))
            ops.gemm_split(y[0].view(-1, Cq), y[1].view(-1, Cq), w_hi, w_lo, out=logits, bias_n=bias)
        else:
            ops.gemm(y[0].view(-1, Cq), w_hi, out=logits, bias_n=bias)
        return logits.view(B, H, W, oc.out_channels)

    def _outc_bwd(self, xl, d16, bias_partial, inv_scale, grads, dlogits_f32=None, d_lo=None):
        """parameter gradients of the 1x1 classifier and dU fp32 [B,H,W,Cq] (scaled) for the stage below"""
        dt = config.operand_dtype
        oc = self.outc.conv
        C = oc.out_channels
        B, H, W, Cq = xl.shape
        CP = d16.shape[-1]
        if bias_partial is not None:
            ops.reduce_rows(bias_partial, inv_scale, grads["outc.conv.bias"])
        else:
            ops.reduce_rows(dlogits_f32, 1.0, grads["outc.conv.bias"])
        ops.wgrad(d16, xl, C, 1, 1, 1, 0, inv_scale, out=grads["outc.conv.weight"])
        if C <= 8 and Cq % 4 == 0 and Cq <= 1024:
            # a handful of classes: one fp32 pass over dU instead of a K = 8 GEMM (three of them in split precision)
            w32 = self._f32("outc.w32", oc.weight).view(C, Cq)
            return ops.conv1x1_dgrad_small(d16.view(-1, CP), None if d_lo is None else d_lo.view(-1, CP), w32, C).view(B, H, W, Cq)
        # dgrad of the 1x1 conv: dY [P, CP] x W^T; B operand [Cq, CP] = weight^T zero-padded to CP columns
        def wt(part):
            return lambda p: ops.cast_pad(p.float().reshape(C, Cq).t().contiguous(), CP, dt, part=part)
        wd = _pack(self._cache, f"outc.wd{CP}", oc.weight, wt(0))
        dU = torch.empty((B * H * W, Cq), device=d16.device, dtype=torch.float32)
        if d_lo is not None:
            wdlo = _pack(self._cache, f"outc.wdlo{CP}", oc.weight, wt(1))
            ops.gemm_split(d16.view(-1, CP), d_lo.view(-1, CP), wd, wdlo, out=dU)
        else:
            ops.gemm(d16.view(-1, CP), wd, out=dU)
        return dU.view(B, H, W, Cq)

    # ---- functional core -----------------------------------------------------------------------------------------
    def _to_nhwc16(self, x):
        B, C, H, W = x.shape
        x2 = x.detach().permute(0, 2, 3, 1).contiguous().float().view(B * H * W, C)
        hi = ops.cast_pad(x2, C, config.operand_dtype).view(B, H, W, C)
        lo = ops.cast_pad(x2, C, config.operand_dtype, part=1).view(B, H, W, C) if config.split_conv else None
        return hi, lo

    def _forward_core(self, x16, x_lo, save: bool, training: Optional[bool] = None, need_input_grad: bool = False):
        """(hi, lo|None) NHWC 16-bit [B,h,w,C] -> logits fp32 NHWC [B,4h,4w,classes] and the saved state.
        ``need_input_grad``: also keep the first MaxPool's arg-max (the input is trainable upstream: `train_adapters`)."""
        training = self.training if training is None else training
        if x16.shape[-1] != self.n_channels:
            raise ValueError(f"UNet: input has {x16.shape[-1]} channels, built for {self.n_channels}")
        split = x_lo is not None
        sv = {}
        x3 = (x16, x_lo)
        p3h, p3l, sv["idx3"] = ops.maxpool2_fwd(x3[0], x3[1], save_idx=save and need_input_grad)   # frozen input: no arg-max kept
        sv["x3_shape"] = x16.shape
        x4, sv["down3"] = self._dconv_fwd("d3", self.down3.maxpool_conv[1], (p3h, p3l), save, training)
        p4h, p4l, sv["idx4"] = ops.maxpool2_fwd(x4[0], x4[1], save_idx=save)
        x5, sv["down4"] = self._dconv_fwd("d4", self.down4.maxpool_conv[1], (p4h, p4l), save, training)
        y, sv["up1.up"] = self._up_fwd("u1", self.up1.up, x5, x4, save)
        y, sv["up1"] = self._dconv_fwd("u1c", self.up1.conv, y, save, training)
        y, sv["up2.up"] = self._up_fwd("u2", self.up2.up, y, x3, save)
        y, sv["up2"] = self._dconv_fwd("u2c", self.up2.conv, y, save, training)
        y, sv["up3.up"] = self._up_fwd("u3", self.up3.up, y, None, save)
        y, sv["up3"] = self._dconv_fwd("u3c", self.up3.conv, y, save, training)
        y, sv["up4.up"] = self._up_fwd("u4", self.up4.up, y, None, save)
        y, sv["up4"] = self._dconv_fwd("u4c", self.up4.conv, y, save, training)
        logits = self._outc_fwd(y)
        sv["x_last"] = y[0] if save else None
        sv["x4_shape"] = x4[0].shape
        return logits, sv

    def _backward_core(self, saved, d16, bias_partial, inv_scale, grads, dlogits_f32=None, stage_done=None, d_lo=None,
                       need_input_grad: bool = False):
        """d16 (+ d_lo): 16-bit [B,H,W,CP] = loss_scale * dL/dlogits (pad channels zero); same contract as
        ``FeatureDecoder._backward_core``.  ``need_input_grad`` (forward run with it too): returns loss_scale * dL/d(input)
        fp32 NHWC = the skip gradient out of up2's concat + the MaxPool transpose of down3's input gradient."""
        def done():
            if stage_done is not None:
                stage_done()

        dU = self._outc_bwd(saved["x_last"], d16, bias_partial, inv_scale, grads, dlogits_f32, d_lo)
        done()
        dcat = self._dconv_bwd("u4c", self.up4.conv, saved["up4"], dU, inv_scale, grads, "up4.conv", True)
        dU = self._up_bwd("u4", self.up4.up, saved["up4.up"], dcat, inv_scale, grads, "up4.up")
        done()
        dcat = self._dconv_bwd("u3c", self.up3.conv, saved["up3"], dU, inv_scale, grads, "up3.conv", True)
        dU = self._up_bwd("u3", self.up3.up, saved["up3.up"], dcat, inv_scale, grads, "up3.up")
        done()
        dcat = self._dconv_bwd("u2c", self.up2.conv, saved["up2"], dU, inv_scale, grads, "up2.conv", True)
        dU = self._up_bwd("u2", self.up2.up, saved["up2.up"], dcat, inv_scale, grads, "up2.up")
        g3 = None
        if need_input_grad:    # x3 = the head's input: skip gradient = leading channels of up2's concat
            B3, H3, W3, C3 = saved["x3_shape"]
            g3 = torch.empty((B3, H3, W3, C3), device=d16.device, dtype=torch.float32)
            ops.copy_channels(dcat.view(-1, dcat.shape[-1])[:, :C3], g3.view(-1, C3))
        done()
        dcat = self._dconv_bwd("u1c", self.up1.conv, saved["up1"], dU, inv_scale, grads, "up1.conv", True)
        dU = self._up_bwd("u1", self.up1.up, saved["up1.up"], dcat, inv_scale, grads, "up1.up")
        done()
        # x4 receives the skip gradient (leading channels of up1's concat) plus the MaxPool transpose of down4's input
        Bq, H4, W4, C4 = saved["x4_shape"]
        g4 = torch.empty((Bq, H4, W4, C4), device=d16.device, dtype=torch.float32)
        ops.copy_channels(dcat.view(-1, dcat.shape[-1])[:, :C4], g4.view(-1, C4))
        dP = self._dconv_bwd("d4", self.down4.maxpool_conv[1], saved["down4"], dU, inv_scale, grads,
                             "down4.maxpool_conv.1", True)
        ops.maxpool2_bwd(dP, saved["idx4"], g4)
        done()
        dP3 = self._dconv_bwd("d3", self.down3.maxpool_conv[1], saved["down3"], g4, inv_scale, grads, "down3.maxpool_conv.1",
                              need_input_grad)
        done()
        if need_input_grad:
            ops.maxpool2_bwd(dP3, saved["idx3"], g3)
        return g3

    # ---- reference-shaped entry point ------------------------------------------------------------------------------
    def forward(self, x):
        """`unet_parts.py:126-137`: (B, C, h, w) fp32 -> logits (B, classes, 4h, 4w) fp32."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _DecoderFn.apply(self, x, *list(self.parameters()))
        logits, _ = self._forward_core(*self._to_nhwc16(x), save=False)
        return logits.permute(0, 3, 1, 2)
