"""HIP-backed Dice loss — API mirror of `segloss/dice.py:5-36` plus the fused training loss of
`train.py:422-428` (bilinear resize -> softmax -> DC, which applies softmax again).

Forward and backward are single fused HIP passes over the NHWC logits (``asis_dice_fwd`` /
``asis_dice_bwd`` / ``asis_resize_bilinear_bwd``); nothing of size B*C*H*W is materialised in the
forward.  ``DC`` keeps the reference's call convention (``output`` NCHW, ``target`` one-hot of the
same shape or label map (B,1,H,W)) and is differentiable through torch autograd.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


def _labels(target: torch.Tensor, out_shape) -> torch.Tensor:
    """label map (B,1,H,W)/(B,H,W) or one-hot (B,C,H,W) -> int64 [B,H,W] (one-hot -> argmax)."""
    if target.dim() == len(out_shape) and all(int(i) == int(j) for i, j in zip(out_shape, target.shape)) \
            and target.shape[1] != 1:
        return target.argmax(1).long().contiguous()
    if target.dim() == 4:
        target = target[:, 0]
    return target.long().contiguous()


class _SegLossFn(torch.autograd.Function):
    """loss(logits NCHW, labels [B,H,W]) through ``asis_seg_loss_fwd/bwd`` (+ the resize transpose when H,W differ)."""

    @staticmethod
    def forward(ctx, logits_nchw, labels, n_region, mode, eps, n_ce, ce_weight):
        lg = logits_nchw.detach().permute(0, 2, 3, 1).contiguous().float()  # no copy when it is an NHWC buffer view
        loss, coef, _ = ops.seg_loss_fwd(lg, labels, n_region, mode, eps, n_ce, ce_weight, 1.0)
        ctx.lg, ctx.labels, ctx.coef, ctx.cfg, ctx.ce_weight = lg, labels, coef, (n_region, mode, n_ce), ce_weight
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        lg = ctx.lg
        B, h, w, C = lg.shape
        n_region, mode, n_ce = ctx.cfg
        dz = ops.seg_loss_bwd(lg, ctx.labels, ctx.coef, n_region, mode, n_ce, ctx.ce_weight)
        H, W = ctx.labels.shape[-2:]
        if (H, W) != (h, w):
            dz, _ = ops.resize_bilinear_bwd(dz, h, w, torch.float32)
        return dz.permute(0, 3, 1, 2) * gout, None, None, None, None, None, None


def seg_loss(logits: torch.Tensor, target: torch.Tensor, n_region: int, mode: int, eps: float, n_ce: int = 0,
             ce_weight=None) -> torch.Tensor:
    """Generic fused segmentation loss on NCHW logits (resized to the target's H x W when they differ)."""
    if ce_weight is not None:
        ce_weight = ce_weight.detach().float().contiguous().to(logits.device)
    return _SegLossFn.apply(logits, _labels(target, logits.shape), n_region, mode, eps, n_ce, ce_weight)


class _DiceFn:
    @staticmethod
    def apply(logits_nchw, labels, n_softmax, eps):
        return _SegLossFn.apply(logits_nchw, labels, n_softmax, ops.LOSS_DICE, eps, 0, None)


def resize_softmax_dc(logits: torch.Tensor, target: torch.Tensor, n_softmax: int = 2, eps: float = 10e-20):
    """`train.py:422-428` fused: F.interpolate(logits, target HxW, bilinear) -> softmax -> DC (softmax again).
    logits (B,C,h,w) fp32, target (B,H,W) labels.  n_softmax=1 gives the validation dice term (`train.py:618`)."""
    return _DiceFn.apply(logits, _labels(target, logits.shape), n_softmax, eps)


class DC(nn.Module):
    def __init__(self, nb_classes):
        super().__init__()
        self.softmax = nn.Softmax(1)
        self.nb_classes = nb_classes

    def dice(self, output, target):
        return _DiceFn.apply(output, _labels(target, output.shape), 1, 10e-20)

    def forward(self, output, target):
        return self.dice(output, target)
