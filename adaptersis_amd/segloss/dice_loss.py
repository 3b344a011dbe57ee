"""HIP-backed mirror of `segloss/dice_loss.py` (the region losses the training scripts import:
`train.py:50`, `train_mla.py:50`).  tp/fp/fn of `get_tp_fp_fn` (`dice_loss.py:31-81`) are functions of the three sums
the fused kernel keeps per (b, c): tp = I, fp = Sp - I, fn = St - I — so every loss here is one forward pass over the
logits and one backward pass (``asis_seg_loss_fwd/bwd``).  Options the scripts never set (``batch_dice``,
``do_bg=False``, ``square``, ``loss_mask``) are rejected instead of silently ignored."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .dice import seg_loss


def softmax_helper(x):
    """Marker for ``apply_nonlin`` (nnU-Net's ``softmax_helper``): the kernel applies the softmax itself."""
    return torch.softmax(x, 1)


def _nonlin_count(apply_nonlin) -> int:
    if apply_nonlin is None:
        return 0
    if apply_nonlin is softmax_helper or isinstance(apply_nonlin, nn.Softmax):
        return 1
    raise NotImplementedError("apply_nonlin must be None, softmax_helper or nn.Softmax(1)")


class _RegionLoss(nn.Module):
    MODE = ops.LOSS_SOFTDICE

    def __init__(self, apply_nonlin=None, batch_dice=False, do_bg=True, smooth=1., square=False):
        super().__init__()
        if batch_dice or not do_bg or square:
            raise NotImplementedError("batch_dice / do_bg=False / square are not built (never set by the training scripts)")
        self.apply_nonlin, self.batch_dice, self.do_bg, self.smooth, self.square = apply_nonlin, batch_dice, do_bg, smooth, square
        self._n = _nonlin_count(apply_nonlin)

    def forward(self, x, y, loss_mask=None):
        if loss_mask is not None:
            raise NotImplementedError("loss_mask is not built")
        return seg_loss(x, y, self._n, self.MODE, float(self.smooth))


class SoftDiceLoss(_RegionLoss):
    """`dice_loss.py:255-291`: ``-mean (2tp+s)/(2tp+fp+fn+s)``; ``x`` is used as given unless ``apply_nonlin``."""
    MODE = ops.LOSS_SOFTDICE


class TverskyLoss(_RegionLoss):
    """`dice_loss.py:333-372`: ``-mean (tp+s)/(tp+.3fp+.7fn+s)``."""
    MODE = ops.LOSS_TVERSKY

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.alpha, self.beta = 0.3, 0.7


class DC_and_CE_loss(nn.Module):
    """`dice_loss.py:445-459`: ``CrossentropyND()(x, t) + SoftDiceLoss()(x, t)`` in one fused pass."""

    def __init__(self, aggregate="sum"):
        super().__init__()
        if aggregate != "sum":
            raise NotImplementedError("nah son")
        self.aggregate = aggregate

    def forward(self, net_output, target):
        return seg_loss(net_output, target, 0, ops.LOSS_SOFTDICE, 1.0, n_ce=1)
