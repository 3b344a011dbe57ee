"""HIP-backed mirror of `segloss/ND_Crossentropy.py:11-32`."""
from __future__ import annotations

from torch import nn

from .. import ops
from .dice import seg_loss


class CrossentropyND(nn.Module):
    """Mean (optionally class-weighted) cross entropy over all pixels of NCHW logits ("network has to have NO
    NONLINEARITY")."""

    def __init__(self, weight=None):
        super().__init__()
        self.weight = weight

    def forward(self, inp, target):
        return seg_loss(inp, target, 0, ops.LOSS_NONE, 0.0, n_ce=1, ce_weight=self.weight)
