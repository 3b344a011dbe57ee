"""Flat fp32 parameter / gradient buckets and the HIP SGD step.

``FlatBucket`` re-points the parameters of the trainable modules at views of ONE contiguous fp32
buffer (and their ``.grad`` at views of a second one), ordered so that gradients become complete
front-to-back during the backward pass.  That gives: one ``asis_sgd_momentum`` launch per step, RCCL
all-reduces over contiguous ranges with no packing copies, and ``state_dict`` / checkpoint
compatibility (parameters stay ``nn.Parameter`` objects with their reference key names).

``SGD`` mirrors ``torch.optim.SGD`` as configured in `train.py:178-191` / `train_mla.py:178-183`
(momentum, weight decay, dampening 0, no Nesterov; ``param_groups[i]["lr"]`` is what
``CosineAnnealingLR`` mutates).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import torch
from torch import nn

from . import ops


class FlatBucket:
    def __init__(self, named_params: Sequence[Tuple[str, nn.Parameter]], momentum: bool = True):
        named_params = [(n, p) for n, p in named_params if p.requires_grad]
        if not named_params:
            raise ValueError("FlatBucket: no trainable parameters")
        dev = named_params[0][1].device
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        sizes = [p.numel() for p in self.params]
        # 16-byte align every tensor inside the bucket
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += (s + 3) // 4 * 4
        self.numel = o
        self.offsets = offs
        self.flat = torch.zeros(o, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(o, device=dev, dtype=torch.float32)
        # gradient-only buckets (parameters that are all-reduced but never optimised) carry no momentum buffer
        self.momentum = torch.zeros(o, device=dev, dtype=torch.float32) if momentum else None
        with torch.no_grad():
            for p, off in zip(self.params, offs):
                v = self.flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.detach().float())
                p.data = v
                p.grad = self.grad[off:off + p.numel()].view(p.shape)
        self.views: Dict[str, torch.Tensor] = {n: p.grad for n, p in zip(self.names, self.params)}

    def range_of(self, names: Iterable[str]) -> Tuple[int, int]:
        idx = [self.names.index(n) for n in names]
        lo = min(self.offsets[i] for i in idx)
        hi = max(self.offsets[i] + (self.params[i].numel() + 3) // 4 * 4 for i in idx)
        return lo, hi


class SGD:
    """``torch.optim.SGD``-shaped optimizer over FlatBuckets (one kernel launch per bucket and step)."""

    def __init__(self, buckets: Sequence[FlatBucket], lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        self.buckets = list(buckets)
        self.param_groups: List[dict] = [{"params": b.params, "lr": lr, "initial_lr": lr, "momentum": momentum,
                                          "weight_decay": weight_decay, "dampening": 0, "nesterov": False}
                                         for b in self.buckets]
        self._steps = 0
        # overflow guard of the static loss scale (config.loss_scale, 16-bit gradient tensors): a step whose all-reduced
        # gradients hold an inf / NaN is skipped on the device — no host sync; ``skipped_steps`` reads the counter
        self.guard = torch.zeros(2, device=self.buckets[0].flat.device, dtype=torch.int32) \
            if self.buckets and self.buckets[0].flat.is_cuda else None

    @property
    def skipped_steps(self) -> int:
        return 0 if self.guard is None else int(self.guard[1].item())

    def zero_grad(self, set_to_none: bool = False):
        pass  # every gradient element is overwritten by the backward kernels each step

    def step(self, inv_scale: float = 1.0):
        if self.guard is not None:
            for i, b in enumerate(self.buckets):
                ops.grad_guard(b.grad, self.guard, i == 0)
        for i, (b, g) in enumerate(zip(self.buckets, self.param_groups)):
            ops.sgd_momentum(b.flat, b.grad, b.momentum, g["lr"], g["momentum"], g["weight_decay"], inv_scale,
                             self._steps == 0, self.guard, count_skip=(i == 0))
            for p in b.params:  # changed in place behind torch's back: invalidate the packed 16-bit copies
                p._asis_gen = getattr(p, "_asis_gen", 0) + 1
        self._steps += 1

    def state_dict(self):
        return {"state": {i: {"momentum_buffer": b.momentum.clone()} for i, b in enumerate(self.buckets)},
                "steps": self._steps,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        # validate first (a `torch.optim.SGD` entry of a reference checkpoint holds per-parameter buffers): nothing is
        # modified when the entry does not fit, and `restart_from_checkpoint` reports + skips it
        state = sd.get("state", {})
        bufs = []
        for i, b in enumerate(self.buckets):
            ent = state.get(i, state.get(str(i)))
            mb = ent.get("momentum_buffer") if isinstance(ent, dict) else None
            if not torch.is_tensor(mb) or mb.numel() != b.numel:
                raise ValueError(f"optimizer state does not match flat bucket {i} ({b.numel} elements): not written by "
                                 "adaptersis_amd.optim.SGD with the same trainable set")
            bufs.append(mb)
        if len(sd.get("param_groups", [])) != len(self.param_groups):
            raise ValueError("optimizer state has a different number of parameter groups")
        for b, mb in zip(self.buckets, bufs):
            b.momentum.copy_(mb.reshape(-1))
        self._steps = int(sd.get("steps", 1))
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
