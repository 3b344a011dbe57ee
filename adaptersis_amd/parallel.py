"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm); images are independent, so the only exchange steps are the
gradient all-reduce of the trainable parameters and the tiny BatchNorm statistic sums.

``StageReducer`` all-reduces contiguous ranges of the flat gradient bucket as soon as the backward pass
has produced them (final conv first, decoder_1 last), on a side stream so the transfers overlap the
remaining backward kernels.  Gradients are pre-divided by the world size by the producing kernels (the
1/world factor is folded into the loss-scale removal), so a SUM all-reduce yields DDP's mean.
The 8 GPUs of an MI355X node are a full xGMI mesh (7 links x ~153 GB/s per GPU); the decoder-only bucket
is 62.8 MB fp32, ~0.7 ms on a ring, against >100 ms of compute per step.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


# Rehearsal switch (tests/test_gpu_rccl.py): issue every collective of the step even in a 1-rank process group, so the RCCL
# calls themselves — backend init on the device, fp32 / fp64 all-reduces from the compute stream and from side streams —
# run on a one-GPU box, where they are the identity.  Never set in production.
FORCE_COLLECTIVES = False


def collectives_on(group=None) -> bool:
    """True when the data-parallel exchanges have to be issued (more than one rank, or the rehearsal switch)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


class StageReducer:
    def __init__(self, flat_grad: torch.Tensor, ranges: Sequence[Tuple[int, int]], group=None):
        self.flat, self.ranges, self.group = flat_grad, list(ranges), group
        self._next = 0
        self._handles: List = []
        self._stream: Optional[torch.cuda.Stream] = None

    def begin(self):
        self._next = 0
        self._handles = []

    def stage_done(self):
        """Call after the kernels writing the next range have been enqueued on the current stream."""
        lo, hi = self.ranges[self._next]
        self._next += 1
        if not collectives_on(self.group):
            return
        chunk = self.flat[lo:hi]
        if chunk.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            ev = torch.cuda.Event()
            ev.record()
            self._stream.wait_event(ev)
            with torch.cuda.stream(self._stream):
                dist.all_reduce(chunk, group=self.group)
        else:  # CPU tensors (gloo): asynchronous work handles
            self._handles.append(dist.all_reduce(chunk, group=self.group, async_op=True))

    def finish(self):
        """Make the reduced gradients visible to the optimizer kernel on the current stream."""
        assert self._next == len(self.ranges), "not every stage was reduced"
        for h in self._handles:
            h.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
