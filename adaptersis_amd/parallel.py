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


# SyncBatchNorm layers ask ``bn_collectives_on`` (not ``collectives_on``): inside ``local_batchnorm()`` they take LOCAL batch
# statistics although a process group is up.  The sharded validation of ``train.validate_network`` (--shard_val) needs it: the
# reference lets every rank evaluate the whole val set (`train.py:125-130,236-238`), so its encoder SyncBatchNorm sees world
# copies of ONE batch — the same statistics as that batch alone; when the ranks evaluate DIFFERENT batches, local statistics
# keep every batch's result identical to the reference's (and ranks with fewer batches issue no unmatched collective).
_LOCAL_BN = 0


class local_batchnorm:
    """context manager: train-mode (Sync)BatchNorm layers use this rank's statistics only"""

    def __enter__(self):
        global _LOCAL_BN
        _LOCAL_BN += 1
        return self

    def __exit__(self, *exc):
        global _LOCAL_BN
        _LOCAL_BN -= 1
        return False


def bn_collectives_on(group=None) -> bool:
    return _LOCAL_BN == 0 and collectives_on(group)


# SyncBatchNorm element counts (`nn.SyncBatchNorm` gathers every rank's count: `backbones/encoders.py:12-40`).  With the
# reference's loader (`DistributedSampler`, `train.py:167-175`) every rank holds the same number of images in every iteration —
# also in the short last batch of an epoch — so a layer's global count is its local count x world, known on the host without an
# exchange.  A caller whose per-rank batches can differ announces it once per step with ``set_batch_ratio``: the ranks'
# batch sizes are summed (one int all-reduce and a host read — a host sync, which is why it is opt-in) and every train-mode
# BatchNorm of that step divides its all-reduced sums by local count x (global batch / local batch).
_BATCH_RATIO = None


def set_batch_ratio(local_batch=None, group=None) -> float:
    """``local_batch`` = this rank's image count of the coming step (None: back to equal batches) -> global / local."""
    global _BATCH_RATIO
    if local_batch is None or not collectives_on(group):
        _BATCH_RATIO = None
        return float(world_size(group))
    backend = dist.get_backend(group)
    t = torch.tensor([int(local_batch)], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, group=group)
    _BATCH_RATIO = float(int(t.item())) / float(local_batch)
    return _BATCH_RATIO


def count_scale(group=None) -> float:
    """global element count / local element count of a SyncBatchNorm layer in the current step."""
    return _BATCH_RATIO if _BATCH_RATIO is not None else float(world_size(group))


_GRAD_STREAMS = {}


def grad_side_stream(device) -> "torch.cuda.Stream":
    """The side stream on which parameter-gradient kernels that are off the backward's critical path run (config.wgrad_stream).
    Everything that consumes gradients — StageReducer (all-reduce), the optimizer behind ``finish``, autograd bridges through
    ``join_grad_streams`` — orders itself behind it."""
    key = (device.type, device.index)
    if key not in _GRAD_STREAMS:
        _GRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return _GRAD_STREAMS[key]


def join_grad_streams(stream=None) -> None:
    """``stream`` (default: the current one) waits for every gradient side stream."""
    if not _GRAD_STREAMS:
        return
    stream = stream if stream is not None else torch.cuda.current_stream()
    for s in _GRAD_STREAMS.values():
        stream.wait_stream(s)


def wgrad_on_side_stream(fn, *tensors) -> None:
    """Run ``fn()`` (a weight-gradient launch reading ``tensors``) on the gradient side stream when ``config.wgrad_stream`` is on
    and the operands live on a GPU, else in place.  The side stream first waits for the compute stream (the operands), and the
    operands are recorded on it so that the caching allocator does not hand their memory out before the launch has read it."""
    from . import config
    t0 = tensors[0]
    if not (config.wgrad_stream and t0.is_cuda):
        fn()
        return
    side = grad_side_stream(t0.device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    for t in tensors:
        t.record_stream(side)


class StageReducer:
    """``compress="bf16"``: a range travels as bfloat16 (half the bytes: the 1.2 GB fp32 backbone bucket of BASELINE config 4 is
    ~15 ms on an xGMI ring) — packed into a staging buffer right before its all-reduce, summed by the
    collective in bf16, unpacked into the fp32 bucket behind it, all on the reducer's side stream.  bf16 keeps fp32's exponent
    range (the unscaled ~1e-7 Dice gradients survive) at 8 significant bits: a relative error <= 2^-9 per rank term, i.e. the
    torch-side `bf16_compress_hook`.  SegEngine uses it for the backbone bucket of the unfrozen flow only (default since round 5;
    ``ASIS_GRAD_COMPRESS=none`` / ``SegEngine(grad_compress="none")``: fp32 like the reference's DDP, `train.py:84-116`); the
    two-rank error bound is pinned in tests/test_dist_gloo.py and tests/test_gpu_rccl.py."""

    def __init__(self, flat_grad: torch.Tensor, ranges: Sequence[Tuple[int, int]], group=None, compress: Optional[str] = None):
        if compress not in (None, "bf16"):
            raise ValueError("StageReducer: compress must be None or 'bf16'")
        self.flat, self.ranges, self.group, self.compress = flat_grad, list(ranges), group, compress
        self._next = 0
        self._handles: List = []
        self._stream: Optional[torch.cuda.Stream] = None
        self._staging: Optional[torch.Tensor] = None      # bf16 transport buffer, as long as the longest range

    def begin(self):
        self._next = 0
        self._handles = []

    def _stage_buf(self, n: int) -> torch.Tensor:
        if self._staging is None:
            longest = max(hi - lo for lo, hi in self.ranges)
            self._staging = torch.empty(longest, device=self.flat.device, dtype=torch.bfloat16)
        return self._staging[:n]

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
            join_grad_streams(self._stream)      # weight gradients of this stage enqueued on the side stream
            with torch.cuda.stream(self._stream):
                if self.compress:
                    from . import ops
                    # one staging buffer per reducer: ranges are reduced one after the other on this one stream
                    buf = self._stage_buf(hi - lo)
                    ops.grad_pack_bf16(chunk, buf)
                    dist.all_reduce(buf, group=self.group)
                    ops.grad_unpack_bf16(buf, chunk)
                else:
                    dist.all_reduce(chunk, group=self.group)
        elif self.compress:  # CPU tensors (gloo rehearsal of the same protocol): blocking, ranges share the staging buffer
            buf = self._stage_buf(hi - lo)
            buf.copy_(chunk)
            dist.all_reduce(buf, group=self.group)
            chunk.copy_(buf)
        else:  # CPU tensors (gloo): asynchronous work handles
            self._handles.append(dist.all_reduce(chunk, group=self.group, async_op=True))

    def finish(self):
        """Make the reduced gradients visible to the optimizer kernel on the current stream."""
        assert self._next == len(self.ranges), "not every stage was reduced"
        for h in self._handles:
            h.wait()
        if self.flat.is_cuda:
            join_grad_streams()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
