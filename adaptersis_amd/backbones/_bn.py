"""Train-mode BatchNorm plumbing shared by the CNN encoder and the decode heads.

Statistics come either from the implicit-GEMM epilogue partials or from ``asis_colstats``; they are
reduced in double, all-reduced across ranks when ``sync`` is set (= ``nn.SyncBatchNorm``,
`backbones/encoders.py:12-40`; 2C+1 doubles per layer) and finalised on the device.  Nothing here
synchronises the host.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import ops, parallel


def finalize(partial: torch.Tensor, count: int, bn: torch.nn.Module, sync: bool = False, update: bool = True):
    """partial fp32 [nparts, 2, C] -> (scale, shift, mean, invstd, total_count)."""
    sums = ops.reduce_partials(partial)
    total = float(count)
    if sync and parallel.bn_collectives_on():
        dist.all_reduce(sums)  # 2C doubles
        total = float(count) * parallel.count_scale()   # x world, or x (global / local batch) after parallel.set_batch_ratio
    rm = bn.running_mean if (update and bn.track_running_stats) else None
    rv = bn.running_var if (update and bn.track_running_stats) else None
    nbt = bn.num_batches_tracked if (update and bn.track_running_stats) else None
    mom = bn.momentum if bn.momentum is not None else 0.1
    w = bn.weight.detach() if bn.weight is not None else None
    b = bn.bias.detach() if bn.bias is not None else None
    scale, shift, mean, invstd = ops.bn_finalize(sums, total, w, b, bn.eps, mom, rm, rv, nbt)
    return scale, shift, mean, invstd, total
