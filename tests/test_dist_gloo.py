"""CPU, world_size 2, gloo: the data-parallel plumbing (flat gradient bucket, per-stage all-reduce with the
1/world pre-scaling, optimizer state) that the engine drives with RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd.optim import FlatBucket
    from adaptersis_amd.parallel import StageReducer, world_size
    torch.manual_seed(0)
    lin1, lin2 = torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)
    named = [("b.weight", lin2.weight), ("b.bias", lin2.bias), ("a.weight", lin1.weight), ("a.bias", lin1.bias)]
    w0 = [p.detach().clone() for _, p in named]
    bucket = FlatBucket(named)
    assert all(torch.equal(p.detach(), w) for (_, p), w in zip(named, w0)), "re-pointing must keep values"
    ranges = [bucket.range_of(["b.weight", "b.bias"]), bucket.range_of(["a.weight", "a.bias"])]
    red = StageReducer(bucket.grad, ranges)
    assert world_size() == world
    # rank-local "gradients", already divided by world like the backward kernels do
    for i, (_, p) in enumerate(named):
        p.grad.copy_(torch.full_like(p, float((rank + 1) * (i + 1))) / world)
    red.begin()
    red.stage_done()
    red.stage_done()
    red.finish()
    expect = [sum((r + 1) * (i + 1) for r in range(world)) / world for i in range(4)]
    ok = all(torch.allclose(p.grad, torch.full_like(p, e)) for (_, p), e in zip(named, expect))
    out[rank] = bool(ok) and bucket.grad.is_contiguous() and bucket.numel % 4 == 0
    dist.destroy_process_group()


def test_stage_reducer_two_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}
