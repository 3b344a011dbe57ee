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


def _worker_multi(rank, world, port, out):
    """The bucket layout of the train_adapters / end-to-end engines: an optimised bucket with per-stage ranges, a second
    optimised bucket reduced in one piece, and a gradient-only bucket (no momentum) reduced in chunks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd.optim import FlatBucket
    from adaptersis_amd.parallel import StageReducer
    torch.manual_seed(1)
    dec = torch.nn.Linear(4, 4)
    ada = torch.nn.Linear(4, 2)
    vit = torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Linear(6, 6), torch.nn.Linear(6, 6))
    b_dec = FlatBucket([("dec.weight", dec.weight), ("dec.bias", dec.bias)])
    b_ada = FlatBucket([("ada.weight", ada.weight), ("ada.bias", ada.bias)])
    names = [(f"{i}.{n}", p) for i in (2, 1, 0) for n, p in vit[i].named_parameters()]   # gradient-ready order
    b_vit = FlatBucket(names, momentum=False)
    assert b_vit.momentum is None and b_dec.momentum is not None
    r_dec = StageReducer(b_dec.grad, [(0, b_dec.numel)])
    r_ada = StageReducer(b_ada.grad, [(0, b_ada.numel)])
    r_vit = StageReducer(b_vit.grad, [b_vit.range_of(["2.weight", "2.bias", "1.weight", "1.bias"]),
                                      b_vit.range_of(["0.weight", "0.bias"])])
    for b in (b_dec, b_ada, b_vit):
        b.grad.fill_(float(rank + 1) / world)
    for r in (r_dec, r_ada, r_vit):
        r.begin()
    r_dec.stage_done(); r_vit.stage_done(); r_ada.stage_done(); r_vit.stage_done()
    for r in (r_dec, r_ada, r_vit):
        r.finish()
    mean = sum(r + 1 for r in range(world)) / world
    ok = all(torch.allclose(b.grad, torch.full_like(b.grad, mean)) for b in (b_dec, b_ada, b_vit))
    # identical parameters after the same reduced gradients (the gradient-only bucket is never stepped)
    from adaptersis_amd import optim
    try:
        optim.SGD([b_dec, b_ada], lr=0.1, momentum=0.9)
        ok = ok and True
    except Exception:
        ok = False
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_multi_bucket_reducers_two_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_multi, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}
