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


def _worker_bcast(rank, world, port, out):
    """modules built with torch's default per-process init must leave `train_seg` identical on all ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd.backbones.adapter_blocks import CACNN, CAViT
    from adaptersis_amd.backbones.decoders import FeatureDecoder
    from adaptersis_amd.backbones.encoders import FeatureEncoder
    from adaptersis_amd.optim import FlatBucket
    from adaptersis_amd.train import broadcast_module_states
    torch.manual_seed(100 + rank)        # what separate processes get: different random initialisations
    D = 64
    mods = [FeatureEncoder(embed_dim=D), CAViT(dim=D, n_levels=3, num_heads=8, init_values=0.0, n_points=4),
            CACNN(dim=D, n_levels=1, num_heads=8, n_points=4, with_cffn=True, cffn_ratio=0.25),
            FeatureDecoder(embed_dim=D, num_classes=2, features=[D, 32, 16, 16, 8])]
    mods[0].stem[1].num_batches_tracked.fill_(rank + 3)   # integer buffer
    before = torch.cat([p.detach().reshape(-1) for p in mods[3].parameters()]).clone()
    broadcast_module_states(mods)
    bucket = FlatBucket([(f"{i}.{n}", p) for i, m in enumerate(mods) for n, p in m.named_parameters()])
    bufs = torch.cat([b.detach().double().reshape(-1) for m in mods for b in m.buffers()])
    gathered = [torch.empty_like(bucket.flat) for _ in range(world)]
    dist.all_gather(gathered, bucket.flat)
    gb = [torch.empty_like(bufs) for _ in range(world)]
    dist.all_gather(gb, bufs)
    same = all(torch.equal(g, gathered[0]) for g in gathered) and all(torch.equal(g, gb[0]) for g in gb)
    changed = not torch.equal(before, torch.cat([p.detach().reshape(-1) for p in mods[3].parameters()]))
    out[rank] = bool(same) and (changed if rank > 0 else not changed) and int(mods[0].stem[1].num_batches_tracked) == 3
    dist.destroy_process_group()


def test_broadcast_module_states_two_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_bcast, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def _worker_compress(rank, world, port, out):
    """bf16 transport of a gradient bucket (StageReducer(compress="bf16")): same protocol as the fp32 exchange,
    result within bf16's rounding of the exact mean, tiny magnitudes (the unscaled Dice gradients) survive."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd.parallel import StageReducer
    g = torch.Generator().manual_seed(7 + rank)
    n = 4096
    flat = (torch.randn(n, generator=g) * torch.logspace(-9, 0, n)) / world      # magnitudes 1e-9 .. 1, pre-divided by world
    exact = flat.clone()
    dist.all_reduce(exact)
    ranges = [(0, 1024), (1024, 4096)]
    red = StageReducer(flat, ranges, compress="bf16")
    red.begin(); red.stage_done(); red.stage_done(); red.finish()
    rel = ((flat - exact).abs() / (exact.abs() + 1e-30))
    # each rank's term rounds to 8 significant bits (2^-9 relative to ITS magnitude) and the bf16 sum rounds once more; where the
    # two terms nearly cancel the error relative to the small SUM is larger, so the bound is on the error relative to the terms
    gathered = [torch.empty(n) for _ in range(world)]
    mine = (torch.randn(n, generator=torch.Generator().manual_seed(7 + rank)) * torch.logspace(-9, 0, n)) / world
    dist.all_gather(gathered, mine)
    scale = sum(t.abs() for t in gathered)
    ok = bool(((flat - exact).abs() <= scale * 2.0 ** -7).all()) and bool((flat != 0).sum() > n * 0.99)
    same = [torch.empty(n) for _ in range(world)]
    dist.all_gather(same, flat)
    out[rank] = ok and all(torch.equal(same[0], t) for t in same)     # every rank holds the same reduced values
    dist.destroy_process_group()


def test_bf16_compressed_reducer_two_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_compress, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def _worker_shard_val(rank, world, port, out):
    """--shard_val: the ranks' batches partition the reference's sequential batches; inside
    ``parallel.local_batchnorm()`` SyncBatchNorm layers issue no collective although a process group is up; the metric sums of the
    shards all-reduce to the whole-set averages that every rank of the reference computes redundantly."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd import parallel
    from adaptersis_amd.train import BatchShardSampler
    from adaptersis_amd.utils.misc import MetricLogger
    n, bs = 53, 12                                            # 5 batches (12, 12, 12, 12, 5): rank 0 gets 3, rank 1 gets 2
    mine = list(BatchShardSampler(n, bs, rank, world))
    ref = [list(range(i, min(i + bs, n))) for i in range(0, n, bs)]
    ok = mine == ref[rank::world]
    assert parallel.collectives_on() and parallel.bn_collectives_on()
    with parallel.local_batchnorm():
        ok = ok and parallel.collectives_on() and not parallel.bn_collectives_on()
    ok = ok and parallel.bn_collectives_on()
    per_image = torch.arange(n, dtype=torch.float64) * 0.5 + 1.0           # a per-image metric
    ml = MetricLogger()
    for b in mine:                                            # different batch counts per rank: no collective in the loop
        ml.meters["acc1"].update(float(per_image[b].mean()), n=len(b))
    ml.synchronize_between_processes()
    ok = ok and abs(ml.meters["acc1"].global_avg - float(per_image.mean())) < 1e-12 and ml.meters["acc1"].count == n
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_sharded_validation_two_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_shard_val, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def _worker_shard_val_empty(rank, world, port, out):
    """fewer validation batches than ranks.  The rank with an empty shard creates no meter inside the loop; with the
    meters of ``train._val_meters`` made up front every rank issues the same three barrier + all-reduce pairs and the summary
    line formats on all of them (before the fix: ranks 0 waits in a barrier the empty rank never enters)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from adaptersis_amd.train import BatchShardSampler, _val_meters, _val_summary
    n, bs = 7, 12                                             # ONE batch of 7 images: rank 0 gets it, rank 1 nothing
    mine = list(BatchShardSampler(n, bs, rank, world))
    ok = len(mine) == (1 if rank == 0 else 0)
    ml = _val_meters()
    for b in mine:
        ml.update(loss=0.25)
        ml.meters["acc1"].update(0.5, n=len(b))
        ml.meters["dice"].update(0.75, n=len(b))
    ml.synchronize_between_processes()
    ok = ok and ml.meters["acc1"].count == n and abs(ml.meters["acc1"].global_avg - 0.5) < 1e-12
    ok = ok and abs(ml.meters["dice"].global_avg - 0.75) < 1e-12 and abs(ml.meters["loss"].global_avg - 0.25) < 1e-12
    ok = ok and _val_summary(ml) == "* Acc@1 0.500 loss 0.250 Dice 0.750"
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_sharded_validation_fewer_batches_than_ranks_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_shard_val_empty, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}
