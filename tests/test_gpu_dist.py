"""GPU, 2 processes sharing the one card, gloo backend (RCCL needs one device per rank and multi-GPU runs belong to
the driver): the engine's data-parallel step — SyncBatchNorm statistic exchange in the encoder, per-stage gradient
all-reduce on the side stream, 1/world scaling — leaves both ranks with identical decoder weights that equal the
single-process result obtained by averaging the two ranks' local gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd.utils import weights as W
    from tests.test_gpu_step import build_engine
    dev = torch.device("cuda:0")
    eng, _ = build_engine("vit_tiny_test", "kernel", dev, (128, 32, 16, 16, 8), lr=0.05)
    img, tgt = W.synthetic_batch(4, 224, seed=7)
    sl = slice(rank * 2, rank * 2 + 2)  # DistributedSampler-style shard
    loss = eng.train_step(img[sl].to(dev), tgt[sl].to(dev))
    torch.cuda.synchronize()
    out[rank] = {"loss": float(loss), "grad": eng.bucket.grad.cpu(), "w": eng.bucket.flat.cpu(),
                 "rm": eng.backbone_encoder.stem[1].running_mean.cpu()}
    dist.destroy_process_group()


def test_two_rank_step_matches_gradient_average(dev):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert torch.equal(r0["grad"], r1["grad"]), "all-reduced gradients must be identical on every rank"
    assert torch.equal(r0["w"], r1["w"]), "weights must stay in lock-step"
    assert torch.equal(r0["rm"], r1["rm"]), "SyncBatchNorm running stats must agree"
    assert r0["loss"] != r1["loss"]  # each rank saw its own shard
    # single-process check of the SyncBatchNorm statistics: encoder running_mean after one step on the FULL batch
    from adaptersis_amd.utils import weights as W
    from tests.test_gpu_step import build_engine
    eng, _ = build_engine("vit_tiny_test", "kernel", dev, (128, 32, 16, 16, 8), lr=0.05)
    img, tgt = W.synthetic_batch(4, 224, seed=7)
    eng.train_step(img.to(dev), tgt.to(dev))
    full_rm = eng.backbone_encoder.stem[1].running_mean.cpu()
    assert torch.allclose(full_rm, r0["rm"], rtol=1e-5, atol=1e-7), "2-rank SyncBN == 1-rank BN over the whole batch"


def _worker_enc(rank, world, port, out):
    """FeatureEncoder forward + backward (SyncBatchNorm in both directions) on this rank's half of the batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out[rank] = _encoder_grads(slice(rank * 2, rank * 2 + 2), 1.0 / world, reduce=True)
    dist.destroy_process_group()


def _encoder_grads(sl, inv, reduce):
    from adaptersis_amd.backbones.encoders import FeatureEncoder
    from adaptersis_amd.utils import weights as W
    dev = torch.device("cuda:0")
    D = 128
    enc = FeatureEncoder(embed_dim=D)
    enc.load_state_dict(W.make_encoder_state_dict(D))
    enc = enc.to(dev)
    img, _ = W.synthetic_batch(4, 224, seed=11)
    c, shapes, saved = enc.forward_tokens_train(img[sl].to(dev))
    dc = W.tensor("dist.dc", (4, c.shape[1], D), 1.0)[sl].to(dev).contiguous()
    grads = {n: torch.zeros_like(p) for n, p in enc.named_parameters()}
    enc.backward_tokens(saved, dc, inv, grads)
    if reduce:
        flat = torch.cat([g.reshape(-1) for g in grads.values()])
        dist.all_reduce(flat)            # what the engine's encoder bucket reducer does (SUM of 1/world-scaled gradients)
        o = 0
        for g in grads.values():
            g.copy_(flat[o:o + g.numel()].view(g.shape)); o += g.numel()
    torch.cuda.synchronize()
    return {n: g.cpu() for n, g in grads.items()}


def test_two_rank_encoder_backward_matches_full_batch(dev):
    """SyncBatchNorm gamma / beta gradients must be LOCAL sums (averaged by the bucket all-reduce), not the
    all-reduced sums again — 2 ranks x half batch == 1 process x full batch for every encoder parameter."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_enc, args=(world, _free_port(), out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    full = _encoder_grads(slice(0, 4), 1.0 / world, reduce=False)
    for n, g in full.items():
        assert torch.equal(r0[n], r1[n]), n
        if float(g.norm()) == 0:
            continue
        err = float((r0[n].double() - g.double()).norm() / g.double().norm())
        assert err < 2e-3, (n, err)
