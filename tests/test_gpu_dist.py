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
