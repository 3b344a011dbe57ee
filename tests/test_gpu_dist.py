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


def _worker_e2e(rank, world, port, out):
    """Config-4 engine (backbone unfrozen) with a backbone chunk SMALLER than n_last_blocks: the first chunk holds norm.*,
    whose gradients only exist after the four adapter stages."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd.utils import weights as W
    from tests.test_gpu_e2e import build_e2e_engine
    dev = torch.device("cuda:0")
    eng, _ = build_e2e_engine("vit_tiny_test", dev, (128, 32, 16, 16, 8), blocks_per_bucket=2)
    img, tgt = W.synthetic_batch(4, 224, seed=3)
    sl = slice(rank * 2, rank * 2 + 2)
    eng.train_step(img[sl].to(dev), tgt[sl].to(dev))
    torch.cuda.synchronize()
    out[rank] = {"vit": eng.vit_bucket.grad.cpu(), "norm_w": eng.vit_bucket.views["norm.weight"].cpu(),
                 "fire_at": list(eng._fire_at)}
    dist.destroy_process_group()


def test_two_rank_unfrozen_step_reduces_final_norm_gradients(dev):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_e2e, args=(world, _free_port(), out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert max(r0["fire_at"]) <= 0, r0["fire_at"]          # depth 4, n_last_blocks 4: nothing fires before block 0 is done
    assert float(r0["norm_w"].abs().sum()) > 0
    assert torch.equal(r0["norm_w"], r1["norm_w"]), "final-norm gradients were not all-reduced"
    assert torch.equal(r0["vit"], r1["vit"])


def _worker_ragged(rank, world, port, out):
    """SyncBatchNorm with DIFFERENT per-rank batches (1 and 2 images): parallel.set_batch_ratio announces it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd import parallel
    from adaptersis_amd.backbones.encoders import FeatureEncoder
    from adaptersis_amd.utils import weights as W
    dev = torch.device("cuda:0")
    enc = FeatureEncoder(embed_dim=128)
    enc.load_state_dict(W.make_encoder_state_dict(128))
    enc = enc.to(dev)
    img, _ = W.synthetic_batch(3, 224, seed=5)
    sl = slice(0, 1) if rank == 0 else slice(1, 3)
    ratio = parallel.set_batch_ratio(sl.stop - sl.start)
    _, c, shapes = enc.forward_tokens(img[sl].to(dev), need_c1=False)
    torch.cuda.synchronize()
    parallel.set_batch_ratio(None)
    out[rank] = {"c": c.cpu(), "rm": enc.stem[1].running_mean.cpu(), "rv": enc.stem[1].running_var.cpu(), "ratio": ratio}
    dist.destroy_process_group()


def test_two_rank_syncbn_with_ragged_batches_matches_full_batch(dev):
    """`nn.SyncBatchNorm` weights every rank's statistics by its element count (`backbones/encoders.py:12-40`): ranks holding
    1 and 2 images must reproduce one process normalising all 3."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_ragged, args=(world, _free_port(), out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert r0["ratio"] == 3.0 and r1["ratio"] == 1.5
    from adaptersis_amd.backbones.encoders import FeatureEncoder
    from adaptersis_amd.utils import weights as W
    enc = FeatureEncoder(embed_dim=128)
    enc.load_state_dict(W.make_encoder_state_dict(128))
    enc = enc.to(dev)
    img, _ = W.synthetic_batch(3, 224, seed=5)
    _, c, _ = enc.forward_tokens(img.to(dev), need_c1=False)
    c = c.cpu()
    assert torch.allclose(r0["rm"], enc.stem[1].running_mean.cpu(), rtol=1e-5, atol=1e-7)
    assert torch.allclose(r0["rv"], enc.stem[1].running_var.cpu(), rtol=1e-5, atol=1e-7)
    assert torch.equal(r0["rm"], r1["rm"])
    for got, ref in ((r0["c"], c[:1]), (r1["c"], c[1:])):
        assert float((got.double() - ref.double()).norm() / ref.double().norm()) < 1e-5


def test_bench_two_ranks_rehearsal_over_gloo(dev):
    """`python bench.py --gpus 2` end to end on the one-GPU box: both ranks on device 0, collectives over gloo
    (ASIS_BENCH_BACKEND=gloo) — the launcher, the barriers, the MAX-over-ranks timing, the per-stage gradient all-reduce, the
    secondary passes (each builds a second engine over the same modules on every rank) and the single JSON line of the N > 1
    path the driver runs on its 8-GPU node.  Small geometry (vit_tiny_test, 224^2, 2 images per rank): a rehearsal, not a
    measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(ASIS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--arch",
                        "vit_tiny_test", "--size", "224", "--batch", "2", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["config"]["global_batch"] == 4 and j["scaling"] == "weak"
    assert j["value"] > 0 and j["config"]["skipped_optimizer_steps"] == 0
    assert set(j["secondary"]) >= {"all_reference_calls", "bf16", "bf16_at_1e-3"} and j["host_enqueue_ms_per_step"] > 0
