"""GPU: the RCCL code path itself on the one-GPU box.  The driver's scaling runs are the only place N > 1 ranks ever meet
real RCCL, so this rehearses every collective call site of the step in a ONE-rank ``nccl`` process group with
``parallel.FORCE_COLLECTIVES`` (a 1-rank all-reduce is the identity): backend initialisation on the device, the per-stage
gradient all-reduces on the reducer's side stream, the fp64 SyncBatchNorm statistic exchanges issued from the encoder's side
stream, the SyncBN backward exchange, the chunked backbone bucket of config 4 — and the results must equal the same step
without any collective, bit for bit.  Runs in a child process (a process group is process-global state)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import os, sys, torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["ASIS_ROOT"])
    from adaptersis_amd import parallel
    from adaptersis_amd.utils import weights as W
    from tests.test_gpu_e2e import build_e2e_engine
    from tests.test_gpu_step import build_engine
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    img, tgt = W.synthetic_batch(2, 224)
    img, tgt = img.to(dev), tgt.to(dev)

    def run(kind):
        if kind == "frozen":
            eng, _ = build_engine("vit_tiny_test", "kernel", dev, (128, 32, 16, 16, 8), lr=0.05)
        else:
            eng, _ = build_e2e_engine("vit_tiny_test", dev, (128, 32, 16, 16, 8), blocks_per_bucket=2)
        losses = [float(eng.train_step(img, tgt)) for _ in range(2)]
        torch.cuda.synchronize()
        out = {"loss": losses, "w": eng.bucket.flat.clone(), "rm": eng.backbone_encoder.stem[1].running_mean.clone()}
        if kind != "frozen":
            out.update(vit=eng.vit_bucket.grad.clone(), ada=eng.adapter_bucket.flat.clone(), enc=eng.encoder_bucket.flat.clone())
        return out

    base = {k: run(k) for k in ("frozen", "e2e")}                       # no process group: no collective is issued
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    parallel.FORCE_COLLECTIVES = True
    t = torch.ones(4, device=dev, dtype=torch.float64)
    dist.all_reduce(t)                                                   # fp64 on RCCL (the SyncBN statistics dtype)
    assert t.tolist() == [1.0] * 4
    forced = {k: run(k) for k in ("frozen", "e2e")}
    # bf16 transport of a gradient range (StageReducer(compress="bf16")): pack kernel -> RCCL bf16 all-reduce -> unpack kernel on
    # the reducer's side stream; with one rank the result is the range rounded once to bf16 (RNE), tiny magnitudes kept
    g = (torch.randn(8192 + 4, device=dev) * torch.logspace(-9, 0, 8192 + 4, device=dev)).contiguous()
    want = g.to(torch.bfloat16).float()
    red = parallel.StageReducer(g, [(0, 4096), (4096, 8192 + 4)], compress="bf16")
    red.begin(); red.stage_done(); red.stage_done(); red.finish()
    torch.cuda.synchronize()
    assert torch.equal(g, want), float((g - want).abs().max())
    dist.barrier()
    dist.destroy_process_group()
    for k in base:
        assert base[k]["loss"] == forced[k]["loss"], (k, base[k]["loss"], forced[k]["loss"])
        for name in base[k]:
            if name == "vit":
                # the backbone bucket travels as bfloat16 by default (SegEngine grad_compress, round 5): with one rank the
                # exchange returns every range rounded once to bf16
                assert torch.equal(base[k][name].to(torch.bfloat16).float(), forced[k][name]), (k, name)
            elif name != "loss":
                assert torch.equal(base[k][name], forced[k][name]), (k, name)
    print("RCCL_REHEARSAL_OK", base["frozen"]["loss"], base["e2e"]["loss"])
''')


def test_every_collective_call_site_runs_on_rccl_single_rank(dev):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ASIS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_REHEARSAL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


CHILD2 = textwrap.dedent('''
    import os, sys, torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["ASIS_ROOT"])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("ASIS_2RANK_BACKEND", "nccl")             # "gloo": both ranks share device 0 (one-GPU box)
    idx = rank if backend == "nccl" else 0
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptersis_amd.utils import weights as W
    from tests.test_gpu_e2e import build_e2e_engine
    from tests.test_gpu_step import build_engine
    img, tgt = W.synthetic_batch(2 * world, 224, seed=7)
    sl = slice(2 * rank, 2 * rank + 2)                                   # DistributedSampler-style shard
    res = {}
    for kind in ("frozen", "e2e"):
        if kind == "frozen":
            eng, _ = build_engine("vit_tiny_test", "kernel", dev, (128, 32, 16, 16, 8), lr=0.05)
        else:
            eng, _ = build_e2e_engine("vit_tiny_test", dev, (128, 32, 16, 16, 8), blocks_per_bucket=2)
        for _ in range(2):
            eng.train_step(img[sl].to(dev), tgt[sl].to(dev))
        torch.cuda.synchronize()
        flats = [eng.bucket.flat, eng.backbone_encoder.stem[1].running_mean]
        if kind != "frozen":
            flats += [eng.vit_bucket.grad, eng.adapter_bucket.flat, eng.encoder_bucket.flat]
        for i, t in enumerate(flats):                                    # every rank must hold the same values
            lo, hi = t.clone(), t.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), (kind, i)
            assert bool(torch.isfinite(t).all()), (kind, i)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_2RANK_OK")
''')


def _run_two_ranks(extra_env):
    env = dict(os.environ, ASIS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    path = os.path.join(ROOT, "gpurun_out", "_rccl2_child.py")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(CHILD2)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
                           "--nproc-per-node=2", path], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)


@pytest.mark.parametrize("compress", ["none", "bf16"])
def test_two_rank_body_over_gloo_on_one_card(compress):
    """The body of the two-real-rank RCCL test below — two steps of the frozen-backbone engine and of the unfrozen one, then
    bit-equality of weights, running statistics and reduced gradients across the ranks — with both ranks on the one visible
    device over gloo, so that its logic executes on the one-GPU box too; ``compress``: the 16-bit gradient
    transport (ASIS_GRAD_COMPRESS) on every bucket."""
    r = _run_two_ranks({"ASIS_2RANK_BACKEND": "gloo", "ASIS_GRAD_COMPRESS": compress})
    assert r.returncode == 0 and "RCCL_2RANK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_two_real_ranks_over_rccl_when_two_devices_are_visible():
    """N > 1 on real RCCL: two ranks, one device each, two steps of the frozen-backbone engine and of the
    unfrozen one (SyncBN exchanges, per-stage decoder all-reduces, chunked backbone bucket with blocks_per_bucket <
    n_last_blocks, adapter / encoder buckets) — weights, running statistics and reduced gradients must be identical on
    both ranks.  Needs two visible devices: skipped on the one-GPU box (counting devices does not initialise HIP here)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (the driver's multi-GPU node)")
    r = _run_two_ranks({"ASIS_2RANK_BACKEND": "nccl"})
    assert r.returncode == 0 and "RCCL_2RANK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
