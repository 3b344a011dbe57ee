#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the ViT-L/14 588x588 adapter fine-tune step on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one `train.py:268-436` iteration on a batch of 12 synthetic 588x588 images per
GPU, already resident in HBM: CNN encoder, ViT-L pass A (24 blocks, cls+pos) + pass B (21 blocks), 4 x [block,
CAViT, CACNN], decoder forward, resize+softmax+Dice, decoder backward, gradient all-reduce (N > 1), SGD.
Nothing is cached between steps.  THREE calls of the reference step are elided in the timed region, none of which can change
a result (``config.dead_cacnn_elided``, ``config.c1_elided``, ``config.patch_embed_shared`` in the JSON line report them):
  * the last stage's CACNN (`train.py:372-386`), whose output no later line reads (the decoder takes c4 from the encoder, `:395`);
  * the encoder's `fc1` 1x1 conv at 147^2 -> c1 (`backbones/encoders.py:44,55,68`), which `train.py:279` drops;
  * the second evaluation of the patch embedding (pass A computes it inside `get_intermediate_layers`, pass B again on the same
    input and frozen weights: `train.py:287,300`) — computed once and shared.
``--all-reference-calls`` runs all three; without it the line still carries the figure of such a run as
``secondary.all_reference_calls`` (a short extra pass AFTER the timed region: 5 warm-up + 10 steps), next to ``secondary.bf16``
(``--operand bf16`` on the same modules) and ``host_enqueue_ms_per_step`` (how long the host takes to enqueue one step with the
GPU idle at the start: Python + ctypes + HIP launches; far below ms_per_step = the GPU is the limiter).
Weights are random-init of the named architecture (no network: `adaptersis_amd.utils.weights`), data is synthetic of the
named shape.

Output: ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16/f16 (no sparsity)
# what a bare f16 MFMA loop (operands in registers, every SIMD's matrix pipe 100 % busy) sustains on RANDOM data on this pool's
# MI355X: the chip clocks it at 1.65 GHz (2.38 GHz and 2.49 PFLOP/s on all-zero data).  scripts/mfma_peak.hip,
# profiles/r05_mfma_sustained_peak.txt.  Reported beside the nominal peak, never instead of it.
MFMA_F16_SUSTAINED_RANDOM_TFLOPS = 1690.0


def build_engine(arch: str, dev, lr: float, mode: str = "reference_exact", train_encoder: bool = False):
    from adaptersis_amd.backbones.adapter_blocks import CACNN, CAViT
    from adaptersis_amd.backbones.decoders import FeatureDecoder
    from adaptersis_amd.backbones.encoders import FeatureEncoder
    from adaptersis_amd.backbones.engines import SegEngine
    from adaptersis_amd.dinov2.models import vision_transformer as vits
    from adaptersis_amd.utils import weights as W

    D, depth, heads, ffn = W.VIT_CONFIGS[arch]
    model = vits.__dict__[arch](patch_size=14, img_size=518, init_values=1e-5, ffn_layer=ffn, block_chunks=0)
    model.load_state_dict(W.make_vit_state_dict(arch, layerscale="kernel"))
    enc = FeatureEncoder(embed_dim=D)
    enc.load_state_dict(W.make_encoder_state_dict(D))
    cv = CAViT(dim=D, n_levels=3, num_heads=8, init_values=0.0, n_points=4)
    cv.load_state_dict(W.make_cavit_state_dict(D, mode="kernel"))
    cn = CACNN(dim=D, n_levels=1, num_heads=8, n_points=4, with_cffn=True, cffn_ratio=0.25)
    cn.load_state_dict(W.make_cacnn_state_dict(D, mode="kernel"))
    dec = FeatureDecoder(embed_dim=D, num_classes=2, features=[D, 512, 256, 128, 64])
    dec.load_state_dict(W.make_feature_decoder_state_dict(D, 2, features=(D, 512, 256, 128, 64)))
    return SegEngine(model.to(dev).eval(), enc.to(dev), cv.to(dev), cn.to(dev), dec.to(dev), lr=lr, mode=mode,
                     train_encoder=train_encoder)


def build_engine_cfg(cfg: int, arch: str, dev, lr: float, setr_only: bool = False):
    """BASELINE configs other than the headline one: 2 = ViT-B + adapters + UNet head (CE + DC), 4 = unfrozen
    end-to-end ViT + DecoderSETR (full backward, backbone gradients all-reduced, decoder-only optimiser)."""
    from adaptersis_amd.backbones.adapter_blocks import CACNN, CAViT
    from adaptersis_amd.backbones.decoders import DecoderSETR
    from adaptersis_amd.backbones.encoders import FeatureEncoder
    from adaptersis_amd.backbones.engines import EndToEndEngine, SegEngine
    from adaptersis_amd.backbones.unet_parts import UNet
    from adaptersis_amd.dinov2.models import vision_transformer as vits
    from adaptersis_amd.utils import weights as W

    D, depth, heads, ffn = W.VIT_CONFIGS[arch]
    model = vits.__dict__[arch](patch_size=14, img_size=518, init_values=1e-5, ffn_layer=ffn, block_chunks=0)
    if arch == "vit_giant2":
        # 1.1 B parameters: the name-keyed CPU Philox generator of the parity tests takes minutes; the bench only needs
        # well-scaled random weights, drawn on the device (same distributions: U(-a, a) with a = sqrt(3 / fan_in) etc.)
        model = model.to(dev)
        g = torch.Generator(device=dev).manual_seed(0)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() >= 2 and "pos_embed" not in n and "cls_token" not in n and "mask_token" not in n:
                    fan_in = p[0].numel()
                    p.copy_((torch.rand(p.shape, device=dev, generator=g) * 2 - 1) * (3.0 / fan_in) ** 0.5)
                elif n.endswith("gamma"):
                    p.copy_(0.05 + 0.45 * torch.rand(p.shape, device=dev, generator=g))
                elif "norm" in n and n.endswith("weight"):
                    p.copy_(1.0 + 0.2 * (torch.rand(p.shape, device=dev, generator=g) - 0.5))
                else:
                    p.copy_(0.2 * (torch.rand(p.shape, device=dev, generator=g) - 0.5))
    else:
        model.load_state_dict(W.make_vit_state_dict(arch, layerscale="kernel"))
    if cfg == 4 and setr_only:
        # the reference's own end-to-end script (`eval/eval_dinov2_setr_cross_ete.py`): ViT -> DecoderSETR, no adapters
        dec = DecoderSETR(D, 2)
        dec.load_state_dict(W.make_setr_state_dict(D, 2))
        return EndToEndEngine(model.to(dev), dec.to(dev), lr=lr)
    enc = FeatureEncoder(embed_dim=D)
    enc.load_state_dict(W.make_encoder_state_dict(D))
    cv = CAViT(dim=D, n_levels=3, num_heads=8, init_values=0.0, n_points=4)
    cv.load_state_dict(W.make_cavit_state_dict(D, mode="kernel"))
    cn = CACNN(dim=D, n_levels=1, num_heads=8, n_points=4, with_cffn=True, cffn_ratio=0.25)
    cn.load_state_dict(W.make_cacnn_state_dict(D, mode="kernel"))
    if cfg == 4:
        # BASELINE config 4: the train.py adapter flow with the backbone unfrozen — both ViT passes with
        # weight gradients, adapters + encoder + decoder trained, 304 M + 31 M gradients all-reduced in buckets
        from adaptersis_amd.backbones.decoders import FeatureDecoder
        dec = FeatureDecoder(embed_dim=D, num_classes=2, features=[D, 512, 256, 128, 64])
        dec.load_state_dict(W.make_feature_decoder_state_dict(D, 2, features=(D, 512, 256, 128, 64)))
        return SegEngine(model.to(dev), enc.to(dev), cv.to(dev), cn.to(dev), dec.to(dev), lr=lr, mode="train_adapters",
                         train_encoder=True, train_backbone=True)
    if cfg == 5:
        from adaptersis_amd.backbones.decoders import DecoderMLA
        dec = DecoderMLA(img_size=588, mla_channels=D, num_classes=11)
        dec.load_state_dict(W.make_decoder_mla_state_dict(D, 128, 11))
        return SegEngine(model.to(dev).eval(), enc.to(dev), cv.to(dev), cn.to(dev), dec.to(dev), lr=lr, momentum=0.9,
                         weight_decay=0.0, num_classes=11, loss="iou")
    dec = UNet(D, 2)
    dec.load_state_dict(W.make_unet_state_dict(D, 2))
    return SegEngine(model.to(dev).eval(), enc.to(dev), cv.to(dev), cn.to(dev), dec.to(dev), lr=lr, loss="ce_dc")


def synthetic(batch: int, size: int, rank: int, dev, num_classes: int = 2):
    """images U[0,1) (no mean/std normalisation), binary masks ~30 % foreground, one
    all-background image per batch (Dice epsilon path); multi-class: piecewise-constant 14x14 blocks; seed 0 + rank."""
    g = torch.Generator(device="cpu").manual_seed(rank)
    img = torch.rand(batch, 3, size, size, generator=g)
    if num_classes == 2:
        tgt = (torch.rand(batch, size, size, generator=g) > 0.7).long()
        tgt[-1].zero_()
    else:
        gsz = (size + 13) // 14
        blocks = torch.randint(0, num_classes, (batch, gsz, gsz), generator=g)
        tgt = blocks.repeat_interleave(14, 1).repeat_interleave(14, 2)[:, :size, :size].contiguous()
    return img.to(dev), tgt.to(dev)


def host_cores():
    """-> (cores this process may really use, how that number was derived).  The affinity mask, cut to the cgroup CPU quota
    when one is visible; on a box of the GPU pool (marker: ``GRAFT_REPO_ROOT``) that shows its whole host in the mask and no
    quota, the pool's documented share of 16 cores per GPU (256 oracle threads on that share ran 12x slower than 16);
    ``ASIS_CPU_THREADS`` overrides.  An unconstrained many-core host keeps its affinity count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "affinity mask"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = int(f.read()), int(f2.read())
                if q > 0:
                    quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, how = quota, "cgroup CPU quota"
    elif quota is None and n > 64 and os.environ.get("GRAFT_REPO_ROOT"):
        n, how = 16, "GPU-pool share (16 cores per GPU; no cgroup quota visible, affinity mask = whole host)"
    if os.environ.get("ASIS_CPU_THREADS"):
        n, how = int(os.environ["ASIS_CPU_THREADS"]), "ASIS_CPU_THREADS"
    return max(1, n), how


def cpu_baseline(arch: str, size: int, batch: int = 1):
    """Reference CPU path (the fp32 eager oracle restatement, parity-pinned to the imported reference) on this box's
    host cores: WHOLE `train.py:268-436` steps at batch ``batch`` timed end to end (encoder, ViT pass A + pass B,
    4 adapter stages, decoder forward, loss, decoder backward, SGD) — one untimed warm-up step, then two timed ones
    (``value`` = batch / their mean), plus one timed step at the other batch size of {1, 2} in ``sample``."""
    from adaptersis_amd.utils import weights as W
    from oracle import ref_torch as O  # cpu_baseline leg only

    cores, cores_how = host_cores()
    torch.set_num_threads(cores)
    D, depth, heads, _ = W.VIT_CONFIGS[arch]
    vsd = W.make_vit_state_dict(arch, layerscale="kernel")
    esd, csd, nsd = W.make_encoder_state_dict(D), W.make_cavit_state_dict(D), W.make_cacnn_state_dict(D)
    dsd = W.make_feature_decoder_state_dict(D, 2, features=(D, 512, 256, 128, 64))
    N = (size // 14) ** 2

    def t(fn):
        t0 = time.perf_counter()
        out = fn()
        return time.perf_counter() - t0, out

    with torch.no_grad():  # warm the thread pool / allocator on one block (not part of the timed step)
        xa = W.tensor("cb.xa", (1, N + 1, D), 1.0)
        O.block(xa, vsd, "blocks.0", heads)
        t_a, _ = t(lambda: O.block(xa, vsd, "blocks.0", heads))
    params = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in dsd.items()}
    mom = {}

    def step(img, tgt):
        t0 = time.perf_counter()
        with torch.no_grad():
            cat = O.adapter_forward(img, vsd, esd, csd, nsd, heads)
        t_fwd = time.perf_counter()
        for p_ in params.values():
            p_.grad = None
        loss = O.train_step_loss(cat, tgt, params, 2)
        loss.backward()
        with torch.no_grad():   # torch.optim.SGD(momentum 0.99, wd 3e-5), train.py:178-191
            tr = {k: p_ for k, p_ in params.items() if p_.grad is not None}
            O.sgd_momentum_step(tr, {k: p_.grad for k, p_ in tr.items()}, mom, 0.01)
        return time.perf_counter() - t0, t_fwd - t0, float(loss)

    img, tgt = W.synthetic_batch(batch, size)
    step(img, tgt)                                       # warm-up (allocator, thread pool, momentum buffers): not timed
    runs = [step(img, tgt) for _ in range(2)]
    t_step = sum(r[0] for r in runs) / len(runs)
    other = 2 if batch == 1 else 1
    t_other = step(*W.synthetic_batch(other, size))[0]
    return {
        "value": round(batch / t_step, 5), "unit": "img/s", "cores": cores, "kind": "port",
        "cores_from": cores_how, f"value_batch{other}": round(other / t_other, 5),
        "sample": (f"whole train.py steps timed end to end, {arch} {size}x{size}: 1 warm-up + 2 timed steps at batch {batch} "
                   f"({', '.join('%.1fs' % r[0] for r in runs)}; features {runs[-1][1]:.1f}s, decoder fwd+loss+bwd+SGD "
                   f"{runs[-1][0] - runs[-1][1]:.1f}s), then one timed step at batch {other} ({t_other:.1f}s = "
                   f"{other / t_other:.4f} img/s); one block at N={N + 1} alone: {t_a:.2f}s; loss {runs[-1][2]:.4f}; "
                   f"fp32 eager torch {torch.__version__}, {cores} threads ({cores_how})"),
    }


def gemm_sources_sha256():
    """hash of adaptersis_amd/csrc/gemm* (the kernels the roofline figures are about): scripts/pmc_traffic.py stores it with a
    traffic measurement, and a measurement taken on other kernel sources is not reported"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "adaptersis_amd", "csrc", "gemm*"))):
        if f.endswith((".h", ".hip")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


def pmc_traffic(kernel: str, variants):
    """HBM-side bytes per launch of the dominant kernel — the launch-weighted mean over its listed template forms — from the
    committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of this very command, gfx950 read-side
    doubling applied: scripts/pmc_traffic.py).  PMC counters cannot be collected from inside the timed run, so this is the
    figure of the last profiled run — reported ONLY when that run's GEMM kernel sources are the ones of this tree (sha256 of
    adaptersis_amd/csrc/gemm*); otherwise ``traffic`` is null and the reason is in ``traffic_source``."""
    here = os.path.dirname(os.path.abspath(__file__))
    cands = sorted((f for f in os.listdir(os.path.join(here, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True) \
        if os.path.isdir(os.path.join(here, "profiles")) else []
    if not cands:
        return None, {"reason": "no profiles/*_pmc_traffic.json"}
    fn = cands[0]
    with open(os.path.join(here, "profiles", fn)) as f:
        doc = json.load(f)
    src = {"file": "profiles/" + fn, "commit": doc.get("commit"), "command": doc.get("command")}
    if doc.get("gemm_sources_sha256") != gemm_sources_sha256():
        src["reason"] = ("stale: the PMC passes were taken on other GEMM kernel sources (gemm_sources_sha256 differs); re-run "
                         "scripts/collect_profiles.sh on this tree")
        return None, src
    rows = doc["kernels"]
    tot = n = 0
    for name, v in rows.items():
        if any(k in name for k in kernel) and any(var in name for var in variants):
            tot += v["hbm_bytes_total"]
            n += v["launches"]
    return (round(tot / n) if n else None), src


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: run N ranks as children of this (GPU-free) process through
    ``python -m torch.distributed.run`` on 127.0.0.1, pass their stdout / stderr through (rank 0 prints the JSON line)
    and return the launcher's exit code (non-zero when any rank failed)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    # --standalone: torchrun's own c10d rendezvous on a port IT picks and holds (no bind-then-close race), on 127.0.0.1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=12, help="images per GPU (README.md:45-61 of the reference)")
    ap.add_argument("--arch", default=None)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 3 = the headline metric (default); 2 = ViT-B + UNet head; 4 = unfrozen end-to-end; 5 = ViT-g + MLA head, 11 classes")
    ap.add_argument("--size", type=int, default=588)
    ap.add_argument("--operand", default=None, choices=[None, "f16", "bf16"])
    ap.add_argument("--train-adapters", action="store_true",
                    help="config 3 with the adapter backward (CAViT + CACNN gradients, all-reduced and optimised with the decoder)")
    ap.add_argument("--train-encoder", action="store_true", help="with --train-adapters: also the CNN encoder (the full optimiser list of train.py:178-186)")
    ap.add_argument("--e2e-setr", action="store_true", help="config 4 as the reference's own end-to-end script: ViT -> DecoderSETR without adapters")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="profiling aid: every launch on the compute stream (encoder / V^T / weight-gradient / dual-trunk side streams "
                         "off), so that a rocprofv3 kernel trace of this command shows un-overlapped launch durations")
    ap.add_argument("--run-dead-cacnn", action="store_true",
                    help="also run the last stage's CACNN, whose output nothing reads (train.py:372-386; elided by default: identical results)")
    ap.add_argument("--all-reference-calls", action="store_true",
                    help="run every call of the reference step in the timed region: the dead CACNN, the encoder's c1 conv and the "
                         "second patch embedding (all three elided by default: identical results)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short extra passes after the timed region (secondary.all_reference_calls, secondary.bf16, host enqueue time)")
    ap.add_argument("--cpu-baseline-batch", type=int, default=1, help="batch of the timed CPU (oracle) step: 1 or 2")
    a = ap.parse_args()
    # dmabuf IPC only on this pool: must be in the environment before the HIP runtime starts in THIS process too (a rank started
    # by the driver's torchrun inherits it from the driver; this is the default for any other launcher)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, BEFORE this process makes any GPU call (a process that
        # has initialised HIP must never exec / fork GPU children); the parent only relays output and the exit code
        return launch_ranks(a.gpus)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")

    if os.environ.get("ASIS_BENCH_RANKCHECK"):
        # launcher rehearsal without GPUs (tests/test_bench_launcher.py): rendezvous over gloo, count the ranks, stop
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rankcheck": True, "n_gpus": a.gpus, "n_ranks_seen": dist.get_world_size(), "sum": float(t)}), flush=True)
        dist.destroy_process_group()
        return 0
    # the in-tree library is built (normally a no-op) BEFORE the first GPU call of this process: no compiler children
    # under an initialised HIP runtime / a preloaded profiler.  File-locked: one rank builds, the others wait.
    from adaptersis_amd.build import build_library
    build_library()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP device); there is no CPU fallback")
    # ASIS_BENCH_BACKEND=gloo (rehearsal on a one-GPU box, tests/test_gpu_dist.py): every rank on device 0, collectives over gloo —
    # the N > 1 code path of this file (barriers, MAX over ranks, secondary passes, one JSON line) without a second device
    backend = os.environ.get("ASIS_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" == RCCL on ROCm

    # torchrun exports OMP_NUM_THREADS=1: give every rank its share of the host cores for the (CPU, Philox) weight generation
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, cores // max(1, world))))

    from adaptersis_amd import config, ops
    if a.single_stream:
        config.encoder_stream = config.vt_stream = config.wgrad_stream = config.dual_stream = False
    if a.run_dead_cacnn:
        config.elide_dead_cacnn = False
    if a.all_reference_calls:
        config.elide_dead_cacnn = config.elide_c1 = config.share_patch_embed = False
    if a.operand:
        config.set_operand_dtype(torch.float16 if a.operand == "f16" else torch.bfloat16)
        config.loss_scale = 65536.0 if a.operand == "f16" else 1.0

    a.arch = a.arch or {2: "vit_base", 5: "vit_giant2"}.get(a.config, "vit_large")
    if a.config == 3:
        eng = build_engine(a.arch, dev, lr=0.01, mode="train_adapters" if a.train_adapters else "reference_exact",
                           train_encoder=a.train_encoder)
    else:
        eng = build_engine_cfg(a.config, a.arch, dev, lr=0.01, setr_only=a.e2e_setr)
    img, tgt = synthetic(a.batch, a.size, rank, dev, 11 if a.config == 5 else 2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        eng.train_step(img, tgt)
    barrier()
    # ---- the timed region: exactly `steps` steps, no instrumentation (no per-launch events) ----------------
    ops.PROFILE = None
    t0 = time.perf_counter()
    loss = None
    for _ in range(a.steps):
        loss = eng.train_step(img, tgt)
    barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el)
    loss_v = float(loss)
    # ---- second pass of the same steps with a HIP event pair around every GEMM launch: the roofline figures ----
    prof = None if a.no_kernel_timing else []
    if prof is not None:
        ops.PROFILE = prof
        enc_stream, config.encoder_stream = config.encoder_stream, False   # one stream: a launch's event pair times that launch alone
        vt_stream, config.vt_stream = config.vt_stream, False
        wg_stream, config.wgrad_stream = config.wgrad_stream, False    # weight gradients too: nothing runs beside a timed launch
        du_stream, config.dual_stream = config.dual_stream, False
        for _ in range(a.steps):
            eng.train_step(img, tgt)
        barrier()
        config.encoder_stream = enc_stream
        config.vt_stream = vt_stream
        config.wgrad_stream = wg_stream
        config.dual_stream = du_stream
        ops.PROFILE = None

    # ---- secondary figures (never inside the timed region): short passes of 5 warm-up + 10 steps each ----------------
    secondary, host_enqueue_ms = None, None
    if not a.no_secondary:
        def timed(e, warm=5, steps=10):
            for _ in range(warm):
                e.train_step(img, tgt)
            barrier()
            t_ = time.perf_counter()
            for _ in range(steps):
                e.train_step(img, tgt)
            barrier()
            dt_ = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(dt_, op=dist.ReduceOp.MAX)
            return float(dt_) / steps

        # host enqueue time of one step: GPU idle at the start, five steps enqueued back to back, clock stopped when the last
        # launch call returns (before any synchronisation)
        barrier()
        t_ = time.perf_counter()
        for _ in range(5):
            eng.train_step(img, tgt)
        host_enqueue_ms = (time.perf_counter() - t_) / 5 * 1e3
        barrier()
        secondary = {}
        if a.config == 3 and not a.all_reference_calls:
            saved = (config.elide_dead_cacnn, config.elide_c1, config.share_patch_embed)
            config.elide_dead_cacnn = config.elide_c1 = config.share_patch_embed = False
            t_all = timed(eng)
            config.elide_dead_cacnn, config.elide_c1, config.share_patch_embed = saved
            secondary["all_reference_calls"] = {
                "value": round(a.batch * world / t_all, 3), "unit": "img/s", "ms_per_step": round(t_all * 1e3, 3), "steps": 10, "warmup": 5,
                "what": "the same step with the dead CACNN call, the encoder's c1 conv and the second patch embedding executed"}
        if a.config == 3 and a.operand is None and not a.train_adapters:
            # bf16 operands on the same modules (derived 16-bit operand caches are keyed by dtype); a new engine = new flat buckets
            from adaptersis_amd.backbones.engines import SegEngine
            old_dt, old_ls = config.operand_dtype, config.loss_scale
            config.set_operand_dtype(torch.bfloat16)
            config.loss_scale = 1.0
            try:
                eng_bf = SegEngine(eng.model, eng.backbone_encoder, eng.cross_vit, eng.cross_cnn, eng.seg_decoder, lr=0.01,
                                   mode="reference_exact")
                t_bf = timed(eng_bf)
                secondary["bf16"] = {"value": round(a.batch * world / t_bf, 3), "unit": "img/s", "ms_per_step": round(t_bf * 1e3, 3),
                                     "steps": 10, "warmup": 5,
                                     "what": "--operand bf16 (no LayerNorm fold, no MX pass, no halo-tile convolution: f16-only paths); "
                                             "parity of this mode: tests/test_gpu_numerics.py"}
                del eng_bf
                # ... and the bf16 configuration that holds north_star's 1e-3 on the stress golden (tests/test_gpu_numerics.py):
                # precise_level 2 (every linear layer of the blocks on hi + lo bf16 operands) + the adapters' MSDA layers split
                old_pl = config.precise_level_policy
                config.precise_level_policy = 2
                try:
                    eng_bp = SegEngine(eng.model, eng.backbone_encoder, eng.cross_vit, eng.cross_cnn, eng.seg_decoder, lr=0.01,
                                       mode="reference_exact")
                    t_bp = timed(eng_bp, warm=3, steps=6)
                    secondary["bf16_at_1e-3"] = {"value": round(a.batch * world / t_bp, 3), "unit": "img/s", "ms_per_step": round(t_bp * 1e3, 3),
                                                 "steps": 6, "warmup": 3,
                                                 "what": "bf16 operands at precise_level 2 + split adapter layers: the bf16 policy whose stress "
                                                         "golden meets 1e-3 (three 16-bit K parts per block GEMM)"}
                    del eng_bp
                finally:
                    config.precise_level_policy = old_pl
            finally:
                config.set_operand_dtype(old_dt)
                config.loss_scale = old_ls

    # ---- roofline of the dominant kernel: the dense MFMA GEMM (csrc/gemm_big.h) ---------------------------
    roof = None
    if prof:
        dense = [(f, s.elapsed_time(e)) for (kind, f, s, e, _) in prof if kind == "gemm"]
        alg_bytes = sum(nb for (kind, _, _, _, nb) in prof if kind == "gemm") / len(dense)
        n = len(dense)
        flops = sum(f for f, _ in dense) / n
        avg_ms = sum(t for _, t in dense) / n
        conv = [(f, s.elapsed_time(e)) for (kind, f, s, e, _) in prof if kind == "conv"]
        if os.environ.get("ASIS_BENCH_SHAPES") and rank == 0:  # per-shape table (flops identify the shape) on stderr
            by = {}
            for (kind, f, s, e, nb) in prof:
                by.setdefault((kind, f, nb), []).append(s.elapsed_time(e))
            for (kind, f, nb), ts in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                print(f"  {kind:5s} {f / 1e9:9.1f} GF {nb / 1e6:8.1f} MB  x{len(ts) // a.steps:3d}/step  avg {sum(ts) / len(ts) * 1e3:8.1f} us  "
                      f"{f / (sum(ts) / len(ts)) / 1e9:7.1f} TF/s  total {sum(ts) / a.steps:7.2f} ms/step", file=sys.stderr)
        achieved = flops / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(("gemm_big_kernel", "gemm_p8_kernel"), ("Lb0ELb0ELi32ELi4E", "Lb0ELb0ELi64ELi1ELb1E", "gemm_p8_kernel"))
        roof = {"bound": "mfma", "kernel": "the dense LDS-DMA MFMA GEMM on v_mfma_f32_16x16x32 (all dense GEMM launches of the step): gemm_p8_kernel (csrc/gemm_p8.h, persistent 8-phase 256x256x64 form: K <= 2048 launches with >= 256 tiles), gemm_big_kernel (csrc/gemm_big.h: the one-tile-per-workgroup 8-phase form for K = 4096 and the batched V^T GEMMs, 256x128x32 two-workgroup form for the rest)",
                "achieved": round(achieved, 1), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4),
                "peak_sustained": MFMA_F16_SUSTAINED_RANDOM_TFLOPS, "frac_of_sustained": round(achieved / MFMA_F16_SUSTAINED_RANDOM_TFLOPS, 4),
                "peak_sustained_source": "bare f16 MFMA loop on random data, matrix pipe 100 % busy, clocked at 1.65 GHz by the chip "
                                         "(scripts/mfma_peak.hip, profiles/r05_mfma_sustained_peak.txt); `peak` is the nominal 2.4 GHz figure",
                "traffic": traffic, "traffic_source": traffic_src,
                "measured_in": "a second pass of the same steps after the timed region, every launch on ONE stream (encoder / V^T / "
                               "weight-gradient / dual-trunk side streams off) with a HIP event pair around each GEMM launch",
                "launches_per_step": n // a.steps, "avg_launch_ms": round(avg_ms, 4),
                "gflop_per_launch": round(flops / 1e9, 2), "algorithmic_bytes_per_launch": round(alg_bytes),
                "share_of_step_time": round(sum(t for _, t in dense) / a.steps / (elapsed / a.steps * 1e3), 3),
                "conv_gemm_share_of_step_time": round(sum(t for _, t in conv) / a.steps / (elapsed / a.steps * 1e3), 3)}

    if rank == 0:
        global_batch = a.batch * world
        out = {
            "metric": "training images/sec, ViT-L/14 588^2 adapter fine-tune" if a.config == 3 else
                      f"training images/sec, BASELINE config {a.config}",
            "value": round(global_batch * a.steps / elapsed, 3), "unit": "img/s", "n_gpus": world,
            "n_ranks_seen": (dist.get_world_size() if world > 1 else 1), "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if config.operand_dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": {
                3: f"{a.arch}/14 frozen + CAViT/CACNN adapters (n_last_blocks=4) + FeatureDecoder, "
                   f"{a.size}x{a.size}, batch {a.batch}/GPU, reference_exact train.py step "
                   "(fwd + decoder bwd + all-reduce + SGD), random-init weights" +
                   (" + train_adapters (CAViT/CACNN backward through the 4 stages and 3 frozen blocks)" if a.train_adapters else "") +
                   (" + encoder backward" if a.train_encoder else ""),
                2: f"BASELINE config 2: {a.arch}/14 frozen + CAViT/CACNN adapters + UNet head, CE + DC loss, "
                   f"{a.size}x{a.size}, batch {a.batch}/GPU (fwd + UNet bwd + all-reduce + SGD), random-init weights",
                5: f"BASELINE config 5: {a.arch}/14 (SwiGLU) frozen + CAViT/CACNN adapters + DecoderMLA head, 11 classes, soft-IoU "
                   f"loss (train_mla.py / train_multi_class.py flow), {a.size}x{a.size}, batch {a.batch}/GPU, random-init weights",
                4: (f"BASELINE config 4: {a.arch}/14 unfrozen end-to-end + DecoderSETR, CE + DC loss, {a.size}x{a.size}, "
                    f"batch {a.batch}/GPU (fwd + full bwd incl. all ViT blocks + full-gradient all-reduce + decoder SGD), "
                    "random-init weights") if a.e2e_setr else
                   (f"BASELINE config 4: {a.arch}/14 UNFROZEN in the train.py adapter flow (both ViT passes with weight "
                    f"gradients) + CAViT/CACNN + encoder + FeatureDecoder all trained, {a.size}x{a.size}, batch {a.batch}/GPU "
                    "(fwd + full bwd + bucketed all-reduce of backbone/adapter/encoder/decoder gradients + SGD), random-init weights")}[a.config],
                       "global_batch": global_batch, "image_size": a.size, "parallelism": f"dp{world}",
                       "split_precision_convs": bool(config.split_conv), "loss": loss_v,
                       "side_streams": {"encoder": bool(config.encoder_stream), "vt": bool(config.vt_stream),
                                        "wgrad": bool(config.wgrad_stream), "dual_trunk": bool(config.dual_stream)},
                       "split_attn_out": bool(getattr(eng, "split_attn_out", False)), "precise_level": int(getattr(eng, "precise_level", config.precise_level)),
                       "precise_parts": (sorted(getattr(eng, "precise_parts", config.precise_parts)) if int(getattr(eng, "precise_level", 0)) >= 2 else None),
                       "fold_attn_scale": bool(config.fold_attn_scale), "fused_qkv": bool(config.fused_qkv),
                       "dead_cacnn_elided": bool(config.elide_dead_cacnn), "c1_elided": bool(config.elide_c1),
                       "patch_embed_shared": bool(config.share_patch_embed),
                       "ln_fold": bool(config.ln_fold and config.operand_dtype == torch.float16),
                       "mx_conv": bool(config.mx_conv_on()), "mx_dense": bool(config.mx_dense_on()),
                       "conv_halo": bool(ops.CONV_HALO and config.mx_conv_on()), "fuse_cls_up": bool(ops.FUSE_CLS_UP),
                       "skipped_optimizer_steps": int(eng.optimizer.skipped_steps)},
        }
        if host_enqueue_ms is not None:
            out["host_enqueue_ms_per_step"] = round(host_enqueue_ms, 2)
        if secondary:
            out["secondary"] = secondary
        if roof:
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline and a.config == 3:
            out["cpu_baseline"] = cpu_baseline(a.arch, a.size, a.cpu_baseline_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
