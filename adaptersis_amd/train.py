"""Drop-in for the reference's `train.py`: same CLI (`train.py:654-683`), same function signatures
(`train_seg` :63, `train` :260, `validate_network` :449), same checkpoint / log formats (:244-255) — the step
body runs on the HIP engine (`backbones.engines.SegEngine`) instead of eager PyTorch modules.

    python -m adaptersis_amd.train --arch vit_large --patch_size 14 --imsize 588 --batch_size_per_gpu 12 \
        --data_path synthetic --epochs 1 --output_dir /tmp/out
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m adaptersis_amd.train ...

Differences forced by the environment (no network, no torchvision/albumentations/omegaconf here): the four
`configs/eval/vit*14_pretrain.yaml` files reduce to the arch table below; `--data_path synthetic` (default when the
path does not exist) trains on the synthetic set of SURVEY.md §8d; `--data_path <dir>` expects ``images.npy``
(N,3,H,W float in [0,1]) and ``masks.npy`` (N,H,W int) per split directory (``train/``, ``validation/``).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import math
import os
from pathlib import Path

import torch
from torch import nn

from .backbones.adapter_blocks import CACNN, CAViT
from .backbones.decoders import DecoderMLA, FeatureDecoder
from .backbones.encoders import FeatureEncoder
from . import config, parallel
from .backbones.engines import SegEngine
from .dinov2.models import vision_transformer as vits
from .utils import misc as utils
from .utils import weights as W

ARCH_FFN = {"vit_tiny_test": "mlp", "vit_small": "mlp", "vit_base": "mlp", "vit_large": "mlp", "vit_giant2": "swiglufused"}
_ENGINES = {}


class _SegData(torch.utils.data.Dataset):
    """(img float[0,1] CHW, mask long HW, index) like `tools/dataset.py:127-167`."""

    def __init__(self, path, split, imsize, n_synth=24):
        d = os.path.join(path, split)
        if os.path.isfile(os.path.join(d, "images.npy")):
            import numpy as np
            self.img = torch.from_numpy(np.load(os.path.join(d, "images.npy"))).float()
            self.msk = torch.from_numpy(np.load(os.path.join(d, "masks.npy"))).long()
        else:
            self.img, self.msk = W.synthetic_batch(n_synth, imsize, 2, seed=1 if split == "train" else 2)

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return self.img[i], self.msk[i], i


_AUGMENTERS = {}
_IDENTITY = {"crop": None, "flip": False, "rotk": 0, "alpha": 1.0, "beta": 0.0, "gamma": None}


def _to_device_batch(inp, target, train: bool):
    """Batch from the loader -> (float [B,3,S,S] in [0,1], long [B,S,S]) on the device.  uint8 HWC batches (``tools.dataset.Robomis``
    without a host transform) go through the GPU augmentation pipeline (`train.py:139-163`) when training, and through the same
    kernel with identity parameters (= `/255`, `tools/dataset.py:159`; the reference's val transform is a no-op Resize to the
    image size, `train.py:119-122`) when validating."""
    inp, target = inp.cuda(non_blocking=True), target.cuda(non_blocking=True)
    if inp.dtype != torch.uint8:
        return inp, target
    from .tools.augment import TrainAugment
    S = inp.shape[1]
    aug = _AUGMENTERS.get(S)
    if aug is None:
        aug = _AUGMENTERS[S] = TrainAugment(size=S, seed=1000 + utils.get_rank())
    return aug(inp.contiguous(), target.to(torch.uint8).contiguous(), None if train else [dict(_IDENTITY) for _ in range(inp.shape[0])])


def _open_datasets(args):
    """-> (train set, val set, collate_fn).  ``--data_path`` with the reference's layout (`images/<split>/*.png` +
    `annotations/<split>/`, `tools/dataset.py:127-141`) or decode-free ``<split>/images.npy`` uint8 arrays -> ``Robomis`` +
    GPU augmentation; otherwise the float ``.npy`` / synthetic set of ``_SegData``."""
    from .tools.dataset import Robomis, collate_u8
    p = args.data_path
    if os.path.isdir(os.path.join(p, "images", "training")) or os.path.isfile(os.path.join(p, "training", "images.npy")):
        return (Robomis(p, "training", transform=None, imsize=args.imsize), Robomis(p, "validation", transform=None, imsize=args.imsize),
                collate_u8)
    return _SegData(p, "train", args.imsize), _SegData(p, "validation", args.imsize), None


class BatchShardSampler(torch.utils.data.Sampler):
    """``batch_sampler`` of the sharded validation (``--shard_val``): the val set in its natural order cut into batches of
    ``batch_size`` exactly as the reference's sequential loader cuts it (`train.py:125-130`), rank r takes batches
    r, r + world, ...  No padding, no duplicates: ranks may hold different batch counts (validation issues no collective per
    batch — `parallel.local_batchnorm`); every batch has the composition, hence the BatchNorm statistics, of the reference's."""

    def __init__(self, n: int, batch_size: int, rank: int, world: int):
        self.batches = [list(range(i, min(i + batch_size, n))) for i in range(0, n, batch_size)][rank::world]

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def broadcast_module_states(modules, src: int = 0, group=None) -> None:
    """What ``DistributedDataParallel.__init__`` does for the reference (`train.py:84-116`): every parameter and buffer of
    the trainable / SyncBN modules takes rank ``src``'s value.  One flat fp32 message per module (plus one for integer
    buffers), no-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for m in modules:
            ts = [t for t in list(m.parameters()) + list(m.buffers())]
            for dtype in sorted({t.dtype for t in ts}, key=str):
                part = [t for t in ts if t.dtype == dtype]
                flat = torch.cat([t.detach().reshape(-1) for t in part])
                dist.broadcast(flat, src, group=group)
                o = 0
                for t in part:
                    t.copy_(flat[o:o + t.numel()].view(t.shape))
                    o += t.numel()


def _engine_for(model, backbone_encoder, cross_vit, cross_cnn, seg_decoder, lr=0.01, momentum=0.99, weight_decay=3e-5,
                mode="reference_exact", train_encoder=False):
    key = id(seg_decoder)
    if key not in _ENGINES:
        _ENGINES[key] = SegEngine(model, backbone_encoder, cross_vit, cross_cnn, seg_decoder, lr=lr, momentum=momentum,
                                  weight_decay=weight_decay, mode=mode, train_encoder=train_encoder)
    return _ENGINES[key]


def train_seg(args, head: str = "feature"):
    """``head``: "feature" = `train.py` (FeatureDecoder, SGD lr / 0.99 / 3e-5, `train.py:178-191`); "mla" = `train_mla.py`
    (DecoderMLA, SGD lr * batch * world / 16, momentum 0.9, no weight decay, `train_mla.py:178-184`)."""
    utils.init_distributed_mode(args)
    print("\n".join("%s: %s" % (k, str(v)) for k, v in sorted(dict(vars(args)).items())))
    dev = torch.device("cuda", args.gpu)
    arch = args.arch if args.arch in ARCH_FFN else "vit_large"
    D = W.VIT_CONFIGS[arch][0]
    # dinov2/eval/setup.py:62-75 + models/__init__.py:14-29 (teacher, img_size 518, layerscale 1e-5, block_chunks 0)
    model = vits.__dict__[arch](patch_size=args.patch_size, img_size=518, init_values=1e-5, ffn_layer=ARCH_FFN[arch],
                                block_chunks=0)
    if args.pretrained_weights and os.path.isfile(args.pretrained_weights):
        utils.load_pretrained_weights(model, args.pretrained_weights, args.checkpoint_key)   # dinov2/utils/utils.py:20-33
    else:
        model.load_state_dict(W.make_vit_state_dict(arch, patch_size=args.patch_size, layerscale="init"))
        print("No pretrained weights: deterministic synthetic initialisation (adaptersis_amd.utils.weights)")
    model = model.to(dev).eval()
    feature_model = model  # ModelWithIntermediateLayers(model, 4, autocast) collapses into the engine's pass A
    backbone_encoder = FeatureEncoder(embed_dim=D).to(dev)
    cross_vit = CAViT(dim=D, n_levels=3, num_heads=8, init_values=0.0, n_points=4).to(dev)
    cross_cnn = CACNN(dim=D, n_levels=1, num_heads=8, n_points=4, with_cffn=True, cffn_ratio=0.25).to(dev)
    if head == "mla":
        seg_decoder = DecoderMLA(img_size=args.imsize, mla_channels=D, num_classes=2).to(dev)
    else:
        seg_decoder = FeatureDecoder(embed_dim=D, num_classes=2, features=[D, 512, 256, 128, 64]).to(dev)
    # the reference wraps these four modules in DistributedDataParallel (`train.py:84-116`), whose constructor broadcasts
    # rank 0's parameters and buffers: without it every rank would train its own randomly initialised copy
    broadcast_module_states([backbone_encoder, cross_vit, cross_cnn, seg_decoder])
    if head == "mla":
        lr = args.lr * (args.batch_size_per_gpu * utils.get_world_size()) / 16.0  # linear scaling rule
        engine = _engine_for(model, backbone_encoder, cross_vit, cross_cnn, seg_decoder, lr=lr, momentum=0.9, weight_decay=0.0)
    else:
        engine = _engine_for(model, backbone_encoder, cross_vit, cross_cnn, seg_decoder, lr=args.lr,
                             mode="train_adapters" if getattr(args, "train_adapters", False) else "reference_exact",
                             train_encoder=getattr(args, "train_encoder", False))
    optimizer = engine.optimizer

    dataset_train, dataset_val, collate = _open_datasets(args)
    workers = args.num_workers if collate is not None else 0      # PNG decode runs in loader workers; tensors in memory do not need any
    if getattr(args, "shard_val", False) and utils.get_world_size() > 1:
        # the reference lets all ranks evaluate the whole val set; here every rank takes every world-th batch
        val_loader = torch.utils.data.DataLoader(dataset_val, num_workers=workers, pin_memory=True, collate_fn=collate,
                                                 batch_sampler=BatchShardSampler(len(dataset_val), args.batch_size_per_gpu,
                                                                                 utils.get_rank(), utils.get_world_size()))
    else:
        val_loader = torch.utils.data.DataLoader(dataset_val, batch_size=args.batch_size_per_gpu, num_workers=workers, pin_memory=True,
                                                 collate_fn=collate)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset_train, num_replicas=utils.get_world_size(),
                                                               rank=utils.get_rank())
    train_loader = torch.utils.data.DataLoader(dataset_train, sampler=sampler, batch_size=args.batch_size_per_gpu,
                                               num_workers=workers, pin_memory=True, collate_fn=collate)   # no drop_last: train.py:168-174
    print(f"Data loaded with {len(dataset_train)} train and {len(dataset_val)} val imgs.")

    class _Cosine:  # torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, epochs, eta_min=0), stepped per epoch
        def __init__(self, opt, T):
            self.opt, self.T, self.e = opt, T, 0
            self.base = [g["lr"] for g in opt.param_groups]

        def step(self):
            self.e += 1
            for g, b in zip(self.opt.param_groups, self.base):
                g["lr"] = b * (1 + math.cos(math.pi * self.e / self.T)) / 2

        def state_dict(self):
            return {"last_epoch": self.e, "base_lrs": self.base, "T_max": self.T}

        def load_state_dict(self, sd):
            self.e = sd["last_epoch"]
            for g, b in zip(self.opt.param_groups, self.base):
                g["lr"] = b * (1 + math.cos(math.pi * self.e / self.T)) / 2

    scheduler = _Cosine(optimizer, args.epochs)
    to_restore = {"epoch": 0, "best_acc": 0.0}
    # reference keys (`train.py:244-255`) + the adapters / encoder when they train (without them a resume
    # would put the restored momentum on freshly re-randomised weights)
    extra = {}
    if engine.mode == "train_adapters":
        extra = {"cross_vit": cross_vit, "cross_cnn": cross_cnn}
        if engine.train_encoder:
            extra["backbone_encoder"] = backbone_encoder
    utils.restart_from_checkpoint(os.path.join(args.output_dir, "checkpoint.pth.tar"), run_variables=to_restore,
                                  state_dict=seg_decoder, **extra, optimizer=optimizer, scheduler=scheduler)
    start_epoch, best_acc = to_restore["epoch"], to_restore["best_acc"]
    if args.evaluate:
        stats = validate_network(val_loader, model, feature_model, backbone_encoder, cross_vit, cross_cnn, seg_decoder,
                                 args.n_last_blocks, args.avgpool_patchtokens)
        print(f"Accuracy of the network on the {len(dataset_val)} test images: {stats['acc1']:.1f}%")
        return stats
    log_stats = {}
    for epoch in range(start_epoch, args.epochs):
        train_loader.sampler.set_epoch(epoch)
        train_stats = train(model, feature_model, backbone_encoder, cross_vit, cross_cnn, seg_decoder, optimizer,
                            train_loader, epoch, args.n_last_blocks, args.avgpool_patchtokens)
        scheduler.step()
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, "epoch": epoch}
        if epoch % args.val_freq == 0 or epoch == args.epochs - 1:
            test_stats = validate_network(val_loader, model, feature_model, backbone_encoder, cross_vit, cross_cnn,
                                          seg_decoder, args.n_last_blocks, args.avgpool_patchtokens)
            print(f"Accuracy at epoch {epoch} of the network on the {len(dataset_val)} test images: {test_stats['acc1']:.1f}%")
            best_acc = max(best_acc, test_stats["acc1"])
            print(f"Max accuracy so far: {best_acc:.2f}%")
            log_stats = {**log_stats, **{f"test_{k}": v for k, v in test_stats.items()}}
        if utils.is_main_process():
            Path(args.output_dir).mkdir(parents=True, exist_ok=True)
            with (Path(args.output_dir) / "log.txt").open("a") as f:
                f.write(json.dumps(log_stats) + "\n")
            ckpt = {"epoch": epoch + 1,
                    "state_dict": {"module." + k: v for k, v in seg_decoder.state_dict().items()},  # DDP prefix kept
                    "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict(), "best_acc": best_acc}
            for k, mod in extra.items():
                ckpt[k] = {"module." + n: v for n, v in mod.state_dict().items()}
            torch.save(ckpt, os.path.join(args.output_dir, "checkpoint.pth.tar"))
    print("Training of the supervised linear classifier on frozen features completed.\n"
          "Top-1 test accuracy: {acc:.1f}".format(acc=best_acc))
    return log_stats


def train(model, feature_model, backbone_encoder, cross_vit, cross_cnn, seg_decoder, optimizer, loader, epoch, n, avgpool):
    """`train.py:260-445`; the body of the loop is ``SegEngine.train_step``."""
    seg_decoder.train()
    engine = _engine_for(model, backbone_encoder, cross_vit, cross_cnn, seg_decoder)
    if optimizer is not engine.optimizer:
        raise ValueError("train(): pass the engine's optimizer (adaptersis_amd.optim.SGD over the flat bucket)")
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    # SyncBatchNorm element counts: with the reference's DistributedSampler every rank holds the same number of images in every
    # iteration (also in the short last batch); any other loader may not, so the ranks' batch sizes are exchanged per step
    ragged = parallel.collectives_on() and not isinstance(getattr(loader, "sampler", None),
                                                          torch.utils.data.distributed.DistributedSampler)
    for (inp, target, idx) in metric_logger.log_every(loader, 20, "Epoch: [{}]".format(epoch)):
        if ragged:
            parallel.set_batch_ratio(int(inp.shape[0]))
        loss = engine.train_step(*_to_device_batch(inp, target, train=True))
        torch.cuda.synchronize()  # the reference syncs and reads the loss every step (train.py:439-440)
        metric_logger.update(loss=loss.item())
        metric_logger.update(lr=optimizer.param_groups[0]["lr"])
    if ragged:
        parallel.set_batch_ratio(None)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    # overflow guard of the static 16-bit loss scale (optim.SGD): a step whose gradients hold an inf / NaN is skipped on the
    # device.  One host read per epoch; an epoch in which EVERY step was skipped means the scale is too large for this data —
    # the reference's torch.optim.SGD has no skip, so a silent stall would be a behaviour change.
    skipped = optimizer.skipped_steps
    prev = getattr(optimizer, "_skipped_reported", 0)
    optimizer._skipped_reported = skipped
    if skipped > prev:
        print(f"WARNING: {skipped - prev} of {len(loader)} optimizer steps of epoch {epoch} were skipped (non-finite gradients "
              f"under loss scale {config.loss_scale:g}); lower ASIS_LOSS_SCALE if this persists")
        if len(loader) > 0 and skipped - prev >= len(loader):
            raise RuntimeError(f"every optimizer step of epoch {epoch} was skipped: non-finite gradients under the static loss "
                               f"scale {config.loss_scale:g} (set ASIS_LOSS_SCALE to a smaller power of two)")
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def _val_meters():
    """The validation MetricLogger with its three meters created up front, in a fixed order, on EVERY rank: with ``--shard_val``
    a rank whose shard is empty (fewer batches than ranks) must issue the same barrier + all-reduce per meter in
    ``synchronize_between_processes`` as the others, or the job hangs; the summary line reads all three."""
    ml = utils.MetricLogger(delimiter="  ")
    for name in ("loss", "acc1", "dice"):
        ml.meters[name]
    return ml


def _val_summary(metric_logger) -> str:
    return "* Acc@1 {top1.global_avg:.3f} loss {losses.global_avg:.3f} Dice {dice.global_avg:.3f}".format(
        top1=metric_logger.acc1, losses=metric_logger.loss, dice=metric_logger.meters["dice"])


@torch.no_grad()
def validate_network(val_loader, model, feature_model, backbone_encoder, cross_vit, cross_cnn, seg_decoder, n, avgpool):
    """`train.py:448-651`: weighted CE ([0.1, 10]), dice = 1 - DC(logits), pixel accuracy; decoder in eval mode."""
    engine = _engine_for(model, backbone_encoder, cross_vit, cross_cnn, seg_decoder)
    metric_logger = _val_meters()
    wt = torch.tensor([0.1, 10.0], device=next(seg_decoder.parameters()).device)
    sharded = isinstance(getattr(val_loader, "batch_sampler", None), BatchShardSampler)
    with (parallel.local_batchnorm() if sharded else contextlib.nullcontext()):
        for (inp, target, idx) in metric_logger.log_every(val_loader, 20, "Test:"):
            inp, target = _to_device_batch(inp, target, train=False)
            m, dloss = engine.validate_step(inp, target, wt)
            m = m.cpu()
            bs = inp.shape[0]
            metric_logger.update(loss=float(m[0] / m[1]))
            metric_logger.meters["acc1"].update(float(m[2]) / target.numel(), n=bs)
            metric_logger.meters["dice"].update(1.0 - float(dloss), n=bs)
    if sharded:   # per-rank sums -> the whole-set averages every rank of the reference computes redundantly
        metric_logger.synchronize_between_processes()
    print(_val_summary(metric_logger))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def get_args_parser():
    p = argparse.ArgumentParser("Evaluation with semantic segmentation on RobustMIS2019")
    p.add_argument("--n_last_blocks", default=4, type=int)
    p.add_argument("--avgpool_patchtokens", default=False, type=utils.bool_flag)
    p.add_argument("--arch", default="vit_large", type=str)
    p.add_argument("--patch_size", default=14, type=int)
    p.add_argument("--imsize", default=588, type=int)
    p.add_argument("--checkpoint_key", default="teacher", type=str)
    p.add_argument("--epochs", default=100, type=int)
    p.add_argument("--lr", default=0.01, type=float)
    p.add_argument("--batch_size_per_gpu", default=12, type=int)
    p.add_argument("--dist_url", default="env://", type=str)
    p.add_argument("--local-rank", default=0, type=int)
    p.add_argument("--data_path", default="synthetic", type=str)
    p.add_argument("--num_workers", default=10, type=int)
    p.add_argument("--val_freq", default=1, type=int)
    p.add_argument("--output_dir", default=".")
    p.add_argument("--opts", default=[], nargs=argparse.REMAINDER)
    p.add_argument("--num_labels", default=1000, type=int)
    p.add_argument("--evaluate", dest="evaluate", action="store_true")
    p.add_argument("--config_file", type=str)
    p.add_argument("--pretrained_weights", type=str)
    # extension (not in the reference CLI): train CAViT / CACNN too — what `train.py:178-186` lists in its optimiser but
    # its no_grad block keeps from training
    p.add_argument("--train_adapters", action="store_true")
    p.add_argument("--train_encoder", action="store_true", help="with --train_adapters: also the CNN encoder")
    p.add_argument("--shard_val", action="store_true",
                   help="validation sharded over the ranks (every world-th batch per rank, metric sums all-reduced) instead of the "
                        "reference's whole val set on every rank (train.py:125-130,236-238); same batches, same per-batch results")
    return p


if __name__ == "__main__":
    train_seg(get_args_parser().parse_args())
