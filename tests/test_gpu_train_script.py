"""GPU: the drop-in training script (`adaptersis_amd.train` = reference `train.py` API) and the validation
metrics path (`validate_network`, train.py:448-651) against the oracle."""
import json
import os

import pytest
import torch

from adaptersis_amd import train as T
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.test_gpu_step import build_engine

pytestmark = pytest.mark.gpu


def test_validate_step_vs_oracle(dev):
    eng, sds = build_engine("vit_tiny_test", "kernel", dev, (128, 32, 16, 16, 8))
    img, tgt = W.synthetic_batch(2, 224, seed=3)
    wt = torch.tensor([0.1, 10.0], device=dev)
    m, dloss = eng.validate_step(img.to(dev), tgt.to(dev), wt)
    osd = {k: {n: t.clone() for n, t in v.items()} for k, v in sds.items()}
    with torch.no_grad():
        cat = O.adapter_forward(img, osd["vit"], osd["enc"], osd["cv"], osd["cn"], 2)
        loss, dice, acc = O.validate_metrics(cat, tgt, osd["dec"])
    m = m.cpu()
    assert abs(float(m[0] / m[1]) - float(loss)) < 2e-3 * max(1.0, abs(float(loss)))
    assert abs((1.0 - float(dloss)) - float(dice)) < 1e-3
    assert abs(float(m[2]) / tgt.numel() - float(acc)) < 2e-3
    assert eng.seg_decoder.training  # mode restored


def test_train_seg_end_to_end(dev, tmp_path):
    """One epoch of the drop-in script on the synthetic set: loss decreases, log + checkpoint in reference format,
    resume works."""
    args = T.get_args_parser().parse_args(["--arch", "vit_tiny_test", "--imsize", "224", "--batch_size_per_gpu", "4",
                                           "--epochs", "2", "--lr", "0.05", "--data_path", "synthetic",
                                           "--output_dir", str(tmp_path)])
    T._ENGINES.clear()
    stats = T.train_seg(args)
    lines = [json.loads(l) for l in open(os.path.join(tmp_path, "log.txt"))]
    assert len(lines) == 2 and {"train_loss", "train_lr", "test_loss", "test_acc1", "test_dice", "epoch"} <= set(lines[0])
    assert lines[1]["train_loss"] < lines[0]["train_loss"]
    ck = torch.load(os.path.join(tmp_path, "checkpoint.pth.tar"), map_location="cpu")
    assert set(ck) == {"epoch", "state_dict", "optimizer", "scheduler", "best_acc"} and ck["epoch"] == 2
    assert all(k.startswith("module.") for k in ck["state_dict"])
    T._ENGINES.clear()
    args.evaluate = True
    ev = T.train_seg(args)  # resumes from the checkpoint and only validates
    assert abs(ev["acc1"] - lines[1]["test_acc1"]) < 1e-6


def test_train_mla_script(dev, tmp_path):
    """`train_mla.py` drop-in: MLA head, linear-scaled lr, momentum 0.9 / no weight decay, `--local_rank` spelling."""
    from adaptersis_amd import train_mla as TM
    from adaptersis_amd.backbones.decoders import DecoderMLA
    args = TM.get_args_parser().parse_args(["--arch", "vit_tiny_test", "--imsize", "224", "--batch_size_per_gpu", "4",
                                            "--epochs", "2", "--lr", "0.08", "--data_path", "synthetic", "--local_rank", "0",
                                            "--output_dir", str(tmp_path)])
    T._ENGINES.clear()
    TM.train_seg(args)
    (eng,) = T._ENGINES.values()
    assert isinstance(eng.seg_decoder, DecoderMLA) and eng.is_mla
    g = eng.optimizer.param_groups[0]
    assert g["momentum"] == 0.9 and g["weight_decay"] == 0.0 and abs(g["initial_lr"] - 0.08 * 4 / 16.0) < 1e-12
    lines = [json.loads(l) for l in open(os.path.join(tmp_path, "log.txt"))]
    assert len(lines) == 2 and lines[1]["train_loss"] < lines[0]["train_loss"]
    ck = torch.load(os.path.join(tmp_path, "checkpoint.pth.tar"), map_location="cpu")
    assert any(k.startswith("module.mlahead.") for k in ck["state_dict"])
    T._ENGINES.clear()


def test_train_adapters_checkpoint_holds_adapters_and_encoder(dev, tmp_path):
    """SURVEY.md §5 / ADVICE r1: with --train_adapters --train_encoder the checkpoint also carries CAViT, CACNN and the encoder
    (reference keys untouched); a resume restores them onto the freshly constructed modules together with the optimiser
    state of all three buckets."""
    argv = ["--arch", "vit_tiny_test", "--imsize", "224", "--batch_size_per_gpu", "4", "--epochs", "1", "--lr", "0.05",
            "--data_path", "synthetic", "--output_dir", str(tmp_path), "--train_adapters", "--train_encoder"]
    T._ENGINES.clear()
    T.train_seg(T.get_args_parser().parse_args(argv))
    (eng,) = T._ENGINES.values()
    assert eng.optimizer.skipped_steps == 0          # overflow guard of the static loss scale never fired
    want = {"cv": eng.cross_vit.state_dict()["attn.value_proj.weight"].cpu().clone(),
            "cn": eng.cross_cnn.state_dict()["ffn.fc2.weight"].cpu().clone(),
            "enc": eng.backbone_encoder.state_dict()["conv3.0.weight"].cpu().clone(),
            "mom": [b.momentum.cpu().clone() for b in eng.optimizer.buckets]}
    ck = torch.load(os.path.join(tmp_path, "checkpoint.pth.tar"), map_location="cpu")
    assert {"epoch", "state_dict", "optimizer", "scheduler", "best_acc", "cross_vit", "cross_cnn", "backbone_encoder"} == set(ck)
    assert all(k.startswith("module.") for k in ck["cross_vit"]) and len(ck["optimizer"]["state"]) == 3
    T._ENGINES.clear()
    torch.manual_seed(1234)                          # the re-created modules start from a different random init
    args = T.get_args_parser().parse_args(argv)
    args.evaluate = True
    T.train_seg(args)
    (eng2,) = T._ENGINES.values()
    assert eng2 is not eng
    assert torch.equal(eng2.cross_vit.state_dict()["attn.value_proj.weight"].cpu(), want["cv"])
    assert torch.equal(eng2.cross_cnn.state_dict()["ffn.fc2.weight"].cpu(), want["cn"])
    assert torch.equal(eng2.backbone_encoder.state_dict()["conv3.0.weight"].cpu(), want["enc"])
    # (the encoder's SyncBatchNorm stays in train mode during validation, like the reference: its running statistics move on)
    for b, m in zip(eng2.optimizer.buckets, want["mom"]):
        assert torch.equal(b.momentum.cpu(), m)
    T._ENGINES.clear()


def test_sgd_overflow_guard_skips_the_step(dev):
    """A non-finite gradient anywhere in the buckets (saturated 16-bit gradient tensor under the static loss scale) leaves
    parameters and momentum untouched and is counted; the next finite step proceeds."""
    from adaptersis_amd import optim
    lin = torch.nn.Linear(8, 8).to(dev)
    lin2 = torch.nn.Linear(8, 4).to(dev)
    b1, b2 = optim.FlatBucket(list(lin.named_parameters())), optim.FlatBucket(list(lin2.named_parameters()))
    opt = optim.SGD([b1, b2], lr=0.1, momentum=0.9)
    w0, v0 = b1.flat.clone(), b2.flat.clone()
    b1.grad.fill_(1.0); b2.grad.fill_(1.0)
    b2.grad[3] = float("inf")
    opt.step()
    assert torch.equal(b1.flat, w0) and torch.equal(b2.flat, v0) and opt.skipped_steps == 1
    b2.grad[3] = float("nan")
    opt.step()
    assert torch.equal(b1.flat, w0) and opt.skipped_steps == 2
    b2.grad[3] = 1.0
    opt.step()
    assert torch.allclose(b1.flat, w0 - 0.1) and torch.allclose(b2.flat, v0 - 0.1) and opt.skipped_steps == 2


def test_train_seg_on_png_folder_with_gpu_augmentation(dev, tmp_path):
    """The reference's on-disk layout (`tools/dataset.py:127-141`: images/<split>/*.png + annotations/<split>/) through the
    drop-in script: PIL decode in the loader, uint8 batches, the augmentation kernel on the device, one epoch + validation."""
    import numpy as np
    from PIL import Image
    r = np.random.RandomState(0)
    for split, n in (("training", 8), ("validation", 4)):
        (tmp_path / "data" / "images" / split).mkdir(parents=True)
        (tmp_path / "data" / "annotations" / split).mkdir(parents=True)
        for i in range(n):
            img = (r.randint(0, 256, (56, 56, 3)).astype(np.uint8)).repeat(4, 0).repeat(4, 1)          # 224 x 224, blocky
            msk = ((r.random_sample((56, 56)) > 0.7).astype(np.uint8) * 255).repeat(4, 0).repeat(4, 1)
            Image.fromarray(img).save(tmp_path / "data" / "images" / split / f"f{i:03d}.png")
            Image.fromarray(msk).save(tmp_path / "data" / "annotations" / split / f"f{i:03d}.png")
    args = T.get_args_parser().parse_args(["--arch", "vit_tiny_test", "--imsize", "224", "--batch_size_per_gpu", "4", "--epochs", "1",
                                           "--lr", "0.05", "--data_path", str(tmp_path / "data"), "--num_workers", "0",
                                           "--output_dir", str(tmp_path / "out")])
    T._ENGINES.clear()
    T._AUGMENTERS.clear()
    stats = T.train_seg(args)
    assert 224 in T._AUGMENTERS                       # the uint8 path was taken
    assert {"train_loss", "test_loss", "test_acc1", "test_dice"} <= set(stats) and stats["train_loss"] == stats["train_loss"]
    assert 0.0 <= stats["test_acc1"] <= 1.0
    T._ENGINES.clear()
