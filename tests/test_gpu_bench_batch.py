"""GPU: BASELINE configs 2 / 4 / 5 at the BENCH batch (B = 12, 588x588, full depth) on the dispatch `bench.py --config N` times.

At 12 images the stacked launches carry 42 348 rows: tile-count thresholds select other kernels than at the B = 1 / 2 of the
full-depth goldens (persistent 8-phase GEMM with ``A_lo`` parts, the MX-dense ``precise_level 2`` instances, the split sizes of
``wgrad_dense_big_kernel``, the two-batch attention launches) and from the second step of an engine the frozen trunk runs as
two concurrent streams.  Goldens: forward + loss of the imported reference modules at B = 12 with both weight sets
(tests/golden/make_golden.py --only c2_b12 / c5_b12; config 4's forward IS the `train.py` flow of ``step_b12`` — unfreezing
changes what is trained, not the forward values).  lr = 0 keeps the weights, so both steps must meet the same golden.

north_star tolerance: 1e-3 relative (rel-L2) on the logits."""
import pytest
import torch

from adaptersis_amd import config, ops
from adaptersis_amd.backbones.decoders import DecoderMLA, FeatureDecoder
from adaptersis_amd.backbones.engines import SegEngine
from adaptersis_amd.backbones.unet_parts import UNet
from adaptersis_amd.utils import weights as W
from tests.conftest import golden_err, load_golden
from tests.test_gpu_fulldepth import _modules

pytestmark = pytest.mark.gpu
TOL = 1e-3
B = 12


def _need(name, key):
    try:
        g = load_golden(name)
    except FileNotFoundError:
        pytest.skip(f"tests/golden/{name}.pt not generated")
    if key not in g:
        pytest.skip(f"{key} not in tests/golden/{name}.pt")
    return g


@pytest.mark.parametrize("mode", ["init", "kernel"])
def test_config2_bench_batch_two_steps(dev, mode):
    """ViT-B/14 (12 blocks) frozen + adapters(768) + UNet(768), CE + DC, B = 12 (`bench.py --config 2`)."""
    tag = f"c2_b12_{mode}"
    g = _need("c2_b12", f"{tag}.logits")
    D, depth, model, enc, cv, cn = _modules("vit_base", mode, dev)
    dec = UNet(D, 2); dec.load_state_dict(W.make_unet_state_dict(D, 2))
    eng = SegEngine(model, enc, cv, cn, dec.to(dev), lr=0.0, loss="ce_dc")
    img, tgt = W.synthetic_batch(B, 588)
    img, tgt = img.to(dev), tgt.to(dev)
    for step in range(2):
        taps = {}
        loss = eng.train_step(img, tgt, taps)
        e = {"x_final": golden_err(taps["x_final"], g[f"{tag}.x_final"]), "c_final": golden_err(taps["c_final"], g[f"{tag}.c_final"]),
             "logits": golden_err(taps["logits"].permute(0, 3, 1, 2), g[f"{tag}.logits"])}
        print(tag, "step", step, {k: "%.2e" % v for k, v in e.items()}, "loss", float(loss), "golden", float(g[f"{tag}.loss"]))
        assert max(e.values()) < TOL, (step, e)
        assert abs(float(loss) - float(g[f"{tag}.loss"])) < 1e-4, step
    assert eng.optimizer.skipped_steps == 0


@pytest.mark.parametrize("mode", ["init", "kernel"])
def test_config5_bench_batch_two_steps(dev, mode):
    """ViT-g/14 (40 SwiGLU blocks) frozen + adapters(1536) in the `train_mla.py` stage order + DecoderMLA, 11 classes, soft-IoU,
    B = 12 on the engine's default policy for this geometry (`bench.py --config 5`)."""
    tag = f"c5_b12_{mode}"
    g = _need("c5_b12", f"{tag}.output")
    D, depth, model, enc, cv, cn = _modules("vit_giant2", mode, dev)
    dec = DecoderMLA(img_size=588, mla_channels=D, mlahead_channels=128, num_classes=11)
    dec.load_state_dict(W.make_decoder_mla_state_dict(D, 128, 11))
    eng = SegEngine(model, enc, cv, cn, dec.to(dev), lr=0.0, momentum=0.9, weight_decay=0.0, num_classes=11, loss="iou")
    img, tgt = W.synthetic_batch(B, 588, 11)
    img, tgt = img.to(dev), tgt.to(dev)
    for step in range(2):
        taps = {}
        loss = eng.train_step(img, tgt, taps)
        e = {f"in{i}": golden_err(t.transpose(1, 2).reshape(B, D, 42, 42), g[f"{tag}.in{i}"]) for i, t in enumerate(taps["mla_inputs"])}
        e["output"] = golden_err(ops.resize_bilinear_fwd(taps["logits"], 588, 588).permute(0, 3, 1, 2), g[f"{tag}.output"])
        print(tag, "step", step, "precise_level", eng.precise_level, {k: "%.2e" % v for k, v in e.items()}, "loss", float(loss),
              "golden", float(g[f"{tag}.loss"]))
        assert max(e.values()) < TOL, (step, e)
        assert abs(float(loss) - float(g[f"{tag}.loss"])) < 1e-4, step
    assert eng.optimizer.skipped_steps == 0


@pytest.mark.parametrize("mode,tag", [("kernel", "step_b12_kernel"), ("init", "step_b12_exact")])
def test_config4_bench_batch_two_steps(dev, mode, tag):
    """ViT-L/14 (24 blocks) UNFROZEN in the `train.py` adapter flow at B = 12 (`bench.py --config 4`): forward taps + loss against
    the reference's B = 12 forward (``step_b12``); the training forward keeps every activation (19 GB), runs the two-batch
    attention launches with their log-sum-exp, and the backward the two-batch ``asis_attention_bwd_rows``."""
    g = _need("step_b12", f"{tag}.logits")
    D, depth, model, enc, cv, cn = _modules("vit_large", mode, dev, train=True)
    feats = (D, 512, 256, 128, 64)
    dec = FeatureDecoder(embed_dim=D, num_classes=2, features=list(feats)); dec.load_state_dict(W.make_feature_decoder_state_dict(D, 2, features=feats))
    eng = SegEngine(model, enc, cv, cn, dec.to(dev), lr=0.0, mode="train_adapters", train_encoder=True, train_backbone=True)
    img, tgt = W.synthetic_batch(B, 588)
    img, tgt = img.to(dev), tgt.to(dev)
    for step in range(2):
        taps = {}
        loss = eng.train_step(img, tgt, taps)
        e = {"cat": golden_err(taps["cat"].float().permute(0, 3, 1, 2), g[f"{tag}.cat"]),
             "logits": golden_err(taps["logits"].permute(0, 3, 1, 2), g[f"{tag}.logits"])}
        print(tag, "config 4 step", step, {k: "%.2e" % v for k, v in e.items()}, "loss", float(loss), "golden", float(g[f"{tag}.loss"]))
        assert max(e.values()) < TOL, (step, e)
        assert abs(float(loss) - float(g[f"{tag}.loss"])) < 1e-4, step
        # every parameter group received finite, non-zero gradients
        for nm, bucket in (("vit", eng.vit_bucket), ("adapter", eng.adapter_bucket), ("encoder", eng.encoder_bucket), ("decoder", eng.bucket)):
            gsum = float(bucket.grad.float().abs().sum())
            assert gsum == gsum and gsum > 0.0, (nm, gsum)
    assert eng.optimizer.skipped_steps == 0
