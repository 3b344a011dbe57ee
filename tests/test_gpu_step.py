"""Whole training step (`train.py:268-436`) on the GPU through ``SegEngine``:
 * tiny geometry vs the CPU oracle (full tensors, two consecutive steps incl. SGD + BN buffers);
 * ViT-L/14 588x588 B=1 vs the goldens captured from the imported reference (both the
   reference_exact initialisation and non-degenerate 'kernel' weights so that the adapter stream
   is not multiplied away by gamma = 0).
north_star tolerance: 1e-3 relative on the logits (rel-L2 over the tensor)."""
import copy

import pytest
import torch

from adaptersis_amd.backbones.adapter_blocks import CACNN, CAViT
from adaptersis_amd.backbones.decoders import FeatureDecoder
from adaptersis_amd.backbones.encoders import FeatureEncoder
from adaptersis_amd.backbones.engines import SegEngine
from adaptersis_amd.dinov2.models import vision_transformer as vits
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import golden_err, load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-3
# Step-level gradients inherit the forward's ~5e-4 feature error, and the Dice gradient is ill-conditioned in
# the features: the fp32 CPU oracle itself moves decoder_1 gradients by ~6 % under a 4e-4 relative input
# perturbation (tests/test_grad_conditioning.py).  The backward KERNELS are checked on exact inputs in
# tests/test_gpu_modules.py (<= 3e-4).
GRAD_TOL = 1e-1


def build_engine(arch, mode, dev, features=None, lr=0.01):
    D, depth, heads, ffn = W.VIT_CONFIGS[arch]
    features = features or (D, 512, 256, 128, 64)
    sds = dict(vit=W.make_vit_state_dict(arch, layerscale=("kernel" if mode == "kernel" else "init")),
               enc=W.make_encoder_state_dict(D), cv=W.make_cavit_state_dict(D, mode=mode),
               cn=W.make_cacnn_state_dict(D, mode=mode), dec=W.make_feature_decoder_state_dict(D, 2, features=features))
    model = vits.__dict__[arch](patch_size=14, img_size=518, init_values=1e-5, ffn_layer=ffn, block_chunks=0)
    model.load_state_dict(sds["vit"])
    enc = FeatureEncoder(embed_dim=D); enc.load_state_dict(sds["enc"])
    cv = CAViT(dim=D, n_levels=3, num_heads=8, init_values=0.0, n_points=4); cv.load_state_dict(sds["cv"])
    cn = CACNN(dim=D, n_levels=1, num_heads=8, n_points=4, with_cffn=True, cffn_ratio=0.25); cn.load_state_dict(sds["cn"])
    dec = FeatureDecoder(embed_dim=D, num_classes=2, features=list(features)); dec.load_state_dict(sds["dec"])
    eng = SegEngine(model.to(dev).eval(), enc.to(dev), cv.to(dev), cn.to(dev), dec.to(dev), lr=lr)
    return eng, sds


def test_tiny_step_vs_oracle_two_steps(dev):
    """D=128 / depth 4 engine at 224x224 (a geometry the reference itself cannot run: SURVEY.md fact 3)."""
    arch, mode, size, B = "vit_tiny_test", "kernel", 224, 2
    feats = (128, 32, 16, 16, 8)
    eng, sds = build_engine(arch, mode, dev, feats, lr=0.05)
    osd = {k: {n: t.clone() for n, t in v.items()} for k, v in sds.items()}
    dec_params = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k)
                  for k, v in osd["dec"].items()}
    bufs = {}
    for step in range(2):
        img, tgt = W.synthetic_batch(B, size, seed=step)
        taps = {}
        loss = eng.train_step(img.to(dev), tgt.to(dev), taps)
        otaps = {}
        with torch.no_grad():
            ocat = O.adapter_forward(img, osd["vit"], osd["enc"], osd["cv"], osd["cn"], 2, taps=otaps, update_bn=True)
        for p in dec_params.values():
            p.grad = None
        oloss = O.train_step_loss(ocat, tgt, dec_params, 2, otaps, update_bn=True)
        oloss.backward()
        e_cat = rel_l2(taps["cat"].float().permute(0, 3, 1, 2), ocat)
        e_c4 = rel_l2(taps["c"][:, -49:], otaps["c4"])
        e_lg = rel_l2(taps["logits"].permute(0, 3, 1, 2), otaps["logits"])
        print(f"tiny step {step}: cat {e_cat:.2e} c4 {e_c4:.2e} logits {e_lg:.2e}")
        assert e_cat < TOL, step
        assert e_lg < TOL, step
        assert abs(float(loss) - float(oloss)) < 1e-4, step
        names = [k for k, v in dec_params.items() if v.requires_grad]
        errs = {k: rel_l2(eng.bucket.views[k], dec_params[k].grad) for k in names
                if float(dec_params[k].grad.abs().max()) > 1e-7}
        assert max(errs.values()) < GRAD_TOL, errs
        with torch.no_grad():
            O.sgd_momentum_step({k: dec_params[k] for k in names}, {k: dec_params[k].grad for k in names}, bufs, 0.05)
        for k in names:
            assert rel_l2(dict(eng.seg_decoder.named_parameters())[k], dec_params[k]) < 1e-3, (step, k)
    sd = eng.seg_decoder.state_dict()
    assert int(sd["decoder_1.1.num_batches_tracked"]) == 2
    assert rel_l2(sd["decoder_3.1.running_var"], dec_params["decoder_3.1.running_var"]) < 2e-3


@pytest.mark.parametrize("mode,tag", [("init", "step_exact"), ("kernel", "step_kernel")])
def test_vitl_588_step_vs_reference_golden(dev, mode, tag):
    g = load_golden("step")
    eng, _ = build_engine("vit_large", mode, dev)
    img, tgt = W.synthetic_batch(1, 588)
    taps = {}
    loss = eng.train_step(img.to(dev), tgt.to(dev), taps)
    e = {"cat": golden_err(taps["cat"].float().permute(0, 3, 1, 2), g[f"{tag}.cat"]),
         "x_final": golden_err(taps["x_final"], g[f"{tag}.x_final"]),
         "c_final": golden_err(taps["c_final"], g[f"{tag}.c_final"]),
         "logits": golden_err(taps["logits"].permute(0, 3, 1, 2), g[f"{tag}.logits"])}
    print(tag, {k: "%.2e" % v for k, v in e.items()}, "loss", float(loss), "golden", float(g[f"{tag}.loss"]))
    # 1e-3 (north_star) on both: the reference configuration (7e-6 measured) and the stress golden (LayerScale gamma up to
    # 0.5 in all 48 block evaluations, 1.5x qkv weights, adapter gamma != 0: 9.0e-4 measured)
    assert e["logits"] < TOL, e
    assert e["cat"] < TOL and e["x_final"] < TOL and e["c_final"] < TOL, e
    assert abs(float(loss) - float(g[f"{tag}.loss"])) < 1e-4
    gerr = {}
    for k, v in eng.bucket.views.items():
        gold = g[f"{tag}.grad.{k}"]
        if float(gold["sumsq"]) < 1e-16:
            continue
        gerr[k] = golden_err(v, gold)
    print(tag, "grad rel-L2:", {k: "%.1e" % v for k, v in gerr.items()})
    # reference configuration: forward agrees to 7e-6, so no ReLU branch moves and the step-level gradients hold 1e-2
    # (1.8e-3 measured); stress golden: forward 9e-4 -> step-level conditioning bound (tests/test_grad_conditioning.py)
    assert max(gerr.values()) < (1e-2 if mode == "init" else GRAD_TOL), gerr


@pytest.mark.parametrize("mode,tag", [("kernel", "step_b12_kernel"), ("init", "step_b12_exact")])
def test_vitl_588_headline_batch_two_steps(dev, mode, tag):
    """The HEADLINE batch (`README.md:45-61`: --batch_size_per_gpu 12) in the exact dispatch bench.py times: at B = 12 the
    stacked launches carry 42 348 rows, so q|k / proj / fc1 run on the persistent 8-phase GEMM (>= 256 tiles), the attention is
    the folded two-segment launch, and FROM THE SECOND STEP of an engine the trunk runs as two concurrent streams
    (config.dual_stream).  Golden: forward + loss of the imported reference modules at B = 12
    (tests/golden/make_golden.py --only step_b12).  lr = 0 keeps the weights, so BOTH steps must meet the same golden
    (train-mode BatchNorm uses batch statistics; running buffers do not enter)."""
    from adaptersis_amd import config
    g = load_golden("step_b12")
    if f"{tag}.logits" not in g:
        pytest.skip(f"{tag} not in tests/golden/step_b12.pt")
    eng, _ = build_engine("vit_large", mode, dev, lr=0.0)
    img, tgt = W.synthetic_batch(12, 588)
    img, tgt = img.to(dev), tgt.to(dev)
    for step in range(2):
        taps = {}
        loss = eng.train_step(img, tgt, taps)
        e = {"cat": golden_err(taps["cat"].float().permute(0, 3, 1, 2), g[f"{tag}.cat"]),
             "x_final": golden_err(taps["x_final"], g[f"{tag}.x_final"]),
             "logits": golden_err(taps["logits"].permute(0, 3, 1, 2), g[f"{tag}.logits"])}
        print(tag, "step", step, "dual_stream" if (step and config.dual_stream) else "stacked",
              {k: "%.2e" % v for k, v in e.items()}, "loss", float(loss), "golden", float(g[f"{tag}.loss"]))
        assert e["logits"] < TOL and e["cat"] < TOL and e["x_final"] < TOL, (step, e)
        assert abs(float(loss) - float(g[f"{tag}.loss"])) < 1e-4, step
    assert eng.optimizer.skipped_steps == 0


def test_dual_stream_trunk_is_deterministic_and_equivalent(dev):
    """config.dual_stream: the two ViT passes of the trunk as two concurrent launch streams (engines.SegEngine._trunk_dual)
    instead of one row-stacked stream.  The half-size GEMMs take other tile forms than the stacked ones (fewer tiles than the
    8-phase form's threshold), so values move by an fp16 ulp here and there — the dual path must reproduce ITSELF bit for bit
    (a race between the streams would not) and stay within 2e-4 of the in-order path (both hold the goldens: the config /
    step tests pass under ASIS_DUAL_STREAM=1 and 0).  ViT-L width x 4 blocks at 588^2, batch 2."""
    from adaptersis_amd import config
    eng, _ = build_engine("vit_large_d4", "kernel", dev)
    img, _ = W.synthetic_batch(2, 588)
    img = img.to(dev)
    old = config.dual_stream
    try:
        config.dual_stream = False
        a = eng.features(img)
        config.dual_stream = True
        outs = [eng.features(img) for _ in range(3)]
    finally:
        config.dual_stream = old
    torch.cuda.synchronize()
    for b in outs[1:]:
        assert torch.equal(outs[0][0], b[0]) and (b[1] is None or torch.equal(outs[0][1], b[1]))
    assert rel_l2(outs[0][0].float(), a[0].float()) < 2e-4
