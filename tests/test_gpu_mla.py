"""GPU: the `train_mla.py` path (BASELINE config 5 ingredients): DecoderMLA head, the block -> CACNN -> CAViT
interleave with the repeated-block quirk, SwiGLU (ViT-g) blocks, and the multi-class soft-IoU loss."""
import pytest
import torch
import torch.nn.functional as F

from adaptersis_amd import ops
from adaptersis_amd.backbones.adapter_blocks import CACNN, CAViT
from adaptersis_amd.backbones.decoders import DecoderMLA
from adaptersis_amd.backbones.encoders import FeatureEncoder
from adaptersis_amd.backbones.engines import SegEngine
from adaptersis_amd.dinov2.models import vision_transformer as vits
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import golden_err, load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-3


def build_mla_engine(arch, mode, dev, mlahead=128, num_classes=2, loss="dice", img=588):
    D, depth, heads, ffn = W.VIT_CONFIGS[arch]
    sds = dict(vit=W.make_vit_state_dict(arch, layerscale=("kernel" if mode == "kernel" else "init")),
               enc=W.make_encoder_state_dict(D), cv=W.make_cavit_state_dict(D, mode=mode),
               cn=W.make_cacnn_state_dict(D, mode=mode), dec=W.make_decoder_mla_state_dict(D, mlahead, num_classes))
    model = vits.__dict__[arch](patch_size=14, img_size=518, init_values=1e-5, ffn_layer=ffn, block_chunks=0)
    model.load_state_dict(sds["vit"])
    enc = FeatureEncoder(embed_dim=D); enc.load_state_dict(sds["enc"])
    cv = CAViT(dim=D, n_levels=3, num_heads=8, init_values=0.0, n_points=4); cv.load_state_dict(sds["cv"])
    cn = CACNN(dim=D, n_levels=1, num_heads=8, n_points=4, with_cffn=True, cffn_ratio=0.25); cn.load_state_dict(sds["cn"])
    dec = DecoderMLA(img_size=img, mla_channels=D, mlahead_channels=mlahead, num_classes=num_classes)
    dec.load_state_dict(sds["dec"])
    eng = SegEngine(model.to(dev).eval(), enc.to(dev), cv.to(dev), cn.to(dev), dec.to(dev), lr=0.01, momentum=0.9,
                    weight_decay=0.0, num_classes=num_classes, loss=loss)
    return eng, sds


def test_swiglu_block_vs_oracle(dev):
    arch = "vit_tiny_swiglu"
    sd = W.make_vit_state_dict(arch)
    m = vits.vit_tiny_swiglu(img_size=518, init_values=1e-5, ffn_layer="swiglufused", block_chunks=0)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = W.tensor("sw.x", (2, 257, 128), 1.0)
    assert rel_l2(m.blocks[0](x.to(dev)), O.block(x, sd, "blocks.0", 2)) < TOL / 2
    img, _ = W.synthetic_batch(1, 224)
    ref = O.forward_features(img, sd, 2)
    out = m(img.to(dev), is_training=True)
    assert rel_l2(out["x_norm_patchtokens"], ref["x_norm_patchtokens"]) < TOL


@pytest.mark.parametrize("C", [2, 11])
def test_soft_iou_loss_fwd_bwd(dev, C):
    """`segloss/iou_multi.py:9-49` after the softmax of `train_multi_class.py:391-392` (softmax twice)."""
    B, h, H = 2, 40, 56
    lg = W.tensor(f"iou.lg{C}", (B, C, h, h), 3.0).requires_grad_()
    tg = W.synthetic_batch(B, H, C)[1]
    out = torch.softmax(F.interpolate(lg, size=(H, H), mode="bilinear"), 1)
    ref = O.iou_loss(out, tg, num_classes=C)
    ref.backward()
    lg_n = lg.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    loss, coef, _ = ops.dice_fwd(lg_n, tg.to(dev), 2, 1e-6, 512.0, mode=1)
    assert abs(float(loss) - float(ref)) < 2e-6
    dz = ops.dice_bwd(lg_n, tg.to(dev), coef, 2)
    d32, _ = ops.resize_bilinear_bwd(dz, h, h, torch.float32)
    assert rel_l2(d32, lg.grad.permute(0, 2, 3, 1) * 512.0) < 1e-4


def test_decoder_mla_module_forward_backward(dev):
    """Reference-shaped module call + torch autograd through the HIP backward, vs the oracle (exact inputs)."""
    D, hw, B, Cm = 64, 12, 2, 16
    sd = W.make_decoder_mla_state_dict(D, Cm, 2)
    m = DecoderMLA(img_size=hw * 14, mla_channels=D, mlahead_channels=Cm)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    ins = [W.tensor(f"mla.i{i}", (B, D, hw, hw), 1.0) for i in range(4)]
    g = load_golden("small")
    y = m(*[t.to(dev) for t in ins])
    assert golden_err(y, g["mla.out"]) < TOL
    tgt = W.synthetic_batch(B, hw * 14, 2)[1]
    from adaptersis_amd.segloss.dice import resize_softmax_dc
    loss = resize_softmax_dc(y, tgt.to(dev))
    loss.backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    oloss = O.train_step_loss_mla(ins, tgt, p, 2, "dice")
    oloss.backward()
    assert abs(float(loss) - float(oloss)) < 1e-5
    errs = {k: rel_l2(v.grad, p[k].grad) for k, v in m.named_parameters() if float(p[k].grad.abs().max()) > 1e-7}
    print("MLA grads:", {k: "%.1e" % e for k, e in errs.items()})
    # wgrad runs on the hi halves only (dy and x each rounded to 16 bits): the sum over pixels of a mean-free dy
    # cancels, so the relative error of dW sits at a few 1e-3 here (BatchNorm gradients, taken in fp32, are 1e-5)
    assert max(errs.values()) < 1e-2, errs


def test_train_mla_step_vs_reference_golden(dev):
    """Whole `train_mla.py` step (ViT-L width, 4 blocks, 588x588) against the golden from the imported reference."""
    g = load_golden("mla")
    eng, _ = build_mla_engine("vit_large_d4", "kernel", dev)
    img, tgt = W.synthetic_batch(1, 588)
    taps = {}
    loss = eng.train_step(img.to(dev), tgt.to(dev), taps)
    for i, t in enumerate(taps["mla_inputs"]):
        e = golden_err(t.transpose(1, 2).reshape(1, 1024, 42, 42), g[f"mla_kernel.in{i}"])
        assert e < TOL, (i, e)
    out = ops.resize_bilinear_fwd(taps["logits"], 588, 588).permute(0, 3, 1, 2)
    e_out = golden_err(out, g["mla_kernel.output"])
    print("train_mla step: output rel-L2", e_out, "loss", float(loss), float(g["mla_kernel.loss"]))
    assert e_out < TOL
    assert abs(float(loss) - float(g["mla_kernel.loss"])) < 1e-4
    gerr = {k: golden_err(v, g[f"mla_kernel.grad.{k}"]) for k, v in eng.bucket.views.items()
            if float(g[f"mla_kernel.grad.{k}"]["sumsq"]) > 1e-16}
    assert max(gerr.values()) < 1e-1, gerr


def test_multiclass_mla_step_vs_oracle(dev):
    """11 classes + soft-IoU loss (BASELINE config 5 ingredients) at toy width, two steps incl. SGD (momentum 0.9)."""
    eng, sds = build_mla_engine("vit_tiny_test", "kernel", dev, mlahead=16, num_classes=11, loss="iou", img=224)
    osd = {k: {n: t.clone() for n, t in v.items()} for k, v in sds.items()}
    params = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in osd["dec"].items()}
    bufs = {}
    for step in range(2):
        img, tgt = W.synthetic_batch(2, 224, 11, seed=step)
        taps = {}
        loss = eng.train_step(img.to(dev), tgt.to(dev), taps)
        with torch.no_grad():
            maps = O.mla_forward(img, osd["vit"], osd["enc"], osd["cv"], osd["cn"], 2, update_bn=True)
        for p in params.values():
            p.grad = None
        ot = {}
        oloss = O.train_step_loss_mla(maps, tgt, params, 11, "iou", ot, update_bn=True)
        oloss.backward()
        out = ops.resize_bilinear_fwd(taps["logits"], 224, 224).permute(0, 3, 1, 2)
        assert rel_l2(out, ot["out"]) < 1e-3, step
        assert abs(float(loss) - float(oloss)) < 1e-4, step
        names = [k for k, v in params.items() if v.requires_grad]
        with torch.no_grad():
            O.sgd_momentum_step({k: params[k] for k in names}, {k: params[k].grad for k in names}, bufs, 0.01, 0.9, 0.0)
        live = dict(eng.seg_decoder.named_parameters())
        assert max(rel_l2(live[k], params[k]) for k in names) < 1e-3, step
