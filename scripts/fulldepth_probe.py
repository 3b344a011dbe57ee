"""Diagnostic for the full-depth stress goldens (tests/test_gpu_fulldepth.py): where does the logits error come from?
For config 2 (ViT-B/14, 12 blocks, UNet(768)):
   (a) the GPU step as shipped: x_final / logits vs the golden;
   (b) the same with config.precise_attention;
   (c) the UNet head fed with the ORACLE's fp32 adapter stream (oracle = test infrastructure, CPU): the head's own error;
   (d) the oracle's own head fed with the GPU's x_final: the amplification of the input error by the fp32 function itself.
usage: python scripts/fulldepth_probe.py c2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import config, ops
from adaptersis_amd.utils import weights as W
from tests.conftest import golden_err, load_golden, rel_l2
from tests.test_gpu_fulldepth import _modules
from adaptersis_amd.backbones.engines import SegEngine
from adaptersis_amd.backbones.unet_parts import UNet
from oracle import ref_torch as O

dev = torch.device("cuda:0")
torch.set_num_threads(16)


def c2(mode="kernel"):
    g, tag = load_golden("c2full"), f"c2full_{mode}"
    img, tgt = W.synthetic_batch(1, 588)

    def run(label):
        D, depth, model, enc, cv, cn = _modules("vit_base", mode, dev)
        dec = UNet(D, 2); dec.load_state_dict(W.make_unet_state_dict(D, 2))
        eng = SegEngine(model, enc, cv, cn, dec.to(dev), lr=0.01, loss="ce_dc")
        taps = {}
        eng.train_step(img.to(dev), tgt.to(dev), taps)
        ex = golden_err(taps["x_final"], g[f"{tag}.x_final"])
        el = golden_err(taps["logits"].permute(0, 3, 1, 2), g[f"{tag}.logits"])
        print(f"{label}: x_final {ex:.2e} logits {el:.2e}", flush=True)
        return eng, taps
    eng, taps = run("(a) shipped")
    config.precise_attention = True
    run("(b) precise_attention")
    config.precise_attention = False
    # oracle adapter stream (fp32, CPU)
    D, depth, heads, _ = W.VIT_CONFIGS["vit_base"]
    vsd = W.make_vit_state_dict("vit_base", layerscale="kernel" if mode == "kernel" else "init")
    esd, csd, nsd = W.make_encoder_state_dict(D), W.make_cavit_state_dict(D, mode=mode), W.make_cacnn_state_dict(D, mode=mode)
    otaps = {}
    with torch.no_grad():
        O.adapter_forward(img, vsd, {k: t.clone() for k, t in esd.items()}, csd, nsd, heads, taps=otaps)
    xo = otaps["x_stage3"]                      # [1, N, D] fp32
    print(f"    GPU x_final vs oracle (full tensor): {rel_l2(taps['x_final'], xo):.2e}", flush=True)
    # (c) GPU head on the oracle's stream
    dec = eng.seg_decoder
    dec.load_state_dict(W.make_unet_state_dict(D, 2))   # undo the SGD step of (a)
    dec.to(dev)
    xs = xo.to(dev).reshape(-1, D).contiguous()
    hi = ops.cast_pad(xs, D, config.operand_dtype).view(1, 42, 42, D)
    lo = ops.cast_pad(xs, D, config.operand_dtype, part=1).view(1, 42, 42, D)
    with torch.no_grad():
        logits, _ = dec._forward_core(hi, lo, save=True, training=True)
    print(f"(c) GPU UNet on the oracle's x_final: logits {golden_err(logits.permute(0, 3, 1, 2), g[f'{tag}.logits']):.2e}", flush=True)
    # (d) oracle head on the GPU's stream
    usd = W.make_unet_state_dict(D, 2)
    with torch.no_grad():
        xm_g = taps["x_final"].float().cpu().reshape(1, 42 * 42, D).transpose(1, 2).reshape(1, D, 42, 42)
        xm_o = xo.transpose(1, 2).reshape(1, D, 42, 42)
        yo = O.unet(xm_o, {k: v.clone() for k, v in usd.items()}, update_bn=False)
        yg = O.unet(xm_g, {k: v.clone() for k, v in usd.items()}, update_bn=False)
    print(f"(d) fp32 oracle UNet: input rel diff {rel_l2(xm_g, xm_o):.2e} -> logits rel diff {rel_l2(yg, yo):.2e} "
          f"(amplification {rel_l2(yg, yo) / rel_l2(xm_g, xm_o):.2f}x); oracle logits vs golden {golden_err(yo, g[f'{tag}.logits']):.2e}", flush=True)


if __name__ == "__main__":
    c2(sys.argv[2] if len(sys.argv) > 2 else "kernel")
