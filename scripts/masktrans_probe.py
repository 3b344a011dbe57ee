"""Lab: where the MaskTransformer head's error against the fp32 oracle comes from (per intermediate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from adaptersis_amd import ops, config
from adaptersis_amd.backbones.masktrans_block import MaskTransformer
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O

def rel(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
dev = torch.device("cuda:0")
for tag, (n_cls, De, D, heads, GS, B, mode) in dict(mt2=(2, 384, 256, 4, 16, 2, "init"), mt2k=(2, 384, 256, 4, 16, 2, "kernel"),
                                                    mt5=(5, 64, 128, 2, 9, 3, "kernel")).items():
    sd = W.make_masktrans_state_dict(De, D, 2, n_cls, mode=mode)
    m = MaskTransformer(n_cls, 14, De, 2, heads, D, 4 * D, 0.0, 0.0).to(dev); m.load_state_dict(sd); m.train()
    tok = W.tensor(f"{tag}.tok", (B, GS * GS, De), 1.0)
    with torch.no_grad():
        logits, sv = m._forward_core(tok.to(dev), save=True)
        x = F.linear(tok, sd["proj_dec.weight"], sd["proj_dec.bias"]); x = torch.cat((x, sd["cls_emb"].expand(B, -1, -1)), 1)
        N = GS * GS
        x_in = x
        for i in range(2): x = O.masktrans_block(x, sd, f"blocks.{i}", heads)
        x0 = sv["saves"][0][0]
        print(tag, "x before blocks", rel(x0.view(B, N + n_cls, D), x_in), "std", float(x_in.std()))
        print(tag, "x after blocks", rel(sv["x"].view(B, N + n_cls, D), x))
        xd = O.layer_norm(x, sd, "decoder_norm", 1e-5)
        print(tag, "xd16", rel(sv["xd16"].float().view(B, N + n_cls, D), xd))
        P = xd[:, :N] @ sd["proj_patch"]; Cc = xd[:, N:] @ sd["proj_classes"]
        print(tag, "Pall patches", rel(sv["Pall"].view(B, N + n_cls, D)[:, :N], P))
        ch = Cc / Cc.norm(dim=-1, keepdim=True)
        print(tag, "chat", rel(sv["chat"], ch))
        cos = (P / P.norm(dim=-1, keepdim=True)) @ ch.transpose(1, 2)
        print(tag, "cos", rel(sv["cosm"].view(B, N, n_cls), cos), "abs max err", float((sv["cosm"].view(B, N, n_cls).cpu() - cos).abs().max()), "cos std", float(cos.std()))
        lg = O.layer_norm(cos, sd, "mask_norm", 1e-5)
        print(tag, "logits", rel(logits.view(B, N, n_cls), lg))
        # conditioning: the oracle's own masks under a 1e-4 relative perturbation of the cosines
        lg2 = O.layer_norm(cos * (1 + 1e-4 * torch.randn_like(cos)), sd, "mask_norm", 1e-5)
        print(tag, "oracle logits under 1e-4 relative noise on the cosines:", rel(lg2, lg))
