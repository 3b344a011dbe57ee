"""Sanity of the 16-bit backward at full scale: one config-4 step on ViT-L/14 588^2, B=2: any inf / nan in the gradients,
their magnitudes per block, and the largest 16-bit gradient operand (loss scale 2^16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from adaptersis_amd import config, ops
dev = torch.device("cuda:0")
eng = bench.build_engine_cfg(4, "vit_large", dev, lr=0.01)
img, tgt = bench.synthetic(2, 588, 0, dev)
peak = [0.0]
orig = ops.attention_bwd
def wrap(*a, **k):
    r = orig(*a, **k)
    peak[0] = max(peak[0], float(r.float().abs().max()))
    return r
ops.attention_bwd = wrap
loss = eng.train_step(img, tgt)
g = eng.vit_bucket.grad
print("loss", float(loss), "finite", bool(torch.isfinite(g).all()), "max|g|", float(g.abs().max()), "rms", float(g.pow(2).mean().sqrt()))
print("max |dqkv| (16-bit, scaled by", config.loss_scale, "):", peak[0])
for n in ("blocks.23.mlp.fc2.weight", "blocks.12.attn.qkv.weight", "blocks.0.attn.qkv.weight", "patch_embed.proj.weight", "pos_embed"):
    v = eng.vit_bucket.views[n]
    print(f"{n:28s} rms {float(v.pow(2).mean().sqrt()):.3e} max {float(v.abs().max()):.3e} finite {bool(torch.isfinite(v).all())}")
print("decoder finite", bool(torch.isfinite(eng.bucket.grad).all()))
