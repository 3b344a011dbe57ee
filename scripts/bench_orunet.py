"""OR-UNet fuse head step at the reference geometry (ViT-S/14 features, 588 x 588): python scripts/bench_orunet.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd.backbones.or_unet import ORUNetFuseEngine, UNet
from adaptersis_amd.dinov2.models import vision_transformer as vits
from adaptersis_amd.utils import weights as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
arch = "vit_small"
D, depth, heads, ffn = W.VIT_CONFIGS[arch]
model = vits.__dict__[arch](patch_size=14, img_size=518, init_values=1e-5, ffn_layer=ffn, block_chunks=0)
model.load_state_dict(W.make_vit_state_dict(arch))
u = UNet(embed_dim=D)
u.load_state_dict(W.make_or_unet_state_dict(D, 2))
eng = ORUNetFuseEngine(model.to(dev).eval(), u.to(dev).train(), lr=0.01)
img, tg = W.synthetic_batch(B, 588, 2)
img, tg = img.to(dev), tg.to(dev)
for _ in range(2):
    loss = eng.train_step(img, tg)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = eng.train_step(img, tg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"OR-UNet fuse step, {arch}, 588x588, batch {B}: {dt * 1e3:.1f} ms/step = {B / dt:.1f} img/s, loss {float(loss):.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
