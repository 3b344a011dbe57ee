import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from adaptersis_amd import ops, config
from adaptersis_amd.backbones.unet_parts import UNet
from adaptersis_amd.segloss.dice import seg_loss
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
dev = torch.device("cuda:0")
B, hw, HW = 2, 10, 56
usd = W.make_unet_state_dict(384, 2)
x = W.tensor("unet.step.x", (B, 384, hw, hw), 1.0)
tg = W.synthetic_batch(B, HW, 2)[1]; oh = O.one_hot(tg, 2)
# --- oracle on CPU with captured conv-output grads
osd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in usd.items()}
rec = []
last = []
orig = F.conv2d
def conv2d(inp, w, b=None, **kw):
    out = orig(inp, w, b, **kw)
    if w.shape[-1] == 3:
        out.retain_grad(); inp.retain_grad() if inp.requires_grad else None; rec.append((inp, w, out))
    else:
        inp.retain_grad(); out.retain_grad(); last.append((inp, out))
    return out
F.conv2d = conv2d
oy = O.unet(x, osd)
F.conv2d = orig
oo = F.interpolate(oy, size=(HW, HW), mode="bilinear")
(O.cross_entropy_nd(oo, tg) + O.dc_loss(oo, oh)).backward()
# --- HIP with recorded bn_bwd_apply outputs and upsample_bn_relu_bwd inputs
u = UNet(384, 2).to(dev); u.load_state_dict(usd); u.train()
calls = []
o1, o2 = ops.bn_bwd_apply, ops.upsample_bn_relu_bwd
def rec_apply(g, xr, *a, **k):
    r = o1(g, xr, *a, **k); calls.append(("dx", r, g)); return r
def rec_up(dU, *a, **k):
    calls.append(("dU", dU.clone())); return o2(dU, *a, **k)
import adaptersis_amd.backbones.decoders as D
ops.bn_bwd_apply = rec_apply; ops.upsample_bn_relu_bwd = rec_up
y = u(x.to(dev))
y.retain_grad()
loss = seg_loss(y, tg.to(dev), 1, ops.LOSS_DICE, 10e-20, n_ce=1)
loss.backward()
S = config.loss_scale
rl = lambda a, b: float((a.float().cpu() - b.float()).norm() / b.float().norm())
# order of HIP backward stages: up4.b, up4.a, up3.b, up3.a, ... ; oracle rec order is forward: d3a d3b d4a d4b u1a u1b u2a u2b u3a u3b u4a u4b
names = ["d3a","d3b","d4a","d4b","u1a","u1b","u2a","u2b","u3a","u3b","u4a","u4b"]
orc = {n: r for n, r in zip(names, rec)}
hip_order = ["u4b","u4a","u3b","u3a","u2b","u2a","u1b","u1a","d4b","d4a","d3b","d3a"]
dxs = [c for c in calls if c[0] == "dx"]; dUs = [c for c in calls if c[0] == "dU"]
for n, cdx, cdu in zip(hip_order, dxs, dUs):
    inp, w, out = orc[n]
    r = cdx[1]
    hi = r[0].float().permute(0, 3, 1, 2) / S
    full = (r[0].float() + r[1].float()).permute(0, 3, 1, 2) / S if len(r) == 3 else hi
    print(n, "dy(hi+lo)", f"{rl(full, out.grad):.2e}", "dy(hi)", f"{rl(hi, out.grad):.2e}", " max|dy16|", f"{float(r[0].abs().max()):.3g}",
          "rms", f"{float(r[0].float().pow(2).mean().sqrt()):.3g}")

print("dlogits", rl(y.grad, last[0][1].grad), "dU(outc dgrad)", rl(dUs[0][1].permute(0,3,1,2)/S, last[0][0].grad))
for n, cdu in zip(hip_order, dUs):
    pass
# g of last stage vs oracle: relu mask * dU

n, cdx, cdu = hip_order[0], dxs[0], dUs[0]
inp, w, out = orc[n]
full = ((cdx[1][0].float() + cdx[1][1].float()).permute(0, 3, 1, 2) / S).cpu()
ref = out.grad
alpha = float((full * ref).sum() / (ref * ref).sum())
print("best-fit scale", alpha, "residual after scaling", rl(full / alpha, ref))
err_c = ((full - ref).pow(2).sum((0, 2, 3)).sqrt() / ref.pow(2).sum((0, 2, 3)).sqrt())
print("per-channel err: min %.2e median %.2e max %.2e" % (float(err_c.min()), float(err_c.median()), float(err_c.max())))
d = (full - ref)
print("err mean per channel / ref rms:", float(d.mean((0,2,3)).abs().mean() / ref.pow(2).mean().sqrt()))
# g check: oracle g = grad wrt BN output * relu mask: reconstruct from oracle: grad wrt stage output = dUs ref?
g_hip = cdx[2].permute(0, 3, 1, 2).cpu() / S
# oracle g: need grad wrt bn output: recompute: ref dy -> can't invert; instead compare sums
print("sum g hip per-ch vs oracle bias grad:", rl(g_hip.sum((0, 2, 3)), osd["up4.conv.double_conv.4.bias"].grad))

c = int(err_c.argmax())
dch = d[:, c]; rch = ref[:, c]
print("worst channel", c, "err", float(err_c[c]), "ref rms", float(rch.pow(2).mean().sqrt()), "n big diffs", int((dch.abs() > 1e-2 * rch.pow(2).mean().sqrt()).sum()), "of", dch.numel())
print("diff mean", float(dch.mean()), "diff std", float(dch.std()), "corr(diff, ref)", float((dch * rch).sum() / (dch.norm() * rch.norm())))
bnw = osd["up4.conv.double_conv.4.weight"].detach(); bnb = osd["up4.conv.double_conv.4.bias"].detach()
raw_o = out.detach()[:, c]
mu, var = raw_o.mean(), raw_o.var(unbiased=False)
act = ((raw_o - mu) / (var + 1e-5).sqrt() * bnw[c] + bnb[c])
print("gamma", float(bnw[c]), "beta", float(bnb[c]), "frac active", float((act > 0).float().mean()), "min |act|", float(act.abs().min()), "var", float(var))
top = err_c.topk(6)
print("top err channels", top.indices.tolist(), [f"{v:.1e}" for v in top.values.tolist()], "gammas", [f"{float(bnw[i]):.3f}" for i in top.indices], "var", [f"{float(out.detach()[:, i].var()):.2e}" for i in top.indices])
