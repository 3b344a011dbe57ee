import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from adaptersis_amd import ops, config
from adaptersis_amd.backbones.decoders import conv_bn_relu_up_forward, conv_bn_relu_up_backward
from adaptersis_amd.dinov2.layers.blocks import _Packed
from adaptersis_amd.utils import weights as W
dev = torch.device("cuda:0")
torch.backends.cudnn.allow_tf32 = False
for factor in (1, 2):
  for (B, H, Cin, Cout) in ((2, 40, 192, 96), (2, 12, 64, 32)):
    owner = _Packed().to(dev)
    conv = nn.Conv2d(Cin, Cout, 3, padding=1, bias=False).to(dev)
    bn = nn.BatchNorm2d(Cout).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W.tensor("dbg.w", tuple(conv.weight.shape), (2.0 / (Cin * 9)) ** 0.5).to(dev))
        bn.weight.copy_(1.0 + W.tensor("dbg.g", (Cout,), 0.2).to(dev)); bn.bias.copy_(W.tensor("dbg.b", (Cout,), 0.2).to(dev))
    x = F.relu(W.tensor(f"dbg.x{H}", (B, Cin, H, H), 1.0)).to(dev)
    R = W.tensor(f"dbg.r{H}{factor}", (B, Cout, H * factor, H * factor), 1.0).to(dev) + 0.7
    xr = x.clone().requires_grad_(True)
    raw = conv(xr); raw.retain_grad()
    y = F.relu(F.batch_norm(raw, None, None, bn.weight, bn.bias, True, 0.1, 1e-5))
    if factor > 1: y = F.interpolate(y, scale_factor=factor, mode="bilinear", align_corners=True)
    (y * R).sum().backward()
    x2 = x.permute(0, 2, 3, 1).contiguous().view(-1, Cin)
    xh = ops.cast_pad(x2, Cin, torch.float16).view(B, H, H, Cin); xl = ops.cast_pad(x2, Cin, torch.float16, part=1).view(B, H, H, Cin)
    up, st = conv_bn_relu_up_forward(owner, "s", xh, xl, conv, bn, factor, False, True, True)
    yy = (up[0].float() + up[1].float()).permute(0, 3, 1, 2)
    rl = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    S = 65536.0 / (B * H * H)
    dU = (R * S).permute(0, 2, 3, 1).contiguous()
    grads = {"p.0.weight": torch.empty_like(conv.weight), "p.1.weight": torch.empty_like(bn.weight), "p.1.bias": torch.empty_like(bn.bias)}
    # replicate internals to get dx16
    g, partial = ops.upsample_bn_relu_bwd(dU, st.raw, st.scale, st.shift, st.mean, st.invstd, st.factor)
    red = ops.reduce_rows(partial.view(partial.shape[0], 2 * Cout))
    r = ops.bn_bwd_apply(g, st.raw, st.mean, st.invstd, bn.weight.detach().float().contiguous(), red[Cout:], red[:Cout], st.count, torch.float16, True)
    dx = (r[0].float() + r[1].float()).permute(0, 3, 1, 2) / S
    # torch formula on the same device inputs
    n = float(B * H * H)
    xhat = (st.raw - st.mean) * st.invstd
    dbeta, dgamma = red[:Cout], red[Cout:]
    ref_dx = bn.weight.detach() * st.invstd * (g - dbeta / n - xhat * dgamma / n)
    print("   kernel vs formula", rl(r[0].float() + r[1].float(), ref_dx), " dbeta", rl(dbeta / S, bn.bias.grad), "dgamma", rl(dgamma / S, bn.weight.grad),
          "mean", rl(st.mean, raw.detach().mean((0, 2, 3))), "invstd", rl(st.invstd, 1 / torch.sqrt(raw.detach().var((0, 2, 3), unbiased=False) + 1e-5)), "count", st.count)
    dX = conv_bn_relu_up_backward(owner, "s", st, dU, conv, bn, 1.0 / S, grads, "p", True, False)
    print(f"factor {factor} H {H}: fwd {rl(yy, y):.2e}  dy(hi+lo) {rl(dx, raw.grad):.2e} dy(hi) {rl(r[0].float().permute(0,3,1,2)/S, raw.grad):.2e} "
          f"dW {rl(grads['p.0.weight'], conv.weight.grad):.2e} dgamma {rl(grads['p.1.weight'], bn.weight.grad):.2e} dX {rl(dX.permute(0,3,1,2)/S, xr.grad):.2e}"
          f"  max|dy16| {float(r[0].abs().max()):.1f}")
