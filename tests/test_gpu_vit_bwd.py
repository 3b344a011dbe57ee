"""GPU: backward kernels of the ViT block (attention, LayerNorm, GELU, LayerScale, linear layers) against autograd of
the oracle's fp32 restatement on the same (16-bit rounded) operands."""
import pytest
import torch

from adaptersis_amd import ops
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 2, 100), (1, 3, 257), (2, 1, 64), (1, 2, 1765)])
def test_attention_backward(dev, shape, dt):
    B, H, N = shape
    D = H * 64
    scale = 64 ** -0.5
    qkv = W.tensor(f"attnb.qkv{shape}", (B * N, 3 * D), 1.0).to(dt)
    dO = W.tensor(f"attnb.do{shape}", (B * N, D), 1.0).to(dt)
    # ---- reference: fp32 softmax attention on the rounded operands, autograd
    qf, kf, vf = [t.float().view(B, N, H, 64).transpose(1, 2).clone().requires_grad_(True)
                  for t in (qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:])]
    p = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1)
    o_ref = (p @ vf).transpose(1, 2).reshape(B * N, D)
    (o_ref * dO.float()).sum().backward()
    unhead = lambda g: g.transpose(1, 2).reshape(B * N, D)
    # ---- HIP
    g = qkv.to(dev)
    q, k, v = g[:, :D], g[:, D:2 * D], g[:, 2 * D:]
    vt = ops.transpose_tokens(v, B, N)
    lse = torch.empty((B, H, N), device=dev, dtype=torch.float32)
    o = ops.attention_fwd(q, k, vt, B, H, N, scale, lse=lse)
    assert rel_l2(o, o_ref) < (2e-3 if dt == torch.float16 else 1e-2)
    lse_ref = torch.logsumexp(qf.detach() @ kf.detach().transpose(-1, -2) * scale, -1) * 1.4426950408889634
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-3
    dOd = dO.to(dev)
    dqkv = ops.attention_bwd(q, k, v, ops.transpose_tokens(q, B, N), ops.transpose_tokens(k, B, N),
                             ops.transpose_tokens(dOd, B, N), o, dOd, lse, B, H, N, scale)
    tol = 4e-3 if dt == torch.float16 else 2.5e-2
    errs = (rel_l2(dqkv[:, :D], unhead(qf.grad)), rel_l2(dqkv[:, D:2 * D], unhead(kf.grad)), rel_l2(dqkv[:, 2 * D:], unhead(vf.grad)))
    print(shape, dt, "dq dk dv rel-L2:", ["%.2e" % e for e in errs])
    assert max(errs) < tol, errs


def test_transpose_tokens(dev):
    B, N, C = 2, 77, 128
    x = W.tensor("tt.x", (B * N, C + 64), 1.0).to(torch.float16).to(dev)
    t = ops.transpose_tokens(x[:, 64:], B, N)
    assert t.shape == (B, C, 128)
    assert torch.equal(t[:, :, :N], x[:, 64:].view(B, N, C).transpose(1, 2))
    assert float(t[:, :, N:].abs().sum()) == 0
