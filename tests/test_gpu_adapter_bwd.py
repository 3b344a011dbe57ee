"""GPU: backward kernels and module backward of the adapters (`train_adapters` mode) against autograd of the oracle."""
import pytest
import torch
import torch.nn.functional as F

from adaptersis_amd import ops
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu
DT = torch.float16


@pytest.mark.parametrize("case", [(2, 50, 4, 16, [(6, 7), (3, 4), (2, 2)], 4), (1, 37, 8, 32, [(9, 5)], 4),
                                  (2, 20, 2, 24, [(4, 4), (3, 3)], 2)])
def test_msda_backward(dev, case):
    """d value, d offsets, d attention logits of the sampling core vs autograd through grid_sample + softmax."""
    B, Lq, M, Dh, shapes, P = case
    L, D = len(shapes), M * Dh
    Lin = sum(a * b for a, b in shapes)
    value = W.tensor(f"msb.v{case}", (B, Lin, D), 1.0).to(DT)
    off = W.tensor(f"msb.o{case}", (B, Lq, M, L, P, 2), 2.5)
    logit = W.tensor(f"msb.l{case}", (B, Lq, M, L * P), 1.0)
    ref = W.tensor(f"msb.r{case}", (Lq, 2), 0.5, 0.5)
    dout = W.tensor(f"msb.d{case}", (B, Lq, D), 1.0)
    sp = torch.tensor(shapes)
    # oracle autograd
    v_r = value.float().view(B, Lin, M, Dh).clone().requires_grad_(True)
    off_r, lg_r = off.clone().requires_grad_(True), logit.clone().requires_grad_(True)
    normalizer = torch.stack([sp[:, 1], sp[:, 0]], -1).float()
    loc = ref[None, :, None, None, None, :] + off_r / normalizer[None, None, None, :, None, :]
    aw = F.softmax(lg_r, -1).view(B, Lq, M, L, P)
    out = O.ms_deform_attn_core(v_r, sp, loc, aw)
    (out * dout).sum().backward()
    # HIP
    offaw = torch.cat([off.reshape(B * Lq, -1), logit.reshape(B * Lq, -1)], 1).contiguous().to(dev)
    starts = [0]
    for a, b in shapes[:-1]:
        starts.append(starts[-1] + a * b)
    sh = torch.tensor(shapes, dtype=torch.int32, device=dev)
    st = torch.tensor(starts, dtype=torch.int32, device=dev)
    fwd = ops.msda_fwd(value.to(dev), offaw, ref.to(dev), sh, st, B, Lq, M, L, P)
    assert rel_l2(fwd.view(B, Lq, D), out) < 2e-3
    dvalue, doffaw = ops.msda_bwd(value.to(dev), offaw, ref.to(dev), sh, st, dout.reshape(B * Lq, D).contiguous().to(dev),
                                  B, Lq, M, L, P)
    n_off = M * L * P * 2
    e = (rel_l2(dvalue.view(B, Lin, M, Dh), v_r.grad), rel_l2(doffaw[:, :n_off].reshape(off.shape), off_r.grad),
         rel_l2(doffaw[:, n_off:].reshape(logit.shape), lg_r.grad))
    print(case, "dvalue doff dlogit:", ["%.1e" % v for v in e])
    assert max(e) < 1e-4, e


def test_dwconv_gelu_backward(dev):
    B, C = 2, 64
    grids = [(7, 9), (4, 5), (2, 3)]
    Ntok = sum(a * b for a, b in grids)
    x = W.tensor("dwb.x", (B, Ntok, C), 1.0)
    w = W.tensor("dwb.w", (C, 1, 3, 3), 0.4)
    b = W.tensor("dwb.b", (C,), 0.3)
    dy = W.tensor("dwb.dy", (B, Ntok, C), 1.0)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.gelu(O.dwconv(xr, {"p.dwconv.weight": wr, "p.dwconv.bias": br}, "p", grids))
    (y * dy).sum().backward()
    starts = [0]
    for a, bb in grids[:-1]:
        starts.append(starts[-1] + a * bb)
    sh = torch.tensor(grids, dtype=torch.int32, device=dev)
    st = torch.tensor(starts, dtype=torch.int32, device=dev)
    w9 = w.reshape(C, 9).t().contiguous().to(dev)
    dx, partial = ops.dwconv_gelu_bwd(x.to(dev), w9, b.to(dev), sh, st, dy.to(dev), DT)
    red = ops.reduce_rows(partial.view(partial.shape[0], 10 * C)).view(10, C)
    assert rel_l2(dx, xr.grad) < 1e-3
    assert rel_l2(red[:9].t().reshape(C, 1, 3, 3), wr.grad) < 1e-5
    assert rel_l2(red[9], br.grad) < 1e-5
