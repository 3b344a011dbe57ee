"""GPU: every selectable segmentation loss (segloss mirrors over asis_seg_loss_fwd/bwd) against the goldens captured
from the imported reference (tests/golden/loss2.pt) and the validation IoU metrics from the per-class count kernel."""
import numpy as np
import pytest
import torch

from adaptersis_amd import ops
from adaptersis_amd.segloss.dice import DC, resize_softmax_dc, seg_loss
from adaptersis_amd.segloss.dice_loss import DC_and_CE_loss, SoftDiceLoss, TverskyLoss, softmax_helper
from adaptersis_amd.segloss.ND_Crossentropy import CrossentropyND
from adaptersis_amd.segloss import iou_multi
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _cases(C, tg, wts):
    """name -> f(low-res logits NCHW) using the module API; the resize to the label size is fused (seg_loss does it
    when the logits are smaller than the target), the leading softmax of the *_sm cases is the n_region/n_ce count."""
    return {
        "dc_sm": lambda o: resize_softmax_dc(o, tg, 2),
        "ce_dc": lambda o: seg_loss(o, tg, 1, ops.LOSS_DICE, 10e-20, n_ce=1),
        "softdice_sm": lambda o: SoftDiceLoss(apply_nonlin=softmax_helper)(o, tg.unsqueeze(1)),
        "dc_and_ce_sm": lambda o: seg_loss(o, tg, 1, ops.LOSS_SOFTDICE, 1.0, n_ce=2),
        "tversky_sm": lambda o: TverskyLoss(apply_nonlin=softmax_helper)(o, tg.unsqueeze(1)),
        "dc_and_ce_raw": lambda o: DC_and_CE_loss()(o, tg.unsqueeze(1)),
        "ce": lambda o: CrossentropyND()(o, tg),
        "ce_weighted": lambda o: CrossentropyND(weight=wts)(o, tg),
        "iou_sm": lambda o: seg_loss(o, tg, 2, ops.LOSS_IOU, 1e-6),
    }


@pytest.mark.parametrize("C", [2, 8])
def test_losses_vs_reference_golden(dev, C):
    g = load_golden("loss2")
    B, h, H = 3, 20, 28
    lg0 = W.tensor(f"loss2.logits{C}", (B, C, h, h), 3.0).to(dev)
    tg = W.synthetic_batch(B, H, C)[1]
    tg[0] = 0
    tg = tg.to(dev)
    for name, fn in _cases(C, tg, torch.linspace(0.1, 2.0, C).to(dev)).items():
        lg = lg0.clone().requires_grad_(True)
        loss = fn(lg)
        loss.backward()
        assert abs(float(loss) - float(g[f"loss2.c{C}.{name}"])) < 5e-6, (name, float(loss), float(g[f"loss2.c{C}.{name}"]))
        e = rel_l2(lg.grad, g[f"loss2.c{C}.{name}.grad"])
        assert e < 1e-4, (name, e)


def test_full_resolution_modules_vs_oracle(dev):
    """Module API on full-resolution NCHW logits (no resize): DC, SoftDiceLoss(), iou_loss."""
    B, C, H = 2, 5, 36
    lg0 = W.tensor("lossmod.logits", (B, C, H, H), 2.0)
    tg = W.synthetic_batch(B, H, C, seed=3)[1]
    oh = O.one_hot(tg, C)
    for fn, ofn in ((lambda o, t: DC(C)(o, t.unsqueeze(1)), lambda o: O.dc_loss(o, oh)),
                    (lambda o, t: SoftDiceLoss()(o, t.unsqueeze(1)), lambda o: O.soft_dice_loss(o, oh)),
                    (lambda o, t: iou_multi.iou_loss(o, t, num_classes=C), lambda o: O.iou_loss(o, tg, num_classes=C))):
        a = lg0.clone().to(dev).requires_grad_(True)
        la = fn(a, tg.to(dev))
        la.backward()
        b = lg0.clone().requires_grad_(True)
        lb = ofn(b)
        lb.backward()
        assert abs(float(la) - float(lb)) < 5e-6
        assert rel_l2(a.grad, b.grad) < 1e-4


def test_unbuilt_options_fail_loudly():
    with pytest.raises(NotImplementedError):
        SoftDiceLoss(batch_dice=True)
    with pytest.raises(NotImplementedError):
        TverskyLoss(do_bg=False)
    with pytest.raises(NotImplementedError):
        DC_and_CE_loss(aggregate="mean")


@pytest.mark.parametrize("C", [2, 8, 11])
def test_class_counts_and_iou_metrics(dev, C):
    """counts kernel == argmax + bincount; ch_iou / isi_iou from the counts == the oracle on the host arrays."""
    B, h, H = 2, 24, 56
    lg = W.tensor(f"cnt.logits{C}", (B, h, h, C), 2.0).to(dev)
    tg = W.synthetic_batch(B, H, C, seed=5)[1].to(dev)
    red, cnt = ops.ce_acc(lg, tg, None, counts=True)
    full = ops.resize_bilinear_fwd(lg, H, H)
    pred = full.argmax(-1)
    ref = torch.stack([torch.stack([(tg == c).sum(), (pred == c).sum(), ((tg == c) & (pred == c)).sum()]) for c in range(C)])
    assert torch.equal(cnt.long().cpu(), ref.cpu())
    assert int(red[2]) == int((pred == tg).sum())
    yt, yp = tg.cpu().numpy(), pred.cpu().numpy()
    assert abs(iou_multi.ch_iou_from_counts(cnt) - O.ch_iou(yt, yp)) < 1e-12
    assert abs(iou_multi.isi_iou_from_counts(cnt) - O.isi_iou(yt, yp)) < 1e-12
    assert abs(iou_multi.ch_iou(yt, yp) - O.ch_iou(yt, yp)) < 1e-12
    assert abs(iou_multi.isi_iou(yt, yp) - O.isi_iou(yt, yp)) < 1e-12
    z = torch.zeros_like(tg)
    _, c0 = ops.ce_acc(lg, z, None, counts=True)
    assert iou_multi.ch_iou_from_counts(c0) == O.ch_iou(z.cpu().numpy(), yp)
