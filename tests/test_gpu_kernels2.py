"""GPU parity of the adapter / CNN / loss / backward kernels against the CPU oracle and fp32 torch
autograd on the same seeded inputs (through the C ABI)."""
import pytest
import torch
import torch.nn.functional as F

from adaptersis_amd import ops
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import golden_err, load_golden, rel_l2

pytestmark = pytest.mark.gpu
DT = [torch.float16, torch.bfloat16]


def rnd(t, dt):
    return t.to(dt).float()


def i32(x, dev):
    return torch.as_tensor(x, dtype=torch.int32, device=dev).contiguous()


@pytest.mark.parametrize("dt", DT)
def test_msda_core_vs_oracle_and_golden(dev, dt):
    """Same case as tests/golden/make_golden.py:msda_core_case (locations in [-0.25, 1.25]: out-of-range taps)."""
    B, M, Dh, Lq, L, P = 2, 4, 16, 37, 3, 4
    shapes = torch.tensor([[9, 7], [5, 4], [3, 2]])
    S = int(shapes.prod(1).sum())
    starts = torch.tensor([0, 63, 83])
    value = W.tensor("msda.value", (B, S, M, Dh), 1.0)
    loc = W.tensor("msda.loc", (B, Lq, M, L, P, 2), 0.75, 0.5)
    logits = W.tensor("msda.aw", (B, Lq, M, L * P), 2.0)
    # express loc as ref + off / (W_l, H_l) with one reference point per query
    ref = W.tensor("msda.ref", (Lq, 2), 0.5, 0.5)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    off = (loc - ref[None, :, None, None, None, :]) * norm[None, None, None, :, None, :]
    offaw = torch.cat([off.reshape(B * Lq, -1), logits.reshape(B * Lq, -1)], 1).contiguous()
    out = ops.msda_fwd(value.reshape(B, S, M * Dh).to(dev).to(dt), offaw.to(dev), ref.to(dev), i32(shapes, dev),
                       i32(starts, dev), B, Lq, M, L, P)
    aw = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    refo = O.ms_deform_attn_core(rnd(value, dt), shapes, loc, aw)
    assert rel_l2(out.view(B, Lq, -1), refo) < (1e-3 if dt == torch.float16 else 6e-3)
    if dt == torch.float16:
        g = load_golden("small")
        assert golden_err(out.view(B, Lq, M * Dh), g["msda.out"]) < 1e-3


@pytest.mark.parametrize("dt", DT)
def test_dwconv_gelu(dev, dt):
    B, C = 2, 64
    grids = [(9, 9), (4, 4), (2, 2)]
    n = sum(a * b for a, b in grids)
    x = W.tensor("dw.x", (B, n, C), 2.0)
    sd = {"X.dwconv.weight": W.tensor("dw.w", (C, 1, 3, 3), 0.5), "X.dwconv.bias": W.tensor("dw.b", (C,), 0.5)}
    ref = F.gelu(O.dwconv(x, sd, "X", grids))
    w9 = sd["X.dwconv.weight"].reshape(C, 9).t().contiguous()
    starts = [0, 81, 97]
    out = ops.dwconv_gelu(x.to(dev), w9.to(dev), sd["X.dwconv.bias"].to(dev), i32(grids, dev), i32(starts, dev), dt)
    assert rel_l2(out, ref) < (5e-4 if dt == torch.float16 else 4e-3)


def test_stem_conv_and_batchnorm_pipeline(dev):
    """conv3x3_c3 -> colstats -> reduce -> bn_finalize -> apply kernels == conv + BatchNorm2d(train) + ReLU (+pool/upsample)."""
    B, H, Cout = 2, 37, 16
    img, _ = W.synthetic_batch(B, H)
    w = W.tensor("st.w", (Cout, 3, 3, 3), 0.4)
    gamma, beta = W.tensor("st.g", (Cout,), 0.3, 1.0), W.tensor("st.b", (Cout,), 0.2)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    raw_ref = F.conv2d(img, w, None, stride=2, padding=1)
    bn_ref = F.relu(F.batch_norm(raw_ref, rm, rv, gamma, beta, True, 0.1, 1e-5))
    raw = ops.conv3x3_c3(img.to(dev), w.to(dev), 2, 1)
    assert rel_l2(raw, raw_ref.permute(0, 2, 3, 1)) < 1e-6
    sums = ops.reduce_partials(ops.colstats(raw))
    drm, drv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    count = raw.numel() // Cout
    scale, shift, mean, invstd = ops.bn_finalize(sums, count, gamma.to(dev), beta.to(dev), 1e-5, 0.1, drm, drv, nbt)
    assert rel_l2(drm, rm) < 1e-5 and rel_l2(drv, rv) < 1e-5 and int(nbt) == 1
    y = ops.bn_act(raw, scale, shift, True, torch.float16)
    assert rel_l2(y, bn_ref.permute(0, 2, 3, 1)) < 5e-4
    mp = ops.bn_relu_maxpool(raw, scale, shift, torch.float16)
    assert rel_l2(mp, F.max_pool2d(bn_ref, 3, 2, 1).permute(0, 2, 3, 1)) < 5e-4
    for f in (2, 4):
        up = ops.bn_relu_upsample(raw, scale, shift, f, torch.float16)
        ref = F.interpolate(bn_ref, scale_factor=f, mode="bilinear", align_corners=True)
        assert rel_l2(up, ref.permute(0, 2, 3, 1)) < 5e-4


def test_gemm_stats_feed_batchnorm(dev):
    """BN statistics from the GEMM epilogue partials (the conv layers' path)."""
    Bn, Cin, Cout, H = 2, 16, 32, 20
    x = W.tensor("gs.x", (Bn, Cin, H, H), 1.0)
    w = W.tensor("gs.w", (Cout, Cin, 3, 3), 0.2)
    bias = W.tensor("gs.b", (Cout,), 0.5)
    x16 = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    wp = ops.pack_conv_weight(w.to(dev), 0, torch.float16)
    assert torch.equal(wp.cpu(), w.permute(0, 2, 3, 1).reshape(Cout, -1).half())
    tiles = ops.gemm_tiles_m(Bn * H * H)
    stats = torch.empty(tiles, 2, Cout, device=dev)
    raw = ops.conv_gemm(x16, wp, 3, 3, 1, 1, bias_n=bias.to(dev), stats=stats)
    ref = F.conv2d(rnd(x, torch.float16), rnd(w, torch.float16), bias, padding=1)
    sums = ops.reduce_partials(stats)
    scale, shift, mean, invstd = ops.bn_finalize(sums, Bn * H * H, None, None, 1e-5, 0.1)
    assert rel_l2(mean, ref.mean((0, 2, 3))) < 1e-5
    assert rel_l2(invstd, 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)) < 1e-5


def test_pack_dgrad_weight_and_decoder_input(dev):
    Cout, Cin = 6, 16
    w = W.tensor("pk.w", (Cout, Cin, 3, 3), 1.0)
    wd = ops.pack_conv_weight(w.to(dev), 1, torch.float16).cpu().float()
    ref = torch.zeros(Cin, 3, 3, 8)
    ref[..., :Cout] = w.flip(2, 3).permute(1, 2, 3, 0)
    assert torch.equal(wd, ref.reshape(Cin, -1).half().float())
    B, D, h, h4 = 2, 32, 6, 3
    xs, vit = W.tensor("di.x", (B, h * h, D), 1.0), W.tensor("di.v", (B, h * h, D), 1.0)
    c_all = W.tensor("di.c", (B, 50 + h4 * h4, D), 1.0)
    c4 = c_all[:, 50:]
    ref = O.assemble_decoder_input(xs, c4, vit, (h, h), (h4, h4)).permute(0, 2, 3, 1)
    out = ops.decoder_input(xs.to(dev), c_all.to(dev)[:, 50:], vit.to(dev), (h, h), (h4, h4), torch.float16)
    assert torch.equal(out.cpu(), ref.half())


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(6, 16, 3, 3), (64, 24, 3, 3), (8, 4, 1, 1), (192, 256, 3, 3), (64, 128, 3, 3), (128, 48, 3, 3)])
def test_pack_conv_weight_pair_equals_the_single_packs(dev, mode, shape):
    """the one-pass pair (hi + residual, hi + MX) is bit-identical to the three separate packs it replaces"""
    w = W.tensor(f"pkp.w{shape}", shape, 1.0).to(dev)
    for dt in (torch.float16, torch.bfloat16):
        hi, lo, amax = ops.pack_conv_weight_pair(w, mode, dt)
        assert amax is None
        assert torch.equal(hi, ops.pack_conv_weight(w, mode, dt)) and torch.equal(lo, ops.pack_conv_weight(w, mode, dt, 1))
        hi, mx, amax = ops.pack_conv_weight_pair(w, mode, dt, mx=True)
        mx_ref, amax_ref = ops.pack_conv_weight_mx(w, mode, dt)
        assert torch.equal(hi, ops.pack_conv_weight(w, mode, dt)) and torch.equal(amax, amax_ref)
        assert torch.equal(mx.view(torch.int16), mx_ref.view(torch.int16))
        hi2, mx2, amax2 = ops.pack_conv_weight_pair(w, mode, dt, mx=True, amax=amax_ref)
        assert amax2 is amax_ref and torch.equal(mx2.view(torch.int16), mx_ref.view(torch.int16))


@pytest.mark.parametrize("C,n_soft", [(2, 2), (2, 1), (11, 2)])
def test_dice_loss_fwd_bwd(dev, C, n_soft):
    """resize (h->H bilinear) + softmax (x n_soft) + DC, and its gradient wrt the decoder logits."""
    B, h, H = 3, 48, 42
    lg = W.tensor(f"dl.lg{C}", (B, C, h, h), 3.0).requires_grad_()
    tg = W.synthetic_batch(B, H, C)[1]
    out = F.interpolate(lg, size=(H, H), mode="bilinear")
    z = torch.softmax(out, 1) if n_soft == 2 else out
    loss_ref = O.dc_loss(z, O.one_hot(tg, C))
    loss_ref.backward()
    lg_nhwc = lg.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    scale = 1024.0
    loss, coef, sums = ops.dice_fwd(lg_nhwc, tg.to(dev), n_soft, 1e-19, scale)
    assert abs(float(loss) - float(loss_ref)) < 2e-6
    dz = ops.dice_bwd(lg_nhwc, tg.to(dev), coef, n_soft)
    d16, partial = ops.resize_bilinear_bwd(dz, h, h, torch.float16)
    gref = lg.grad.permute(0, 2, 3, 1) * scale
    assert rel_l2(d16[..., :C], gref) < 1e-3
    assert torch.all(d16[..., C:] == 0)
    assert rel_l2(ops.reduce_rows(partial), gref.sum((0, 1, 2))) < 1e-4


def test_dice_all_background_eps_path(dev):
    """An all-background image exercises the 1e-19 epsilon of segloss/dice.py:28."""
    B, C, H = 2, 2, 28
    lg = W.tensor("eps.lg", (B, C, H, H), 3.0)
    tg = torch.zeros(B, H, H, dtype=torch.long)
    ref = O.dc_loss(torch.softmax(lg, 1), O.one_hot(tg, C))
    loss, _, _ = ops.dice_fwd(lg.permute(0, 2, 3, 1).contiguous().to(dev), tg.to(dev), 2)
    assert abs(float(loss) - float(ref)) < 2e-6


@pytest.mark.parametrize("factor", [2, 4])
def test_bn_relu_upsample_backward(dev, factor):
    B, C, H = 2, 16, 9
    x = W.tensor("ub.x", (B, C, H, H), 1.0, 0.2).requires_grad_()
    gamma = W.tensor("ub.g", (C,), 0.3, 1.0).requires_grad_()
    beta = W.tensor("ub.b", (C,), 0.3).requires_grad_()
    y = F.interpolate(F.relu(F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)), scale_factor=factor,
                      mode="bilinear", align_corners=True)
    dU = W.tensor("ub.du", tuple(y.shape), 1.0)
    y.backward(dU)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    sums = ops.reduce_partials(ops.colstats(xd))
    n = B * H * H
    scale, shift, mean, invstd = ops.bn_finalize(sums, n, gamma.detach().to(dev), beta.detach().to(dev), 1e-5, 0.1)
    g, partial = ops.upsample_bn_relu_bwd(dU.permute(0, 2, 3, 1).contiguous().to(dev), xd, scale, shift, mean, invstd,
                                          factor)
    red = ops.reduce_rows(partial.view(partial.shape[0], -1)).view(2, C)
    assert rel_l2(red[0], beta.grad) < 1e-5
    assert rel_l2(red[1], gamma.grad) < 1e-5
    dx, part2 = ops.bn_bwd_apply(g, xd, mean, invstd, gamma.detach().to(dev), red[1].contiguous(), red[0].contiguous(),
                                 n, torch.float16)
    assert rel_l2(dx, x.grad.permute(0, 2, 3, 1)) < 1e-3


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Cin,Cout,H,k,pad", [(16, 24, 11, 3, 1), (64, 128, 20, 3, 1), (8, 2, 33, 3, 1), (32, 40, 10, 1, 0),
                                              (96, 32, 6, 3, 1), (64, 2, 21, 3, 1), (16, 3, 13, 3, 1), (32, 4, 9, 3, 1),
                                              (64, 64, 12, 3, 1), (128, 256, 20, 3, 1), (256, 512, 19, 3, 1),
                                              (128, 256, 37, 3, 1),
                                              # the halo-tile weight-gradient kernel (csrc/convwgrad.hip: Cout % 64 == 0, Cin % 128 == 0):
                                              # ragged tiles both ways, one / several channel-block combinations
                                              (128, 64, 20, 3, 1), (128, 64, 37, 3, 1), (256, 128, 24, 3, 1), (128, 192, 17, 3, 1)])
def test_conv_wgrad_and_dgrad(dev, dt, Cin, Cout, H, k, pad):
    """wgrad (transposed-LDS-read GEMM) and dgrad (implicit GEMM with flipped weights) vs autograd."""
    Bn = 3
    x = W.tensor(f"wg.x{Cin}", (Bn, Cin, H, H), 1.0)
    w = W.tensor(f"wg.w{Cin}", (Cout, Cin, k, k), 0.3)
    xr, wr = rnd(x, dt).requires_grad_(), rnd(w, dt).requires_grad_()
    y = F.conv2d(xr, wr, None, padding=pad)
    dy = rnd(W.tensor(f"wg.dy{Cin}", tuple(y.shape), 1.0), dt)
    y.backward(dy)
    CoP = (Cout + 7) // 8 * 8
    dy16 = torch.zeros(Bn, H, H, CoP, dtype=dt)
    dy16[..., :Cout] = dy.permute(0, 2, 3, 1).to(dt)
    x16 = x.permute(0, 2, 3, 1).contiguous().to(dt).to(dev)
    dW = ops.wgrad(dy16.to(dev), x16, Cout, k, k, 1, pad, 0.5)
    assert dW.shape == w.shape
    assert rel_l2(dW, 0.5 * wr.grad) < 5e-6
    wdg = ops.pack_conv_weight(w.to(dev), 1, dt)
    dx = ops.conv_gemm(dy16.to(dev), wdg, k, k, 1, pad)
    assert rel_l2(dx, xr.grad.permute(0, 2, 3, 1)) < 5e-6


@pytest.mark.parametrize("split", [False, True])
def test_conv_ksplit_matches_single_pass(dev, split):
    """3x3 conv run as three side-by-side K parts (`ksplit=3`: one kernel row each, partial maps summed in a fixed order)
    == the one-pass launch and the fp32 convolution; the bias is added exactly once."""
    Bn, Cin, Cout, H = 2, 64, 160, 23
    x = W.tensor("ks.x", (Bn, Cin, H, H), 1.0)
    w = W.tensor("ks.w", (Cout, Cin, 3, 3), 0.1)
    bias = W.tensor("ks.b", (Cout,), 0.5)
    ref = F.conv2d(x, w, bias, padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous().view(-1, Cin).to(dev)
    dt = torch.float16
    x_hi = ops.cast_pad(xn, Cin, dt).view(Bn, H, H, Cin)
    w_hi = ops.pack_conv_weight(w.to(dev), 0, dt)
    if split:
        x_lo = ops.cast_pad(xn, Cin, dt, part=1).view(Bn, H, H, Cin)
        w_lo = ops.pack_conv_weight(w.to(dev), 0, dt, 1)
        one = ops.conv_gemm_split(x_hi, x_lo, w_hi, w_lo, 3, 3, 1, 1, bias_n=bias.to(dev))
        three = ops.conv_gemm_split(x_hi, x_lo, w_hi, w_lo, 3, 3, 1, 1, bias_n=bias.to(dev), ksplit=3)
        assert rel_l2(three, ref) < 2e-6
    else:
        one = ops.conv_gemm(x_hi, w_hi, 3, 3, 1, 1, bias_n=bias.to(dev))
        three = ops.conv_gemm(x_hi, w_hi, 3, 3, 1, 1, bias_n=bias.to(dev), ksplit=3)
        assert rel_l2(three, ref) < 1e-3
    assert three.shape == one.shape
    assert rel_l2(three, one) < 1e-6


@pytest.mark.parametrize("n,off", [(1000, 0), (1003, 0), (1 << 20, 0), ((1 << 21) + 7, 0), (4099, 1), (3, 0)])
def test_sgd_momentum_matches_torch(dev, n, off):
    """16-byte kernel (aligned buckets, any tail) and the scalar one (``off``: a view one element into the allocation)"""
    p0, g1, g2 = W.tensor("sg.p", (n,), 1.0), W.tensor("sg.g1", (n,), 1.0), W.tensor("sg.g2", (n,), 1.0)
    pt = p0.clone().requires_grad_()
    opt = torch.optim.SGD([pt], lr=0.01, momentum=0.99, weight_decay=3e-5)
    p, buf = torch.empty(n + off, device=dev)[off:].copy_(p0), torch.full((n + off,), float("nan"), device=dev)[off:]
    guard = torch.zeros(2, device=dev, dtype=torch.int32)
    for i, g in enumerate((g1, g2)):
        pt.grad = g.clone()
        opt.step()
        gd = torch.empty(n + off, device=dev)[off:].copy_(g * 64.0)
        ops.grad_guard(gd, guard, True)
        ops.sgd_momentum(p, gd, buf, 0.01, 0.99, 3e-5, 1.0 / 64.0, i == 0, guard if i else None, count_skip=True)   # both entry points
    assert rel_l2(p, pt.detach()) < 1e-6 and guard.tolist() == [0, 0]
    # the guard sees a single inf / NaN wherever it sits (vector body, remainder loop, the n % 4 tail)
    for pos in sorted({0, n // 3, n - 1, max(n - 5, 0)}):
        for bad in (float("inf"), float("-inf"), float("nan")):
            gd = torch.empty(n + off, device=dev)[off:].copy_(g1)
            gd[pos] = bad
            ops.grad_guard(gd, guard, True)
            assert int(guard[0]) == 1, (pos, bad)
    gd = torch.empty(n + off, device=dev)[off:].fill_(3e38)       # large but finite: not an overflow
    ops.grad_guard(gd, guard, True)
    assert int(guard[0]) == 0


def test_split_precision_conv_matches_fp32(dev):
    """hi/lo split operands (3 MFMA passes) reproduce the UNROUNDED fp32 convolution to ~1e-6."""
    Bn, Cin, Cout, H = 2, 32, 48, 19
    x = W.tensor("sp.x", (Bn, Cin, H, H), 1.0)
    w = W.tensor("sp.w", (Cout, Cin, 3, 3), 0.2)
    bias = W.tensor("sp.b", (Cout,), 0.5)
    ref = F.conv2d(x, w, bias, padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous().view(-1, Cin).to(dev)
    x_hi = ops.cast_pad(xn, Cin, torch.float16).view(Bn, H, H, Cin)
    x_lo = ops.cast_pad(xn, Cin, torch.float16, part=1).view(Bn, H, H, Cin)
    w_hi = ops.pack_conv_weight(w.to(dev), 0, torch.float16)
    w_lo = ops.pack_conv_weight(w.to(dev), 0, torch.float16, 1)
    stats = torch.empty(ops.gemm_tiles_m(Bn * H * H), 2, Cout, device=dev)
    y = ops.conv_gemm_split(x_hi, x_lo, w_hi, w_lo, 3, 3, 1, 1, bias_n=bias.to(dev), stats=stats)
    assert rel_l2(y, ref) < 3e-6
    assert rel_l2(stats.sum(0)[0], ref.sum((0, 1, 2))) < 1e-4
    y1 = ops.conv_gemm(x_hi, w_hi, 3, 3, 1, 1, bias_n=bias.to(dev))
    assert rel_l2(y1, ref) > 1e-4  # single pass is limited by the 16-bit rounding of the operands


@pytest.mark.parametrize("dt16,tol", [(torch.float16, 2e-6), (torch.bfloat16, 6e-5)])
@pytest.mark.parametrize("Cout", [2, 3, 8, 11])
def test_smallcout_direct_conv_fwd_and_dgrad(dev, Cout, dt16, tol):
    """Classifier conv (few output channels; MFMA tile kernels at Cin = 64, fp32 vector kernels otherwise) vs autograd,
    split input halves; bf16 halves carry 16 mantissa bits together, hence the wider tolerance."""
    Bn, Cin, H, Wd = 2, 64, 23, 19
    x = W.tensor("sc.x", (Bn, Cin, H, Wd), 1.0).requires_grad_()
    w = W.tensor(f"sc.w{Cout}", (Cout, Cin, 3, 3), 0.3)
    bias = W.tensor(f"sc.b{Cout}", (Cout,), 0.5)
    y = F.conv2d(x, w, bias, padding=1)
    dy = W.tensor(f"sc.dy{Cout}", tuple(y.shape), 1.0)
    y.backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().view(-1, Cin).to(dev)
    x_hi = ops.cast_pad(xn, Cin, dt16).view(Bn, H, Wd, Cin)
    x_lo = ops.cast_pad(xn, Cin, dt16, part=1).view(Bn, H, Wd, Cin)
    out = ops.conv3x3_smallcout_fwd(x_hi, x_lo, w.to(dev), bias.to(dev))
    assert rel_l2(out, y.detach().permute(0, 2, 3, 1)) < tol
    out1 = ops.conv3x3_smallcout_fwd(x_hi, None, w.to(dev), None)
    assert rel_l2(out1, F.conv2d(rnd(x.detach(), dt16), w, None, padding=1).permute(0, 2, 3, 1)) < tol
    if Cout <= 8:
        dn = dy.permute(0, 2, 3, 1).contiguous().view(-1, Cout).to(dev)
        d_hi = ops.cast_pad(dn, 8, dt16).view(Bn, H, Wd, 8)
        d_lo = ops.cast_pad(dn, 8, dt16, part=1).view(Bn, H, Wd, 8)
        dx = ops.conv3x3_smallcout_dgrad(d_hi, d_lo, w.to(dev))
        assert rel_l2(dx, x.grad.permute(0, 2, 3, 1)) < tol


@pytest.mark.parametrize("Cin,Cout,H,stride,pad", [(64, 48, 19, 1, 1), (128, 128, 24, 1, 1), (64, 64, 37, 2, 0)])
def test_fused_split_conv_single_launch(dev, Cin, Cout, H, stride, pad):
    """Cin % 64 == 0 takes the one-launch form (virtual 3K reduction on the large-tile kernel), incl. BN stats."""
    Bn = 2
    x = W.tensor(f"fs.x{Cin}", (Bn, Cin, H, H), 1.0)
    w = W.tensor(f"fs.w{Cin}", (Cout, Cin, 3, 3), 0.2)
    bias = W.tensor(f"fs.b{Cin}", (Cout,), 0.5)
    ref = F.conv2d(x, w, bias, stride=stride, padding=pad).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous().view(-1, Cin).to(dev)
    x_hi = ops.cast_pad(xn, Cin, torch.float16).view(Bn, H, H, Cin)
    x_lo = ops.cast_pad(xn, Cin, torch.float16, part=1).view(Bn, H, H, Cin)
    w_hi = ops.pack_conv_weight(w.to(dev), 0, torch.float16)
    w_lo = ops.pack_conv_weight(w.to(dev), 0, torch.float16, 1)
    M = ref.shape[0] * ref.shape[1] * ref.shape[2]
    stats = torch.full((ops.gemm_tiles_m(M), 2, Cout), float("nan"), device=dev)
    y = ops.conv_gemm_split(x_hi, x_lo, w_hi, w_lo, 3, 3, stride, pad, bias_n=bias.to(dev), stats=stats)
    assert rel_l2(y, ref) < 3e-6
    assert rel_l2(stats.sum(0)[0], ref.sum((0, 1, 2))) < 1e-4
    assert rel_l2(stats.sum(0)[1], (ref * ref).sum((0, 1, 2))) < 1e-5


def test_fused_split_dense_gemm(dev):
    M, N, K = 1300, 384, 256
    a = W.tensor("fd.a", (2, M, K), 1.0).to(dev)
    b = W.tensor("fd.b", (N, K), 0.2).to(dev)
    bias = W.tensor("fd.bias", (N,), 0.5).to(dev)
    a_hi = ops.cast_pad(a.view(-1, K), K, torch.float16).view(2, M, K)
    a_lo = ops.cast_pad(a.view(-1, K), K, torch.float16, part=1).view(2, M, K)
    b_hi, b_lo = ops.cast_pad(b, K, torch.float16), ops.cast_pad(b, K, torch.float16, part=1)
    out = torch.empty(2, M, N + 16, device=dev)[:, :, :N]
    ops.gemm_split(a_hi, a_lo, b_hi, b_lo, out=out, bias_n=bias)
    ref = a @ b.t() + bias
    assert rel_l2(out, ref) < 3e-6


@pytest.mark.parametrize("B,H,Wd,classes", [(2, 20, 24, 2), (1, 7, 13, 3), (3, 4, 8, 2)])
def test_classifier_conv_with_upsample_on_load(dev, B, H, Wd, classes):
    """asis_conv3x3_smallcout_fwd_up / _wgrad_up (`decoders.py:131-135`: BatchNorm, ReLU, Upsample(2, bilinear, align_corners), Conv2d(64,
    classes, 3, padding=1)): the classifier conv and its weight gradient with the upsampled operand evaluated on load, against
    asis_bn_relu_upsample followed by the plain kernels (same arithmetic: equal to fp32 rounding of the conv sums) and fp32 torch."""
    from adaptersis_amd import config
    dt = config.operand_dtype
    C = 64
    raw = (W.tensor(f"cu.raw{H}.{Wd}", (B, H, Wd, C), 1.0) * 2).to(dev)
    scale, shift = (0.5 + W.tensor("cu.sc", (C,), 0.2).abs()).to(dev), W.tensor("cu.sh", (C,), 0.3).to(dev)
    w = W.tensor(f"cu.w{classes}", (classes, C, 3, 3), 0.05).to(dev)
    bias = W.tensor(f"cu.b{classes}", (classes,), 1.0).to(dev)
    assert ops.cls_up_ok(raw, classes)
    hi, lo = ops.bn_relu_upsample(raw, scale, shift, 2, dt, True)
    ref = ops.conv3x3_smallcout_fwd(hi, lo, w, bias)
    out = ops.conv3x3_smallcout_fwd_up(raw, scale, shift, w, bias, dt)
    assert out.shape == ref.shape and rel_l2(out, ref) < 1e-6
    up = F.interpolate(torch.relu(raw * scale + shift).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    assert rel_l2(out, F.conv2d(up, w, bias, padding=1).permute(0, 2, 3, 1)) < 2e-5
    dy32 = W.tensor(f"cu.dy{H}.{Wd}", (B, 2 * H, 2 * Wd, 8), 1.0).to(dev)
    dy32[..., classes:] = 0
    dy = dy32.to(dt).contiguous()
    g_ref = ops.wgrad(dy, hi, classes, 3, 3, 1, 1, 1.0)
    g = ops.conv3x3_smallcout_wgrad_up(dy, raw, scale, shift, classes, 1.0)
    # the on-load values and asis_bn_relu_upsample's may round to neighbouring 16-bit values here and there (FMA contraction differs
    # between the two kernels): 1.7e-5 measured; both sit at the 16-bit operand's distance from the fp32 gradient
    assert rel_l2(g, g_ref) < 5e-5
    wr = w.clone().requires_grad_(True)
    F.conv2d(up, wr, None, padding=1).backward(dy.float()[..., :classes].permute(0, 3, 1, 2))
    assert rel_l2(g, wr.grad) < 1e-3 and rel_l2(g_ref, wr.grad) < 1e-3
