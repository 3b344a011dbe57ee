"""GPU: MX correction operands of the split convolutions (config.mx_conv; include/asis_hip.h: asis_gemm_desc.mx_amax_a / _b).
The rounding residuals of activations and weights as two fp8 (e4m3) bytes per element with one power-of-two scale per tensor;
the two correction terms of a split convolution (`backbones/decoders.py:109-135` conv3x3 stacks at fp32 in the reference) as
ONE block-scaled fp8 MFMA pass.  Checked: the producers' byte images against a host decode, and the convolution on all three
tile forms against fp32 torch on the UNROUNDED operands — the MX result must sit far below the single-16-bit-pass error and
close to the three-16-bit-part result."""
import pytest
import torch
import torch.nn.functional as F

from adaptersis_amd import config, ops
from adaptersis_amd.utils import weights as W
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu


def _e4m3(b: torch.Tensor) -> torch.Tensor:
    b = b.to(torch.int32)
    s, e, m = (b >> 7) & 1, (b >> 3) & 15, b & 7
    v = torch.where(e == 0, m.float() / 8.0 * 2.0 ** -6, (1.0 + m.float() / 8.0) * torch.exp2(e.float() - 7.0))
    return torch.where(s == 1, -v, v)


def _decode(mx: torch.Tensor, amax: float, dt, wside: bool):
    """MX tensor (16-bit container) -> (hi, lo) float tensors"""
    import math
    by = mx.contiguous().view(torch.uint8).view(*mx.shape, 2)
    e = math.floor(math.log2(amax))
    lo_shift = 18 if dt == torch.float16 else 15
    b_hi, b_lo = (by[..., 1], by[..., 0]) if wside else (by[..., 0], by[..., 1])
    return _e4m3(b_hi) * 2.0 ** -(7 - e), _e4m3(b_lo) * 2.0 ** -(lo_shift - e)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_mx_producers_byte_images(dev, dt):
    B, H, Wd, C = 2, 9, 11, 64
    raw = (W.tensor("mx.raw", (B, H, Wd, C), 1.0) * 3).to(dev)
    scale, shift = (0.5 + W.tensor("mx.sc", (C,), 0.2).abs()).to(dev), W.tensor("mx.sh", (C,), 0.3).to(dev)
    amax = ops.bn_relu_absmax(raw, scale, shift)
    want = torch.relu(raw * scale + shift).max()
    assert float(amax) == float(want)
    hi, mx = ops.bn_relu_upsample(raw, scale, shift, 2, dt, True, mx_amax=amax)
    hi_ref, lo_ref = ops.bn_relu_upsample(raw, scale, shift, 2, dt, True)
    assert torch.equal(hi, hi_ref)
    dh, dl = _decode(mx, float(amax), dt, False)
    assert rel_l2(dh, hi_ref.float()) < 0.04            # e4m3: 3 mantissa bits
    assert rel_l2(dl, lo_ref.float()) < 0.04
    assert bool(torch.isfinite(dh).all() and torch.isfinite(dl).all())
    # weights: (lo8, hi8)
    w = W.tensor("mx.w", (32, 64, 3, 3), 0.05).to(dev)
    wmx, wamax = ops.pack_conv_weight_mx(w, 0, dt)
    assert float(wamax) == float(w.abs().max())
    wh, wl = ops.pack_conv_weight(w, 0, dt), ops.pack_conv_weight(w, 0, dt, 1)
    dwh, dwl = _decode(wmx, float(wamax), dt, True)
    assert rel_l2(dwh, wh.float()) < 0.04 and rel_l2(dwl, wl.float()) < 0.04


@pytest.mark.parametrize("Cin,Cout,H,ks", [(128, 512, 24, 1), (128, 512, 24, 3), (256, 128, 40, 1), (128, 64, 56, 1), (192, 256, 33, 1)])
def test_mx_conv_vs_fp32(dev, Cin, Cout, H, ks):
    dt = torch.float16
    B = 4
    x = torch.relu(W.tensor(f"mxc.x{Cin}.{H}", (B, H, H, Cin), 1.0) * 2 + 0.3).to(dev)      # post-ReLU-like activations
    w = W.tensor(f"mxc.w{Cin}.{Cout}", (Cout, Cin, 3, 3), 0.03).to(dev)
    bias = W.tensor(f"mxc.b{Cout}", (Cout,), 1.0).to(dev)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1)
    x2 = x.view(-1, Cin)
    x_hi = ops.cast_pad(x2, Cin, dt).view(B, H, H, Cin)
    x_lo = ops.cast_pad(x2, Cin, dt, part=1).view(B, H, H, Cin)
    w_hi, w_lo = ops.pack_conv_weight(w, 0, dt), ops.pack_conv_weight(w, 0, dt, 1)
    # MX form of x through the producer kernel (identity BatchNorm, factor 1 is not the fused kernel: use factor-2 producer on a
    # half-resolution map would change x; so build the MX bytes with the weight packer's twin for activations)
    amax_x = ops.absmax_f32(x2)
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    assert float(ops.bn_relu_absmax(x, one, zero)) == float(amax_x)
    hi_u, mx_u = ops.bn_relu_upsample(x, one, zero, 1, dt, True, mx_amax=amax_x)          # factor 1: relu(x) = x, no resampling
    assert torch.equal(hi_u, x_hi)
    w_mx, amax_w = ops.pack_conv_weight_mx(w, 0, dt)
    e1 = rel_l2(ops.conv_gemm(x_hi, w_hi, 3, 3, 1, 1, bias_n=bias), ref)
    e3 = rel_l2(ops.conv_gemm_split(x_hi, x_lo, w_hi, w_lo, 3, 3, 1, 1, bias_n=bias, ksplit=ks), ref)
    outs = [ops.conv_gemm_split(x_hi, mx_u, w_hi, w_mx, 3, 3, 1, 1, bias_n=bias, ksplit=ks, mx=(amax_x, amax_w)) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])
    emx = rel_l2(outs[0], ref)
    print(f"conv {Cin}->{Cout} @{H}^2 ksplit {ks}: one 16-bit pass {e1:.2e}, three parts {e3:.2e}, 16-bit + MX pass {emx:.2e}")
    assert e3 < 3e-6 and emx < 0.12 * e1 and emx < 4e-5


@pytest.mark.parametrize("M,N,K", [(4321, 1536, 1536), (2100, 512, 4096), (3000, 4608, 1536)])
def test_mx_dense_gemm_vs_fp32(dev, M, N, K):
    """the linear layers of config.precise_level 2 (`dinov2/layers/block.py:89-114` at fp32 in the reference): hi x hi on the
    16-bit MFMA + one block-scaled fp8 pass over the MX planes made from the stored (hi, lo) pairs, with the fp32 residual /
    LayerScale epilogue of proj / w3 — against fp32 torch on the unrounded operands, next to one 16-bit pass and three parts."""
    dt = torch.float16
    a32 = (W.tensor(f"mxd.a{M}.{K}", (M, K), 1.0) * 2).to(dev)
    w32 = W.tensor(f"mxd.w{N}.{K}", (N, K), 0.04).to(dev)
    bias, gam = W.tensor(f"mxd.b{N}", (N,), 1.0).to(dev), (0.3 + W.tensor(f"mxd.g{N}", (N,), 0.1).abs()).to(dev)
    res = W.tensor(f"mxd.r{M}.{N}", (M, N), 2.0).to(dev)
    ref = res + gam * (a32 @ w32.t() + bias)
    a_hi, a_lo = ops.cast_pad(a32, K, dt), ops.cast_pad(a32, K, dt, part=1)
    w_hi, w_lo = ops.cast_pad(w32, K, dt), ops.cast_pad(w32, K, dt, part=1)
    a_mx, amax_a = ops.mx_from_pair(a_hi, a_lo)
    w_mx, amax_w = ops.mx_from_pair(w_hi, w_lo, wside=True)
    assert float(amax_a) == float(a_hi.float().abs().max()) and float(amax_w) == float(w_hi.float().abs().max())
    dh, dl = _decode(a_mx, float(amax_a), dt, False)
    assert rel_l2(dh, a_hi.float()) < 0.04 and rel_l2(dl, a_lo.float()) < 0.04
    dwh, dwl = _decode(w_mx, float(amax_w), dt, True)
    assert rel_l2(dwh, w_hi.float()) < 0.04 and rel_l2(dwl, w_lo.float()) < 0.04
    kw = dict(out_f32=True, bias_n=bias, scale_n=gam, res=res)
    e1 = rel_l2(ops.gemm(a_hi, w_hi, **kw) - res, ref - res)
    e3 = rel_l2(ops.gemm(a_hi, w_hi, a_lo=a_lo, b_lo=w_lo, **kw) - res, ref - res)
    outs = [ops.gemm(a_hi, w_hi, a_lo=a_mx, b_lo=w_mx, mx=(amax_a, amax_w), **kw) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])
    emx = rel_l2(outs[0] - res, ref - res)
    print(f"dense {M}x{N}x{K}: one 16-bit pass {e1:.2e}, three parts {e3:.2e}, 16-bit + MX pass {emx:.2e}")
    assert e3 < 3e-6 and emx < 0.12 * e1 and emx < 5e-5
    # 16-bit output (no residual): the plain epilogue
    o16 = ops.gemm(a_hi, w_hi, a_lo=a_mx, b_lo=w_mx, mx=(amax_a, amax_w), bias_n=bias)
    assert rel_l2(o16, a32 @ w32.t() + bias) < 6e-4


@pytest.mark.parametrize("M,Hd,K,with_bias", [(2100, 4096, 1536, True), (517, 160, 128, True), (3000, 1024, 256, False), (257, 144, 64, True)])
@pytest.mark.parametrize("form", ["mx", "plain"])
def test_swiglu_epilogue_vs_separate_kernel(dev, M, Hd, K, with_bias, form):
    """ASIS_ACT_SILU_MUL (`dinov2/layers/swiglu_ffn.py:30-34` in the w12 GEMM's epilogue, rows of w12 interleaved by
    ops.swiglu_rows): against fp32 torch, and against the unfused pair (fp32 pre-activation + asis_swiglu) — bit-identical
    where both run the same main loop (the MX split form has one), ragged M and N / 2 tiles included."""
    dt = torch.float16
    a32 = (W.tensor(f"sg.a{M}.{K}", (M, K), 1.0)).to(dev)
    w32 = W.tensor(f"sg.w{Hd}.{K}", (2 * Hd, K), 1.5 / K ** 0.5).to(dev)
    bias = W.tensor(f"sg.b{Hd}", (2 * Hd,), 0.5).to(dev) if with_bias else None
    pre = a32 @ w32.t() + (bias if with_bias else 0)
    ref = F.silu(pre[:, :Hd]) * pre[:, Hd:]
    rows = ops.swiglu_rows(Hd, dev)
    assert sorted(rows.tolist()) == list(range(2 * Hd)) and rows[:32].tolist() == list(range(16)) + list(range(Hd, Hd + 16))
    a_hi, a_lo = ops.cast_pad(a32, K, dt), ops.cast_pad(a32, K, dt, part=1)
    w_hi, w_lo = ops.cast_pad(w32, K, dt), ops.cast_pad(w32, K, dt, part=1)
    wi_hi, wi_lo = ops.cast_pad(w32[rows].contiguous(), K, dt), ops.cast_pad(w32[rows].contiguous(), K, dt, part=1)
    bi = bias[rows].contiguous() if with_bias else None
    assert ops.swiglu_fused_ok(M, Hd, K, form == "mx", form == "mx")
    if form == "mx":
        a_mx, amax_a = ops.mx_from_pair(a_hi, a_lo)
        w_mx, amax_w = ops.mx_from_pair(w_hi, w_lo, wside=True)
        wi_mx, amax_wi = ops.mx_from_pair(wi_hi, wi_lo, wside=True)
        assert torch.equal(amax_w, amax_wi) and torch.equal(wi_mx.view(torch.int16), w_mx.view(torch.int16)[rows])
        kw, kwi = dict(a_lo=a_mx, b_lo=w_mx, mx=(amax_a, amax_w)), dict(a_lo=a_mx, b_lo=wi_mx, mx=(amax_a, amax_wi))
        tol = 4e-4
    else:
        kw, kwi, tol = {}, {}, 8e-4
    out = torch.full((M + 1, Hd), 7.0, device=dev, dtype=dt)            # one guard row behind the output
    fused = ops.gemm(a_hi, wi_hi, bias_n=bi, act=ops.ACT_SILU_MUL, out=out[:M], **kwi)
    assert fused.shape == (M, Hd) and bool((out[M] == 7.0).all())
    unfused = ops.swiglu(ops.gemm(a_hi, w_hi, out_f32=True, bias_n=bias, **kw), dt)
    assert rel_l2(fused, ref) < tol and rel_l2(unfused, ref) < tol
    if form == "mx":
        assert torch.equal(fused, unfused)
    else:
        assert rel_l2(fused, unfused) < 4e-4
    assert torch.equal(fused, ops.gemm(a_hi, wi_hi, bias_n=bi, act=ops.ACT_SILU_MUL, **kwi))
    with pytest.raises(Exception, match="ACT_SILU_MUL"):                   # no fp32 / residual epilogue behind the gate
        ops.gemm(a_hi, wi_hi, bias_n=bi, act=ops.ACT_SILU_MUL, out_f32=True, **kwi)
    with pytest.raises(Exception, match="SILU_MUL"):
        ops.gemm(a_hi, wi_hi, bias_n=bi, act=ops.ACT_SILU_MUL, scale_n=torch.ones(2 * Hd, device=dev), **kwi)


def test_attention_output_in_mx_form(dev):
    """asis_attention_fwd_qkv_mx: the second output plane as the MX form of the output's lo half under the bound max |v| >= max |o|
    (a convex combination of V rows) — the hi plane is unchanged, the plane decodes to (hi, lo) of the fp32 output at e4m3
    precision, and agrees with mx_from_pair of the two-plane output at its own (tighter) maximum up to that precision."""
    dt = torch.float16
    H, D = 4, 256
    segs = [(2, 300), (1, 301)]
    R = sum(b * n for b, n in segs)
    qkv = (W.tensor("amx.qkv", (R, 3 * D), 1.0) * torch.cat([torch.full((D,), 0.4), torch.full((D,), 1.0), torch.full((D,), 3.0)])).to(dev).to(dt)
    o0, lo0 = torch.empty(R, D, device=dev, dtype=dt), torch.empty(R, D, device=dev, dtype=dt)
    ops.attention_fwd_qkv(qkv, segs, H, 0.125, o0, out_lo=lo0)
    amax_v = ops.absmax16(qkv[:, 2 * D:])
    assert float(amax_v) == float(qkv[:, 2 * D:].float().abs().max()) and float(amax_v) >= float(o0.float().abs().max())
    o1, mx1 = torch.empty_like(o0), torch.empty_like(o0)
    ops.attention_fwd_qkv(qkv, segs, H, 0.125, o1, out_lo=mx1, mx_amax=amax_v)
    assert torch.equal(o1, o0)
    dh, dl = _decode(mx1, float(amax_v), dt, False)
    assert rel_l2(dh, o0.float()) < 0.04 and rel_l2(dl, lo0.float()) < 0.06
    with pytest.raises(ValueError, match="second output plane"):
        ops.attention_fwd_qkv(qkv, segs, H, 0.125, o1, mx_amax=amax_v)


def test_layernorm_mx_planes(dev):
    """asis_layernorm_mx: the 16-bit output equals asis_layernorm's, the MX plane decodes to (hi, lo) of the fp32 LayerNorm at
    e4m3 precision under an amax BOUND several binades above the true maximum."""
    import torch.nn.functional as F
    dt = torch.float16
    R, D = 3001, 1536
    x = (W.tensor("lnmx.x", (R, D), 1.0) * 3 + W.tensor("lnmx.m", (R, 1), 1.0)).to(dev)
    w, b = (1 + 0.3 * W.tensor("lnmx.w", (D,), 1.0)).to(dev), (0.2 * W.tensor("lnmx.b", (D,), 1.0)).to(dev)
    bound = ((D - 1) ** 0.5 * w.abs().max() + b.abs().max()).reshape(1).contiguous()
    hi, mx = ops.layernorm_mx(x, w, b, 1e-6, dt, bound)
    assert torch.equal(hi, ops.layernorm(x, w, b, 1e-6, dt))
    y = F.layer_norm(x, (D,), w, b, 1e-6)
    assert float(bound) > 4 * float(y.abs().max())                       # the bound really is loose here
    dh, dl = _decode(mx, float(bound), dt, False)
    assert rel_l2(dh, hi.float()) < 0.04 and rel_l2(dl, (y - hi.float())) < 0.06


@pytest.mark.parametrize("B,Cin,Cout,H,Wd", [(2, 128, 64, 40, 40), (2, 256, 128, 33, 21), (1, 64, 64, 5, 70), (3, 128, 128, 16, 16)])
def test_halo_tile_mx_conv_vs_implicit_gemm_and_fp32(dev, B, Cin, Cout, H, Wd):
    """asis_conv3x3_halo_mx (csrc/convhalo.hip: the narrow decoder convolutions `decoders.py:109-135`, 256 -> 128 and 128 -> 64) on
    ragged tile grids: against the implicit-GEMM MX path on the same operand planes (different summation order: fp32 rounding apart),
    against fp32 torch on the unrounded operands, and its BatchNorm partial sums against the output's column sums."""
    dt = torch.float16
    x = torch.relu(W.tensor(f"halo.x{Cin}.{H}.{Wd}", (B, H, Wd, Cin), 1.0) * 2 + 0.3).to(dev)
    w = W.tensor(f"halo.w{Cin}.{Cout}", (Cout, Cin, 3, 3), 0.03).to(dev)
    bias = W.tensor(f"halo.b{Cout}", (Cout,), 1.0).to(dev)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1)
    amax_x = ops.absmax_f32(x.view(-1, Cin))
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    x_hi, x_mx = ops.bn_relu_upsample(x, one, zero, 1, dt, True, mx_amax=amax_x)
    w_hi = ops.pack_conv_weight(w, 0, dt)
    w_mx, amax_w = ops.pack_conv_weight_mx(w, 0, dt)
    assert ops.conv_halo_ok(x_hi, Cout, 1, 1, force=True)
    outs = [ops.conv3x3_halo_mx(x_hi, x_mx, w_hi, w_mx, (amax_x, amax_w), bias_n=bias, want_stats=True) for _ in range(2)]
    out, stats = outs[0]
    assert torch.equal(out, outs[1][0]) and torch.equal(stats, outs[1][1]), "not reproducible"
    e32 = rel_l2(out, ref)
    print(f"halo conv {Cin}->{Cout} @{H}x{Wd}: vs fp32 {e32:.2e}")
    assert e32 < 4e-5
    if B * H * Wd >= 256:
        gem = ops.conv_gemm_split(x_hi, x_mx, w_hi, w_mx, 3, 3, 1, 1, bias_n=bias, mx=(amax_x, amax_w))
        assert rel_l2(out, gem) < 2e-6
    flat = out.view(-1, Cout).double()
    assert rel_l2(stats[:, 0].double().sum(0), flat.sum(0)) < 1e-5 and rel_l2(stats[:, 1].double().sum(0), (flat * flat).sum(0)) < 1e-5
