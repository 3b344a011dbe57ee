"""Deterministic, RNG-stream-independent synthetic weights.

There is no network on the build or GPU boxes, so neither DINOv2 checkpoints nor
the reference's trained adapters exist.  Every benchmark, parity test and golden
fixture therefore uses weights produced here: a counter-based generator
(Philox raw 32-bit words -> uniform) keyed by the *parameter name*, so a tensor
has the same value no matter in which order, on which machine or with which
torch RNG state it is produced.  The same ``state_dict`` is loaded into the
imported reference (this container only, see ``tests/golden/make_golden.py``),
into the CPU oracle and into the HIP-backed modules.

Key names follow the reference's persistence ABI:
DINOv2 ``cls_token,pos_embed,mask_token,patch_embed.proj.*,blocks.{i}.*,norm.*``
(`dinov2/models/vision_transformer.py:62-162`), ``FeatureEncoder``
(`backbones/encoders.py:9-47`), ``CAViT``/``CACNN``
(`backbones/adapter_blocks.py:102-183`), ``FeatureDecoder``/``DecoderMLA``
(`backbones/decoders.py:7-164`), ``UNet`` (`backbones/unet_parts.py:106-137`).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

VIT_CONFIGS = {
    # name: (embed_dim, depth, heads, ffn)   dinov2/models/vision_transformer.py:305-357
    "vit_tiny_test": (128, 4, 2, "mlp"),  # test-only geometry (head dim 64 like all real archs)
    "vit_tiny_swiglu": (128, 4, 2, "swiglufused"),  # test-only: the ViT-g FFN at toy width
    "vit_large_d4": (1024, 4, 16, "mlp"),  # test-only: ViT-L width (the only D the reference adapters accept), 4 blocks
    "vit_base_d4": (768, 4, 12, "mlp"),  # test-only: ViT-B width (BASELINE config 2; MSDA head dim 96), 4 blocks
    "vit_giant2_d4": (1536, 4, 24, "swiglufused"),  # test-only: ViT-g width (config 5; SwiGLU 8192/4096, MSDA head dim 192)
    "vit_small": (384, 12, 6, "mlp"),
    "vit_base": (768, 12, 12, "mlp"),
    "vit_large": (1024, 24, 16, "mlp"),
    "vit_giant2": (1536, 40, 24, "swiglufused"),
}


def swiglu_hidden(dim: int) -> int:
    """`dinov2/layers/swiglu_ffn.py:66` with mlp_ratio 4."""
    return (int(dim * 4 * 2 / 3) + 7) // 8 * 8


def _uniform(name: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    """U(-1, 1) float64 keyed by (seed, name). Raw Philox words -> exact dyadic rationals."""
    key = (zlib.crc32(name.encode()) << 32) | (seed & 0xFFFFFFFF)
    bitgen = np.random.Philox(key=key)
    n = int(np.prod(shape)) if len(shape) else 1
    raw = bitgen.random_raw(n)  # uint64
    u = ((raw >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)  # (0,1)
    return (2.0 * u - 1.0).reshape(shape)


def tensor(name: str, shape, scale: float = 1.0, shift: float = 0.0, seed: int = 0) -> torch.Tensor:
    """``shift + scale * U(-1,1)`` as fp32."""
    a = shift + scale * _uniform(name, tuple(shape), seed)
    return torch.from_numpy(a.astype(np.float32))


def _linear(sd, prefix, out_f, in_f, seed, wscale=None, bias=True, bscale=0.02):
    wscale = wscale if wscale is not None else (3.0 / in_f) ** 0.5  # variance 1/in_f
    sd[prefix + ".weight"] = tensor(prefix + ".weight", (out_f, in_f), wscale, seed=seed)
    if bias:
        sd[prefix + ".bias"] = tensor(prefix + ".bias", (out_f,), bscale, seed=seed)


def _norm(sd, prefix, dim, seed):
    sd[prefix + ".weight"] = tensor(prefix + ".weight", (dim,), 0.2, 1.0, seed)
    sd[prefix + ".bias"] = tensor(prefix + ".bias", (dim,), 0.1, 0.0, seed)


def _bn(sd, prefix, c, seed):
    sd[prefix + ".weight"] = tensor(prefix + ".weight", (c,), 0.3, 1.0, seed)
    sd[prefix + ".bias"] = tensor(prefix + ".bias", (c,), 0.2, 0.0, seed)
    sd[prefix + ".running_mean"] = torch.zeros(c)
    sd[prefix + ".running_var"] = torch.ones(c)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _conv(sd, prefix, cout, cin, k, seed, bias, groups=1):
    fan_in = (cin // groups) * k * k
    sd[prefix + ".weight"] = tensor(prefix + ".weight", (cout, cin // groups, k, k), (3.0 / fan_in) ** 0.5, seed=seed)
    if bias:
        sd[prefix + ".bias"] = tensor(prefix + ".bias", (cout,), 0.05, seed=seed)


def make_vit_state_dict(arch: str = "vit_large", patch_size: int = 14, img_size: int = 518,
                        layerscale: str = "kernel", seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """DINOv2 ``DinoVisionTransformer`` state dict (block_chunks=0 => flat ``blocks.{i}``).

    layerscale="kernel": gamma in U[0.05, 0.5] so that branches are not multiplied
    away in kernel goldens; "init": the reference's 1e-5 init
    (`dinov2/configs/ssl_default_config.yaml:75`).
    """
    D, depth, heads, ffn = VIT_CONFIGS[arch]
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    npos = (img_size // patch_size) ** 2
    sd["cls_token"] = tensor("cls_token", (1, 1, D), 0.5, seed=seed)
    sd["pos_embed"] = tensor("pos_embed", (1, npos + 1, D), 0.2, seed=seed)
    sd["mask_token"] = torch.zeros(1, D)
    k = 3 * patch_size * patch_size
    sd["patch_embed.proj.weight"] = tensor("patch_embed.proj.weight", (D, 3, patch_size, patch_size),
                                           (3.0 / k) ** 0.5 * 2.0, seed=seed)
    sd["patch_embed.proj.bias"] = tensor("patch_embed.proj.bias", (D,), 0.1, seed=seed)
    for i in range(depth):
        p = f"blocks.{i}"
        _norm(sd, p + ".norm1", D, seed)
        _linear(sd, p + ".attn.qkv", 3 * D, D, seed, wscale=(3.0 / D) ** 0.5 * 1.5)
        _linear(sd, p + ".attn.proj", D, D, seed)
        _norm(sd, p + ".norm2", D, seed)
        if ffn == "mlp":
            _linear(sd, p + ".mlp.fc1", 4 * D, D, seed)
            _linear(sd, p + ".mlp.fc2", D, 4 * D, seed)
        else:
            hid = swiglu_hidden(D)
            _linear(sd, p + ".mlp.w12", 2 * hid, D, seed)
            _linear(sd, p + ".mlp.w3", D, hid, seed)
        for ls in ("ls1", "ls2"):
            if layerscale == "kernel":
                sd[f"{p}.{ls}.gamma"] = tensor(f"{p}.{ls}.gamma", (D,), 0.225, 0.275, seed)
            else:
                sd[f"{p}.{ls}.gamma"] = torch.full((D,), 1e-5)
    _norm(sd, "norm", D, seed)
    return sd


def make_encoder_state_dict(embed_dim: int = 1024, inplanes: int = 64, seed: int = 0):
    """`backbones/encoders.py:9-47`."""
    sd = OrderedDict()
    c = inplanes
    _conv(sd, "stem.0", c, 3, 3, seed, False); _bn(sd, "stem.1", c, seed)
    _conv(sd, "stem.3", c, c, 3, seed, False); _bn(sd, "stem.4", c, seed)
    _conv(sd, "stem.6", c, c, 3, seed, False); _bn(sd, "stem.7", c, seed)
    _conv(sd, "conv2.0", 2 * c, c, 3, seed, False); _bn(sd, "conv2.1", 2 * c, seed)
    _conv(sd, "conv3.0", 4 * c, 2 * c, 3, seed, False); _bn(sd, "conv3.1", 4 * c, seed)
    _conv(sd, "conv4.0", 8 * c, 4 * c, 3, seed, False); _bn(sd, "conv4.1", 8 * c, seed)
    for i, ci in enumerate((c, 2 * c, 4 * c, 8 * c), start=1):
        _conv(sd, f"fc{i}", embed_dim, ci, 1, seed, True)
    return sd


def _msda(sd, prefix, dim, heads, levels, points, seed, mode):
    """`backbones/ops/modules/ms_deform_attn.py:90-118`. mode="init" reproduces
    ``_reset_parameters`` semantics (zero offset/attention weights, ring bias);
    mode="kernel" uses non-degenerate values so every term is exercised."""
    import math
    n_off = heads * levels * points * 2
    n_aw = heads * levels * points
    thetas = torch.arange(heads, dtype=torch.float32) * (2.0 * math.pi / heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(heads, 1, 1, 2).repeat(1, levels, points, 1)
    for i in range(points):
        grid[:, :, i, :] *= i + 1
    if mode == "init":
        sd[prefix + ".sampling_offsets.weight"] = torch.zeros(n_off, dim)
        sd[prefix + ".sampling_offsets.bias"] = grid.reshape(-1).clone()
        sd[prefix + ".attention_weights.weight"] = torch.zeros(n_aw, dim)
        sd[prefix + ".attention_weights.bias"] = torch.zeros(n_aw)
    else:
        sd[prefix + ".sampling_offsets.weight"] = tensor(prefix + ".sampling_offsets.weight", (n_off, dim),
                                                         (3.0 / dim) ** 0.5 * 1.5, seed=seed)
        sd[prefix + ".sampling_offsets.bias"] = grid.reshape(-1) + tensor(prefix + ".sampling_offsets.bias",
                                                                         (n_off,), 0.5, seed=seed)
        sd[prefix + ".attention_weights.weight"] = tensor(prefix + ".attention_weights.weight", (n_aw, dim),
                                                          (3.0 / dim) ** 0.5, seed=seed)
        sd[prefix + ".attention_weights.bias"] = tensor(prefix + ".attention_weights.bias", (n_aw,), 0.5, seed=seed)
    _linear(sd, prefix + ".value_proj", dim, dim, seed, bscale=(0.0 if mode == "init" else 0.02))
    _linear(sd, prefix + ".output_proj", dim, dim, seed, bscale=(0.0 if mode == "init" else 0.02))


def make_cavit_state_dict(dim=1024, heads=8, levels=3, points=4, seed=0, mode="kernel"):
    """`backbones/adapter_blocks.py:149-183`; gamma init 0.0 (`train.py:90`) in mode="init"."""
    sd = OrderedDict()
    _norm(sd, "query_norm", dim, seed)
    _norm(sd, "feat_norm", dim, seed)
    _msda(sd, "attn", dim, heads, levels, points, seed, mode)
    if mode == "init":
        sd["gamma"] = torch.zeros(dim)
    else:
        sd["gamma"] = tensor("cavit.gamma", (dim,), 0.2, 0.3, seed)
    return sd


def make_cacnn_state_dict(dim=1024, heads=8, levels=1, points=4, cffn_ratio=0.25, seed=0, mode="kernel"):
    """`backbones/adapter_blocks.py:102-147`."""
    sd = OrderedDict()
    hid = int(dim * cffn_ratio)
    _norm(sd, "query_norm", dim, seed)
    _norm(sd, "feat_norm", dim, seed)
    _msda(sd, "attn", dim, heads, levels, points, seed + 1, mode)
    _linear(sd, "ffn.fc1", hid, dim, seed)
    _conv(sd, "ffn.dwconv.dwconv", hid, hid, 3, seed, True, groups=hid)
    _linear(sd, "ffn.fc2", dim, hid, seed)
    _norm(sd, "ffn_norm", dim, seed)
    return sd


def make_feature_decoder_state_dict(embed_dim=1024, num_classes=2, features=(1024, 512, 256, 128, 64), seed=0):
    """`backbones/decoders.py:92-135` (note ``features[0]*3`` input channels)."""
    sd = OrderedDict()
    chans = [features[0] * 3, features[1], features[2], features[3], features[4]]
    for i in range(4):
        _conv(sd, f"decoder_{i + 1}.0", chans[i + 1], chans[i], 3, seed, True)
        _bn(sd, f"decoder_{i + 1}.1", chans[i + 1], seed)
    _conv(sd, "final_out", num_classes, chans[4], 3, seed, True)
    return sd


def make_setr_state_dict(in_channels=1024, out_channels=2, features=(512, 256, 128, 64), seed=0):
    """`backbones/decoders.py:167-196` (DecoderSETR: the FeatureDecoder layout fed by ``in_channels``)."""
    sd = OrderedDict()
    chans = [in_channels] + list(features)
    for i in range(4):
        _conv(sd, f"decoder_{i + 1}.0", chans[i + 1], chans[i], 3, seed, True)
        _bn(sd, f"decoder_{i + 1}.1", chans[i + 1], seed)
    _conv(sd, "final_out", out_channels, chans[4], 3, seed, True)
    return sd


def make_setrf_state_dict(in_channels=1024, out_channels=2, features=(512, 256, 128, 64), seed=0):
    """`backbones/decoders.py:205-236` (DecoderSETRF: SETR stages whose 3rd / 4th / final convs take the skip concat)."""
    sd = OrderedDict()
    f = list(features)
    cin = [in_channels, f[0], 2 * f[1], 2 * f[2]]
    for i in range(4):
        _conv(sd, f"decoder_{i + 1}.0", f[i], cin[i], 3, seed, True)
        _bn(sd, f"decoder_{i + 1}.1", f[i], seed)
    _conv(sd, "final_out", out_channels, 2 * f[3], 3, seed, True)
    return sd


def make_decoder_mla_state_dict(mla_channels=1024, mlahead_channels=128, num_classes=2, seed=0):
    """`backbones/decoders.py:7-80`."""
    sd = OrderedDict()
    for h in ("head2", "head3", "head4", "head5"):
        p = f"mlahead.{h}"
        _conv(sd, p + ".0", mlahead_channels, mla_channels, 3, seed, False); _bn(sd, p + ".1", mlahead_channels, seed)
        _conv(sd, p + ".3", mlahead_channels, mlahead_channels, 3, seed, False); _bn(sd, p + ".4", mlahead_channels, seed)
    _conv(sd, "cls.0", 256, 4 * mlahead_channels, 3, seed, True); _bn(sd, "cls.1", 256, seed)
    _conv(sd, "cls_1.0", 128, 256, 3, seed, True); _bn(sd, "cls_1.1", 128, seed)
    _conv(sd, "cls_2.0", 64, 128, 3, seed, True); _bn(sd, "cls_2.1", 64, seed)
    _conv(sd, "cls_3", num_classes, 64, 3, seed, True)
    return sd


def make_unet_state_dict(base: int = 384, n_classes: int = 2, seed: int = 0):
    """Width-generic restatement of `backbones/unet_parts.py:106-124` (``bilinear=False``):
    Down(C,2C), Down(2C,4C), Up(4C,2C), Up(2C,C), Up_wc(C,C/2), Up_wc(C/2,C/4), OutConv(C/4,cls)."""
    sd = OrderedDict()
    C = base

    def dconv(p, cin, cout):
        _conv(sd, p + ".double_conv.0", cout, cin, 3, seed, False); _bn(sd, p + ".double_conv.1", cout, seed)
        _conv(sd, p + ".double_conv.3", cout, cout, 3, seed, False); _bn(sd, p + ".double_conv.4", cout, seed)

    dconv("down3.maxpool_conv.1", C, 2 * C)
    dconv("down4.maxpool_conv.1", 2 * C, 4 * C)

    def convT(p, cin, cout):
        sd[p + ".weight"] = tensor(p + ".weight", (cin, cout, 2, 2), (3.0 / (cin * 4)) ** 0.5 * 2, seed=seed)
        sd[p + ".bias"] = tensor(p + ".bias", (cout,), 0.05, seed=seed)

    convT("up1.up", 4 * C, 2 * C); dconv("up1.conv", 4 * C, 2 * C)
    convT("up2.up", 2 * C, C); dconv("up2.conv", 2 * C, C)
    convT("up3.up", C, C); dconv("up3.conv", C, C // 2)
    convT("up4.up", C // 2, C // 2); dconv("up4.conv", C // 2, C // 4)
    _conv(sd, "outc.conv", n_classes, C // 4, 1, seed, True)
    return sd


def make_masktrans_state_dict(d_encoder: int, d_model: int, n_layers: int = 2, n_cls: int = 2, d_ff: int = None, seed: int = 0,
                              mode: str = "kernel"):
    """``MaskTransformer`` of `eval/eval_dinov2_masktrans.py:400-439` (blocks: `backbones/masktrans_block.py`).
    mode="init": the scales of the reference's own initialisation (`:388-396,427-436`: Linear weights std 0.02 with zero bias,
    LayerNorm 1 / 0, cls_emb std 0.02, proj_patch / proj_classes std d_model^-0.5) — the head as the reference starts training
    it; mode="kernel": unit-gain weights, non-zero biases and LayerNorm affines, so that every term carries signal and the two
    residual branches are as large as the stream itself (stress case for the kernels)."""
    sd = OrderedDict()
    d_ff = d_ff or 4 * d_model
    init = mode == "init"
    r3 = 3.0 ** 0.5
    sd["cls_emb"] = tensor("mt.cls_emb", (1, n_cls, d_model), 0.02 * r3 if init else 0.5, seed=seed)
    sd["proj_patch"] = tensor("mt.proj_patch", (d_model, d_model), (3.0 / d_model) ** 0.5, seed=seed)
    sd["proj_classes"] = tensor("mt.proj_classes", (d_model, d_model), (3.0 / d_model) ** 0.5, seed=seed)

    def lin(p, cout, cin):
        sd[p + ".weight"] = tensor("mt." + p + ".weight", (cout, cin), 0.02 * r3 if init else (3.0 / cin) ** 0.5, seed=seed)
        sd[p + ".bias"] = torch.zeros(cout) if init else tensor("mt." + p + ".bias", (cout,), 0.05, seed=seed)

    def ln(p, c):
        sd[p + ".weight"] = torch.ones(c) if init else tensor("mt." + p + ".weight", (c,), 0.3, 1.0, seed)
        sd[p + ".bias"] = torch.zeros(c) if init else tensor("mt." + p + ".bias", (c,), 0.2, 0.0, seed)

    for i in range(n_layers):
        b = f"blocks.{i}"
        ln(b + ".norm1", d_model); lin(b + ".attn.qkv", 3 * d_model, d_model); lin(b + ".attn.proj", d_model, d_model)
        ln(b + ".norm2", d_model); lin(b + ".mlp.fc1", d_ff, d_model); lin(b + ".mlp.fc2", d_model, d_ff)
    lin("proj_dec", d_model, d_encoder)
    ln("decoder_norm", d_model)
    ln("mask_norm", n_cls)
    return sd


def make_or_unet_state_dict(embed_dim: int = 384, n_classes: int = 2, base: int = 64, seed: int = 0):
    """OR-UNet fuse head `eval/eval_dinov2_or_unet_fuse.py:426-447` (bilinear=False): DoubleConv(3, base), four Down, four Up
    (each with its skip), OutConv, and the FCUUp projections expand_block_2/3/4 (embed_dim -> 4*base / 2*base / base).
    The reference's widths are base = 64; the parameter is for small test geometries only."""
    sd = OrderedDict()
    c = [base, 2 * base, 4 * base, 8 * base, 16 * base]

    def dconv(p, cin, cout):
        _conv(sd, p + ".double_conv.0", cout, cin, 3, seed, False); _bn(sd, p + ".double_conv.1", cout, seed)
        _conv(sd, p + ".double_conv.3", cout, cout, 3, seed, False); _bn(sd, p + ".double_conv.4", cout, seed)

    dconv("inc", 3, c[0])
    for i in range(4):
        dconv(f"down{i + 1}.maxpool_conv.1", c[i], c[i + 1])
    for i in range(4):
        cin, cout = c[4 - i], c[3 - i]
        sd[f"up{i + 1}.up.weight"] = tensor(f"up{i + 1}.up.weight", (cin, cin // 2, 2, 2), (3.0 / (cin * 4)) ** 0.5 * 2, seed=seed)
        sd[f"up{i + 1}.up.bias"] = tensor(f"up{i + 1}.up.bias", (cin // 2,), 0.05, seed=seed)
        dconv(f"up{i + 1}.conv", cin, cout)
    _conv(sd, "outc.conv", n_classes, c[0], 1, seed, True)
    for k, cout in ((2, c[2]), (3, c[1]), (4, c[0])):
        _conv(sd, f"expand_block_{k}.conv_project", cout, embed_dim, 1, seed, True)
        _bn(sd, f"expand_block_{k}.bn", cout, seed)
    return sd


def synthetic_batch(batch: int, size: int = 588, num_classes: int = 2, seed: int = 0):
    """SURVEY.md §8d synthetic inputs: images U[0,1) (no mean/std normalisation,
    `tools/dataset.py:159`), binary masks with ~30 % foreground and one
    all-background image per batch (Dice epsilon path); multi-class masks are
    piecewise-constant 14x14 blocks."""
    img = (tensor(f"img{batch}x{size}", (batch, 3, size, size), 0.5, 0.5, seed)).clamp_(0.0, 1.0 - 1e-7)
    if num_classes == 2:
        m = (tensor(f"mask{batch}x{size}", (batch, size, size), 0.5, 0.5, seed) > 0.7).long()
        if batch > 1:
            m[-1].zero_()
    else:
        g = (size + 13) // 14
        blocks = tensor(f"mmask{batch}x{size}", (batch, g, g), 0.5, 0.5, seed)
        m = (blocks * num_classes).long().clamp_(0, num_classes - 1)
        m = m.repeat_interleave(14, 1).repeat_interleave(14, 2)[:, :size, :size].contiguous()
    return img, m
