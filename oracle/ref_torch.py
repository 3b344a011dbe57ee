"""CPU ORACLE — test infrastructure, NOT product code.

Plain fp32 eager-PyTorch restatement of the reference's hot path (the
DINOv2-ViT + adapter + decode-head + Dice/CE segmentation training step of
weimengmeng1999/AdapterSIS).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
``adaptersis_amd`` never does (``tests/test_boundaries.py`` enforces it).

Every function cites the reference file:line it follows (paths relative to the
reference root).  All functions are *functional*: they take a ``state_dict``
with the reference's key names (plus a key prefix) and plain tensors, so the
same dict can be loaded into the imported reference, into this oracle and into
the HIP-backed modules.

Pinning: the reference has no tests, fixtures or golden vectors of its own
(SURVEY.md §4), so this restatement is pinned by golden tensors produced by
importing the reference's modules in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.pt``) and checked in
``tests/test_oracle_golden.py``.

Generalisations beyond the reference: embed dim, pyramid
level shapes and DWConv split sizes are parameters taken from the encoder's
real outputs instead of the hard-coded 1024 / 18x18 / h//8,16,32; with the
reference's 588x588, D=1024 geometry they reduce to the reference exactly.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------
# DINOv2 ViT
# ----------------------------------------------------------------------------
def layer_norm(x, sd: SD, p: str, eps: float = 1e-6):
    """nn.LayerNorm(eps=1e-6): `dinov2/models/vision_transformer.py:89`."""
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def attention(x, sd: SD, p: str, num_heads: int):
    """`dinov2/layers/attention.py:56-69` (MemEffAttention falls through to this
    without xformers, `attention.py:74-77`)."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias"))
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[p + ".proj.weight"], sd.get(p + ".proj.bias"))


def mlp(x, sd: SD, p: str):
    """`dinov2/layers/mlp.py:34-40` (erf GELU) or `swiglu_ffn.py:30-34`."""
    if p + ".fc1.weight" in sd:
        h = F.gelu(F.linear(x, sd[p + ".fc1.weight"], sd.get(p + ".fc1.bias")))
        return F.linear(h, sd[p + ".fc2.weight"], sd.get(p + ".fc2.bias"))
    x12 = F.linear(x, sd[p + ".w12.weight"], sd.get(p + ".w12.bias"))
    x1, x2 = x12.chunk(2, dim=-1)
    return F.linear(F.silu(x1) * x2, sd[p + ".w3.weight"], sd.get(p + ".w3.bias"))


def block(x, sd: SD, p: str, num_heads: int):
    """Eval branch of `dinov2/layers/block.py:111-113` with LayerScale
    (`layer_scale.py:26-27`)."""
    a = attention(layer_norm(x, sd, p + ".norm1"), sd, p + ".attn", num_heads)
    x = x + a * sd[p + ".ls1.gamma"] if p + ".ls1.gamma" in sd else x + a
    m = mlp(layer_norm(x, sd, p + ".norm2"), sd, p + ".mlp")
    x = x + m * sd[p + ".ls2.gamma"] if p + ".ls2.gamma" in sd else x + m
    return x


def patch_embed(img, sd: SD, patch: int = 14):
    """`dinov2/layers/patch_embed.py:68-81`."""
    _, _, H, W = img.shape
    assert H % patch == 0, f"Input image height {H} is not a multiple of patch height {patch}"
    assert W % patch == 0, f"Input image width {W} is not a multiple of patch width: {patch}"
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    return x.flatten(2).transpose(1, 2)


def interpolate_pos_encoding(pos_embed, npatch: int, w: int, h: int, patch: int = 14):
    """`dinov2/models/vision_transformer.py:164-188` — bicubic with the
    ``+0.1`` scale-factor quirk (grid uses (w0+0.1)/sqrt(N), not w0/sqrt(N))."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    pos_embed = pos_embed.float()
    class_pos = pos_embed[:, 0]
    patch_pos = pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0, h0 = w // patch + 0.1, h // patch + 0.1
    s = int(math.sqrt(N))
    patch_pos = F.interpolate(
        patch_pos.reshape(1, s, s, dim).permute(0, 3, 1, 2),
        scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)),
        mode="bicubic",
    )
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1)


def prepare_tokens(img, sd: SD, patch: int = 14):
    """`vision_transformer.py:190-199` (masks=None)."""
    B, _, w, h = img.shape
    x = patch_embed(img, sd, patch)
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), x), dim=1)
    return x + interpolate_pos_encoding(sd["pos_embed"], x.shape[1] - 1, w, h, patch)


def vit_depth(sd: SD) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))


def get_intermediate_layers(img, sd: SD, num_heads: int, n: int = 4, patch: int = 14, norm: bool = True):
    """`vision_transformer.py:237-247,263-287` with return_class_token=True:
    returns [(patch_tokens, cls_token)] for the last n blocks, final norm applied."""
    x = prepare_tokens(img, sd, patch)
    depth = vit_depth(sd)
    outs = []
    for i in range(depth):
        x = block(x, sd, f"blocks.{i}", num_heads)
        if i >= depth - n:
            outs.append(x)
    if norm:
        outs = [layer_norm(o, sd, "norm") for o in outs]
    return [(o[:, 1:], o[:, 0]) for o in outs]


def forward_features(img, sd: SD, num_heads: int, patch: int = 14):
    """`vision_transformer.py:212-235` (is_training=True dict, config-4 path)."""
    x = prepare_tokens(img, sd, patch)
    for i in range(vit_depth(sd)):
        x = block(x, sd, f"blocks.{i}", num_heads)
    xn = layer_norm(x, sd, "norm")
    return {"x_norm_clstoken": xn[:, 0], "x_norm_patchtokens": xn[:, 1:], "x_prenorm": x}


# ----------------------------------------------------------------------------
# Adapter: deform inputs, MSDA, CAViT, CACNN
# ----------------------------------------------------------------------------
def get_reference_points(spatial_shapes: Sequence[Tuple[int, int]]):
    """`backbones/adapter_blocks.py:9-22`: cell centres, (x, y) order."""
    pts = []
    for (H_, W_) in spatial_shapes:
        ry = (torch.arange(H_, dtype=torch.float32) + 0.5)
        rx = (torch.arange(W_, dtype=torch.float32) + 0.5)
        # linspace(0.5, H-0.5, H) == arange+0.5 exactly for these sizes
        ry = torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32)
        rx = torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32)
        gy, gx = torch.meshgrid(ry, rx, indexing="ij")
        pts.append(torch.stack((gx.reshape(-1)[None] / W_, gy.reshape(-1)[None] / H_), -1))
    return torch.cat(pts, 1)[:, :, None]


def deform_inputs(h: int, w: int, patch: int = 14, cnn_shapes: Optional[Sequence[Tuple[int, int]]] = None):
    """`adapter_blocks.py:24-38`.  ``cnn_shapes`` (the encoder's real c2,c3,c4
    shapes) replaces the reference's ``h//8,h//16,h//32`` guess, which is only
    right at 588."""
    if cnn_shapes is None:
        cnn_shapes = [(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)]
    cnn_shapes = [tuple(s) for s in cnn_shapes]
    vit_shape = [(h // patch, w // patch)]

    def starts(shapes):
        sizes = [a * b for a, b in shapes]
        return torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)[:-1]), dtype=torch.long)

    d1 = [get_reference_points(vit_shape), torch.tensor(cnn_shapes, dtype=torch.long), starts(cnn_shapes)]
    d2 = [get_reference_points(cnn_shapes), torch.tensor(vit_shape, dtype=torch.long), starts(vit_shape)]
    return d1, d2


def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """`backbones/ops/modules/ms_deform_attn.py:33-54` (grid_sample form)."""
    N_, S_, M_, D_ = value.shape
    _, Lq_, M_, L_, P_, _ = sampling_locations.shape
    shapes = [(int(a), int(b)) for a, b in spatial_shapes]
    value_list = value.split([H_ * W_ for H_, W_ in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lid, (H_, W_) in enumerate(shapes):
        v = value_list[lid].flatten(2).transpose(1, 2).reshape(N_ * M_, D_, H_, W_)
        g = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(N_ * M_, 1, Lq_, L_ * P_)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(N_, M_ * D_, Lq_)
    return out.transpose(1, 2).contiguous()


def ms_deform_attn(query, reference_points, feat, spatial_shapes, sd: SD, p: str,
                   n_heads: int, n_levels: int, n_points: int):
    """`ms_deform_attn.py:120-185` (ratio=1.0, no padding mask)."""
    N, Lq, C = query.shape
    _, Lin, _ = feat.shape
    assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == Lin
    value = F.linear(feat, sd[p + ".value_proj.weight"], sd[p + ".value_proj.bias"]).view(N, Lin, n_heads, C // n_heads)
    off = F.linear(query, sd[p + ".sampling_offsets.weight"], sd[p + ".sampling_offsets.bias"])
    off = off.view(N, Lq, n_heads, n_levels, n_points, 2)
    aw = F.linear(query, sd[p + ".attention_weights.weight"], sd[p + ".attention_weights.bias"])
    aw = F.softmax(aw.view(N, Lq, n_heads, n_levels * n_points), -1).view(N, Lq, n_heads, n_levels, n_points)
    if reference_points.shape[-1] != 2:
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]))
    normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1).to(query.dtype)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = ms_deform_attn_core(value, spatial_shapes, loc, aw)
    return F.linear(out, sd[p + ".output_proj.weight"], sd[p + ".output_proj.bias"])


def cavit(query, reference_points, feat, spatial_shapes, sd: SD, p: str = "", n_heads=8, n_levels=3, n_points=4):
    """`adapter_blocks.py:170-183`: ``query + gamma * MSDA(LN(query), LN(feat))``."""
    pre = p + "." if p else ""
    a = ms_deform_attn(layer_norm(query, sd, pre + "query_norm"), reference_points,
                       layer_norm(feat, sd, pre + "feat_norm"), spatial_shapes, sd, pre + "attn",
                       n_heads, n_levels, n_points)
    return query + sd[pre + "gamma"] * a


def dwconv(x, sd: SD, p: str, grids: Sequence[Tuple[int, int]]):
    """`adapter_blocks.py:67-80`: one shared depthwise 3x3 applied to each token
    grid.  The reference hard-codes n=18*18 and (2H+1, H, H/2); ``grids`` are
    the real (h, w) of each pyramid level."""
    B, N, C = x.shape
    outs, s = [], 0
    for (gh, gw) in grids:
        xi = x[:, s:s + gh * gw].transpose(1, 2).reshape(B, C, gh, gw)
        xi = F.conv2d(xi, sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=1, groups=C)
        outs.append(xi.flatten(2).transpose(1, 2))
        s += gh * gw
    assert s == N
    return torch.cat(outs, dim=1)


def conv_ffn(x, sd: SD, p: str, grids):
    """`adapter_blocks.py:93-100`."""
    x = F.linear(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])
    x = F.gelu(dwconv(x, sd, p + ".dwconv", grids))
    return F.linear(x, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def cacnn(query, reference_points, feat, spatial_shapes, grids, sd: SD, p: str = "", n_heads=8, n_levels=1, n_points=4):
    """`adapter_blocks.py:130-147`."""
    pre = p + "." if p else ""
    a = ms_deform_attn(layer_norm(query, sd, pre + "query_norm"), reference_points,
                       layer_norm(feat, sd, pre + "feat_norm"), spatial_shapes, sd, pre + "attn",
                       n_heads, n_levels, n_points)
    query = query + a
    return query + conv_ffn(layer_norm(query, sd, pre + "ffn_norm"), sd, pre + "ffn", grids)


# ----------------------------------------------------------------------------
# CNN encoder / decoders (BatchNorm in TRAIN mode: batch statistics)
# ----------------------------------------------------------------------------
def batch_norm_train(x, sd: SD, p: str, eps: float = 1e-5, momentum: float = 0.1, update: bool = False):
    """nn.BatchNorm2d / SyncBatchNorm (single process) in train mode
    (`encoders.py:12-40` is never put in eval mode; `decoders.py:111-131`)."""
    rm, rv = (sd[p + ".running_mean"], sd[p + ".running_var"]) if update else (None, None)
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, momentum, eps)
    if update and p + ".num_batches_tracked" in sd:
        sd[p + ".num_batches_tracked"] += 1
    return y


def feature_encoder(img, sd: SD, p: str = "", update_bn: bool = False):
    """`backbones/encoders.py:49-74` -> (c1 map, c2, c3, c4 tokens) + real shapes."""
    pre = p + "." if p else ""

    def cbr(x, conv, bn, stride, pad):
        x = F.conv2d(x, sd[pre + conv + ".weight"], None, stride=stride, padding=pad)
        return F.relu(batch_norm_train(x, sd, pre + bn, update=update_bn))

    x = cbr(img, "stem.0", "stem.1", 2, 1)
    x = cbr(x, "stem.3", "stem.4", 1, 1)
    x = cbr(x, "stem.6", "stem.7", 1, 1)
    c1 = F.max_pool2d(x, 3, 2, 1)
    c2 = cbr(c1, "conv2.0", "conv2.1", 2, 0)
    c3 = cbr(c2, "conv3.0", "conv3.1", 2, 0)
    c4 = cbr(c3, "conv4.0", "conv4.1", 2, 1)
    shapes = [tuple(c.shape[-2:]) for c in (c2, c3, c4)]
    c1 = F.conv2d(c1, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
    c2 = F.conv2d(c2, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])
    c3 = F.conv2d(c3, sd[pre + "fc3.weight"], sd[pre + "fc3.bias"])
    c4 = F.conv2d(c4, sd[pre + "fc4.weight"], sd[pre + "fc4.bias"])
    bs, dim = c1.shape[:2]
    c2 = c2.view(bs, dim, -1).transpose(1, 2)
    c3 = c3.view(bs, dim, -1).transpose(1, 2)
    c4 = c4.view(bs, dim, -1).transpose(1, 2)
    return c1, c2, c3, c4, shapes


def feature_decoder(x, sd: SD, p: str = "", update_bn: bool = False, training: bool = True):
    """`backbones/decoders.py:109-164`: 4x [conv3x3 -> BN -> ReLU -> bilinear x2 align_corners=True] ->
    conv3x3.  BN uses batch statistics in training (`train.py:262`) and the running statistics under
    ``seg_decoder.eval()`` (`train.py:451`)."""
    pre = p + "." if p else ""
    for i in range(1, 5):
        q = f"{pre}decoder_{i}"
        x = F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], padding=1)
        if training:
            x = F.relu(batch_norm_train(x, sd, q + ".1", update=update_bn))
        else:
            x = F.relu(F.batch_norm(x, sd[q + ".1.running_mean"], sd[q + ".1.running_var"], sd[q + ".1.weight"],
                                    sd[q + ".1.bias"], False, 0.1, 1e-5))
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(x, sd[pre + "final_out.weight"], sd[pre + "final_out.bias"], padding=1)


def decoder_setrf(x, c1, c2, c3, sd: SD, p: str = "", update_bn: bool = False):
    """`backbones/decoders.py:205-257` DecoderSETRF: the SETR stages with the CNN pyramid fused in — after stage 2 the
    map is zero-padded (centred, `:240-243`) to c3's size and concatenated with it, likewise c2 after stage 3 and c1
    after stage 4, then the final conv3x3."""
    pre = p + "." if p else ""

    def stage(x, i):
        q = f"{pre}decoder_{i}"
        x = F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], padding=1)
        x = F.relu(batch_norm_train(x, sd, q + ".1", update=update_bn))
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)

    def fuse(x, c):
        dy, dx = c.shape[2] - x.shape[2], c.shape[3] - x.shape[3]
        x = F.pad(x, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return torch.cat([x, c], dim=1)

    x = stage(stage(x, 1), 2)
    x = stage(fuse(x, c3), 3)
    x = stage(fuse(x, c2), 4)
    x = fuse(x, c1)
    return F.conv2d(x, sd[pre + "final_out.weight"], sd[pre + "final_out.bias"], padding=1)


def decoder_mla(i0, i1, i2, i3, sd: SD, img_size: int = 588, p: str = "", update_bn: bool = False):
    """`backbones/decoders.py:7-89`."""
    pre = p + "." if p else ""

    def head(x, h):
        q = f"{pre}mlahead.{h}"
        x = F.conv2d(x, sd[q + ".0.weight"], None, padding=1)
        x = F.relu(batch_norm_train(x, sd, q + ".1", update=update_bn))
        x = F.conv2d(x, sd[q + ".3.weight"], None, padding=1)
        x = F.relu(batch_norm_train(x, sd, q + ".4", update=update_bn))
        return F.interpolate(x, 4 * x.shape[-1], mode="bilinear", align_corners=True)

    x = torch.cat([head(i0, "head2"), head(i1, "head3"), head(i2, "head4"), head(i3, "head5")], dim=1)
    for name in ("cls", "cls_1", "cls_2"):
        q = pre + name
        x = F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], padding=1)
        x = F.relu(batch_norm_train(x, sd, q + ".1", update=update_bn))
    x = F.conv2d(x, sd[pre + "cls_3.weight"], sd[pre + "cls_3.bias"], padding=1)
    return F.interpolate(x, size=img_size, mode="bilinear")


def unet(x, sd: SD, p: str = "", update_bn: bool = False):
    """Width-generic `backbones/unet_parts.py:126-137` (bilinear=False)."""
    pre = p + "." if p else ""

    def dconv(x, q):
        x = F.conv2d(x, sd[q + ".double_conv.0.weight"], None, padding=1)
        x = F.relu(batch_norm_train(x, sd, q + ".double_conv.1", update=update_bn))
        x = F.conv2d(x, sd[q + ".double_conv.3.weight"], None, padding=1)
        return F.relu(batch_norm_train(x, sd, q + ".double_conv.4", update=update_bn))

    def up(x1, q):
        return F.conv_transpose2d(x1, sd[q + ".weight"], sd[q + ".bias"], stride=2)

    def pad_to(x1, x2):
        dY, dX = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        return F.pad(x1, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])

    x3 = x
    x4 = dconv(F.max_pool2d(x3, 2), pre + "down3.maxpool_conv.1")
    x5 = dconv(F.max_pool2d(x4, 2), pre + "down4.maxpool_conv.1")
    y = dconv(torch.cat([x4, pad_to(up(x5, pre + "up1.up"), x4)], 1), pre + "up1.conv")
    y = dconv(torch.cat([x3, pad_to(up(y, pre + "up2.up"), x3)], 1), pre + "up2.conv")
    y = dconv(up(y, pre + "up3.up"), pre + "up3.conv")
    y = dconv(up(y, pre + "up4.up"), pre + "up4.conv")
    return F.conv2d(y, sd[pre + "outc.conv.weight"], sd[pre + "outc.conv.bias"])


def masktrans_block(x, sd: SD, p: str, num_heads: int, drop=None):
    """`backbones/masktrans_block.py:75-89` (drop_path 0): pre-norm block, nn.LayerNorm default eps 1e-5, attention `:34-72` =
    softmax((q k^T) * head_dim^-0.5) v with qkv / proj biases, FeedForward `:11-31` = fc1 -> GELU -> fc2.
    ``drop`` = (p, masks) replays given keep masks in place of nn.Dropout's own draws (`:19,43-45,66,70,27-29`:
    attention probabilities, projection output, behind GELU, behind fc2): y = x * keep / (1 - p), training-mode nn.Dropout."""
    def dr(t, key):
        if drop is None:
            return t
        pr, masks = drop
        return t * masks[key].to(t.dtype) / (1.0 - pr)
    B, N, C = x.shape
    hd = C // num_heads
    h = layer_norm(x, sd, p + ".norm1", 1e-5)
    qkv = F.linear(h, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"]).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    attn = dr(((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1), "attn")
    a = (attn @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = x + dr(F.linear(a, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"]), "proj")
    h = layer_norm(x, sd, p + ".norm2", 1e-5)
    h = dr(F.gelu(F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])), "gelu")
    return x + dr(F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"]), "fc2")


def mask_transformer(tok, sd: SD, num_heads: int, n_cls: int, p: str = "", taps: Optional[dict] = None, drop=None):
    """`eval/eval_dinov2_masktrans.py:441-462` ``MaskTransformer.forward``: tokens (B, N, d_encoder) -> masks (B, n_cls, GS, GS).
    ``taps``: receives the cosines in front of mask_norm (B, N, n_cls)."""
    pre = p + "." if p else ""
    x = F.linear(tok, sd[pre + "proj_dec.weight"], sd[pre + "proj_dec.bias"])
    x = torch.cat((x, sd[pre + "cls_emb"].expand(x.size(0), -1, -1)), 1)
    i = 0
    while f"{pre}blocks.{i}.norm1.weight" in sd:
        x = masktrans_block(x, sd, f"{pre}blocks.{i}", num_heads, None if drop is None else (drop[0], drop[1][i]))
        i += 1
    x = layer_norm(x, sd, pre + "decoder_norm", 1e-5)
    patches, cls = x[:, :-n_cls] @ sd[pre + "proj_patch"], x[:, -n_cls:] @ sd[pre + "proj_classes"]
    patches = patches / patches.norm(dim=-1, keepdim=True)
    cls = cls / cls.norm(dim=-1, keepdim=True)
    cos = patches @ cls.transpose(1, 2)
    if taps is not None:
        taps["cos"] = cos
    masks = layer_norm(cos, sd, pre + "mask_norm", 1e-5)
    B, N, _ = masks.shape
    gs = int(round(N ** 0.5))
    return masks.reshape(B, gs, gs, n_cls).permute(0, 3, 1, 2)


def dice_of_argmax(output, target, eps: float = 1e-7):
    """`eval/eval_dinov2_masktrans.py:83-92,306-311`: 1 - dice of the hard prediction (argmax) — a constant of the step."""
    preds = torch.softmax(output, 1).max(1)[1]
    o, t = preds.reshape(-1).float(), target.reshape(-1).float()
    return 1.0 - (2.0 * (o * t).sum() + eps) / (o.sum() + t.sum() + eps)


def or_unet_fuse(img, x_o, x_t2, x_d2, sd: SD, p: str = "", update_bn: bool = False):
    """OR-UNet multi-scale fuse head, `eval/eval_dinov2_or_unet_fuse.py:426-486` (``UNet.forward``, bilinear=False,
    dw_stride=1): a full-resolution UNet on the image whose first three encoder levels are fused with ViT feature maps of
    the image at scale 1.5 / 1 / 0.5 through FCUUp (1x1 conv -> BatchNorm(eps=1e-6) -> ReLU -> nearest resize, `:511-530`)
    and FusionModel (add -> ReLU, `:502-510`).  DoubleConv / Down / Up / OutConv: `backbones/unet_parts.py:6-64,95-101`."""
    pre = p + "." if p else ""

    def dconv(x, q):
        x = F.conv2d(x, sd[q + ".double_conv.0.weight"], None, padding=1)
        x = F.relu(batch_norm_train(x, sd, q + ".double_conv.1", update=update_bn))
        x = F.conv2d(x, sd[q + ".double_conv.3.weight"], None, padding=1)
        return F.relu(batch_norm_train(x, sd, q + ".double_conv.4", update=update_bn))

    def fcu(x_r, q, H, W):
        z = F.conv2d(x_r, sd[q + ".conv_project.weight"], sd[q + ".conv_project.bias"])
        z = F.relu(batch_norm_train(z, sd, q + ".bn", update=update_bn, eps=1e-6))
        return F.interpolate(z, size=(H, W))

    def up(x1, x2, q):
        x1 = F.conv_transpose2d(x1, sd[q + ".up.weight"], sd[q + ".up.bias"], stride=2)
        dY, dX = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])
        return dconv(torch.cat([x2, x1], 1), q + ".conv")

    x1 = dconv(img, pre + "inc")
    x1 = F.relu(x1 + fcu(x_t2, pre + "expand_block_4", *x1.shape[2:]))
    x2 = dconv(F.max_pool2d(x1, 2), pre + "down1.maxpool_conv.1")
    x2 = F.relu(x2 + fcu(x_o, pre + "expand_block_3", *x2.shape[2:]))
    x3 = dconv(F.max_pool2d(x2, 2), pre + "down2.maxpool_conv.1")
    x3 = F.relu(x3 + fcu(x_d2, pre + "expand_block_2", *x3.shape[2:]))
    x4 = dconv(F.max_pool2d(x3, 2), pre + "down3.maxpool_conv.1")
    x5 = dconv(F.max_pool2d(x4, 2), pre + "down4.maxpool_conv.1")
    y = up(x5, x4, pre + "up1")
    y = up(y, x3, pre + "up2")
    y = up(y, x2, pre + "up3")
    y = up(y, x1, pre + "up4")
    return F.conv2d(y, sd[pre + "outc.conv.weight"], sd[pre + "outc.conv.bias"])


# ----------------------------------------------------------------------------
# Losses / metrics
# ----------------------------------------------------------------------------
def one_hot(target, num_classes: int):
    """(B,H,W) int64 -> (B,C,H,W) fp32; replaces `segloss/dice.py:13-19` whose
    ``.cuda()`` cannot run on CPU (same maths; branch `dice.py:24-25`)."""
    return F.one_hot(target.long(), num_classes).permute(0, 3, 1, 2).float()


def dc_loss(output, target_onehot):
    """`segloss/dice.py:22-33`: softmax inside, eps 1e-19, ``1 - mean dice``."""
    p = torch.softmax(output, 1)
    axes = list(range(2, p.dim()))
    inter = torch.sum(p * target_onehot, axes)
    dice = (2 * inter) / (torch.sum(p, axes) + torch.sum(target_onehot, axes) + 10e-20)
    return 1.0 - dice.mean()


def soft_dice_loss(x, target_onehot, smooth: float = 1.0):
    """`segloss/dice_loss.py:255-291,31-81` (no nonlin, do_bg, batch_dice=False)."""
    axes = list(range(2, x.dim()))
    tp = (x * target_onehot).sum(axes)
    fp = (x * (1 - target_onehot)).sum(axes)
    fn = ((1 - x) * target_onehot).sum(axes)
    return -((2 * tp + smooth) / (2 * tp + fp + fn + smooth)).mean()


def tversky_loss(x, target_onehot, smooth: float = 1.0, alpha: float = 0.3, beta: float = 0.7):
    """`segloss/dice_loss.py:333-372` (no nonlin, do_bg, batch_dice=False)."""
    axes = list(range(2, x.dim()))
    tp = (x * target_onehot).sum(axes)
    fp = (x * (1 - target_onehot)).sum(axes)
    fn = ((1 - x) * target_onehot).sum(axes)
    return -((tp + smooth) / (tp + alpha * fp + beta * fn + smooth)).mean()


def _iou_bool(a, b):
    """`segloss/iou_multi.py:4-7` on boolean masks."""
    inter = float((a & b).sum())
    union = float(a.sum()) + float(b.sum()) - inter
    return (inter + 1e-6) / (union + 1e-6)


def ch_iou(y_true, y_pred):
    """`segloss/iou_multi.py:51-65` (numpy label arrays): mean IoU over the non-zero classes present in y_true."""
    import numpy as np
    if y_true.sum() == 0:
        return 1 if y_pred.sum() == 0 else 0
    res = [_iou_bool(y_true == k, y_pred == k) for k in set(y_true.flatten().tolist()) if k != 0]
    return float(np.mean(res))


def isi_iou(y_true, y_pred, problem_type: str = "instruments"):
    """`segloss/iou_multi.py:67-88`: classes 1..type_number-1 that occur in y_true or y_pred."""
    import numpy as np
    n = {"binary": 2, "parts": 4, "instruments": 8}[problem_type]
    if y_true.sum() == 0:
        return 1 if y_pred.sum() == 0 else 0
    res = [_iou_bool(y_true == k, y_pred == k) for k in range(1, n)
           if (y_true == k).sum() != 0 or (y_pred == k).sum() != 0]
    return float(np.mean(res))


def cross_entropy_nd(logits, target, weight=None):
    """`segloss/ND_Crossentropy.py:11-32`; with class weights it is the val
    loss of `train.py:616-617`."""
    C = logits.shape[1]
    lg = logits.permute(0, *range(2, logits.dim()), 1).reshape(-1, C)
    return F.cross_entropy(lg, target.reshape(-1).long(), weight=weight)


def dc_and_ce_loss(logits, target, target_onehot):
    """`segloss/dice_loss.py:445-459`."""
    return cross_entropy_nd(logits, target) + soft_dice_loss(logits, target_onehot)


def iou_loss(preds, labels, smooth: float = 1e-6, num_classes: int = 8):
    """`segloss/iou_multi.py:9-49`."""
    oh = F.one_hot(labels.long(), num_classes).permute(0, 3, 1, 2)
    p = F.softmax(preds, dim=1)
    loss = 0.0
    for c in range(num_classes):
        inter = torch.sum(p[:, c] * oh[:, c], dim=[1, 2])
        union = torch.sum(p[:, c], dim=[1, 2]) + torch.sum(oh[:, c], dim=[1, 2]) - inter
        loss = loss + (1 - (inter + smooth) / (union + smooth)).mean()
    return loss / num_classes


# ----------------------------------------------------------------------------
# The training step as `train.py:268-436` assembles it
# ----------------------------------------------------------------------------
def assemble_decoder_input(x_tokens, c4_tokens, vit_tokens, hw: Tuple[int, int], c4_hw: Tuple[int, int]):
    """`train.py:389-406`: tokens -> NCHW maps, zero-pad c4 to the ViT grid,
    channel concat (adapter stream, c4, pass-A last layer)."""
    B, _, D = x_tokens.shape
    h, w = hw
    a = x_tokens.transpose(1, 2).reshape(B, D, h, w)
    v = vit_tokens.transpose(1, 2).reshape(B, D, h, w)
    c = c4_tokens.transpose(1, 2).reshape(B, D, c4_hw[0], c4_hw[1])
    dy, dx = h - c4_hw[0], w - c4_hw[1]
    c = F.pad(c, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return torch.cat((a, c, v), dim=1)


def adapter_forward(img, vit_sd: SD, enc_sd: SD, cavit_sd: SD, cacnn_sd: SD, num_heads: int,
                    patch: int = 14, n_last: int = 4, msda_heads: int = 8, taps: Optional[dict] = None,
                    update_bn: bool = False):
    """`train.py:275-406` up to ``output_last_cat`` (no grad bookkeeping here:
    the caller decides what is detached)."""
    B, _, H, W = img.shape
    c1, c2, c3, c4, shapes = feature_encoder(img, enc_sd, update_bn=update_bn)
    d1, d2 = deform_inputs(H, W, patch, shapes)
    c = torch.cat([c2, c3, c4], dim=1)  # level_embed is a fresh zero tensor each step: train.py:277-282
    feats = get_intermediate_layers(img, vit_sd, num_heads, n=n_last, patch=patch)
    outs = [f for f, _ in feats]  # [last_4, last_3, last_2, last]
    depth = vit_depth(vit_sd)
    x = patch_embed(img, vit_sd, patch)  # no cls, no pos-embed: train.py:300
    for i in range(depth - (n_last - 1)):
        x = block(x, vit_sd, f"blocks.{i}", num_heads)
    if taps is not None:
        taps.update(c2=c2, c3=c3, c4=c4, feats=outs, x_b0=x)
    n_lvl = len(shapes)
    for s in range(n_last):
        if s > 0:
            x = block(x, vit_sd, f"blocks.{depth - (n_last - 1) + s - 1}", num_heads)
        x = cavit(x, d1[0], c, d1[1], cavit_sd, n_heads=msda_heads, n_levels=n_lvl)
        c = cacnn(c, d2[0], x, d2[1], shapes, cacnn_sd, n_heads=msda_heads, n_levels=1)
        x = x + outs[s]
        if taps is not None:
            taps[f"x_stage{s}"] = x
            taps[f"c_stage{s}"] = c
    cat = assemble_decoder_input(x, c4, outs[-1], (H // patch, W // patch), shapes[2])
    if taps is not None:
        taps["output_last_cat"] = cat
    return cat


def mla_forward(img, vit_sd: SD, enc_sd: SD, cavit_sd: SD, cacnn_sd: SD, num_heads: int, patch: int = 14,
                msda_heads: int = 8, update_bn: bool = False):
    """`train_mla.py:266-383`: the four MLA inputs as NCHW maps (output_last, output_last_2, _3, _4).
    Order per stage is block -> CACNN -> CAViT; ``blocks[-2:-1]`` is evaluated twice and ``blocks[-1]`` never
    (`train_mla.py:318,340`); pass A runs last and only its last layer is added to the final map."""
    B, _, H, W = img.shape
    c1, c2, c3, c4, shapes = feature_encoder(img, enc_sd, update_bn=update_bn)
    d1, d2 = deform_inputs(H, W, patch, shapes)
    c = torch.cat([c2, c3, c4], dim=1)
    depth = vit_depth(vit_sd)
    x = patch_embed(img, vit_sd, patch)
    for i in range(depth - 3):
        x = block(x, vit_sd, f"blocks.{i}", num_heads)
    n_lvl = len(shapes)
    x = cavit(x, d1[0], c, d1[1], cavit_sd, n_heads=msda_heads, n_levels=n_lvl)
    outs = [x]
    for bi in (depth - 3, depth - 2, depth - 2):
        x = block(x, vit_sd, f"blocks.{bi}", num_heads)
        c = cacnn(c, d2[0], x, d2[1], shapes, cacnn_sd, n_heads=msda_heads, n_levels=1)
        x = cavit(x, d1[0], c, d1[1], cavit_sd, n_heads=msda_heads, n_levels=n_lvl)
        outs.append(x)
    vit_last = get_intermediate_layers(img, vit_sd, num_heads, n=4, patch=patch)[-1][0]
    last = vit_last + outs[3]
    h, w = H // patch, W // patch
    D = x.shape[-1]
    return [t.transpose(1, 2).reshape(B, D, h, w) for t in (last, outs[2], outs[1], outs[0])]


def train_step_loss_mla(maps, target, dec_sd: SD, num_classes: int = 2, loss: str = "dice", taps: Optional[dict] = None,
                        update_bn: bool = False):
    """`train_mla.py:385-395` (DC) / `train_multi_class.py:391-393` (iou_loss): decoder (resizes to the image
    size itself, `decoders.py:88`) -> softmax -> loss (which applies softmax again)."""
    H = target.shape[-1]
    out = decoder_mla(*maps, sd=dec_sd, img_size=H, update_bn=update_bn)
    prob = torch.softmax(out, 1)
    if loss == "dice":
        val = dc_loss(prob, one_hot(target, num_classes))
    else:
        val = iou_loss(prob, target, num_classes=num_classes)
    if taps is not None:
        taps.update(out=out, loss=val)
    return val


def train_step_loss(cat, target, dec_sd: SD, num_classes: int = 2, taps: Optional[dict] = None,
                    update_bn: bool = False):
    """`train.py:421-428`: decoder -> bilinear resize to (H,W) -> softmax -> DC
    (which applies softmax again: SURVEY.md fact 5)."""
    H, W = target.shape[-2:]
    logits = feature_decoder(cat, dec_sd, update_bn=update_bn)
    out = F.interpolate(logits, size=(H, W), mode="bilinear")
    prob = torch.softmax(out, 1)
    loss = dc_loss(prob, one_hot(target, num_classes))
    if taps is not None:
        taps.update(logits=logits, logits_resized=out, loss=loss)
    return loss


def validate_metrics(cat, target, dec_sd: SD, num_classes: int = 2):
    """`train.py:612-642`: decoder in eval mode -> resize -> weighted CE ([0.1, 10]), dice = 1 - DC(logits),
    pixel accuracy."""
    H, W = target.shape[-2:]
    out = F.interpolate(feature_decoder(cat, dec_sd, training=False), size=(H, W), mode="bilinear")
    wt = torch.tensor([0.1, 10.0]) if num_classes == 2 else None
    loss = F.cross_entropy(out, target.long(), weight=wt)
    dice = 1 - dc_loss(out, one_hot(target, num_classes))
    acc = (out.argmax(1) == target).float().mean()
    return loss, dice, acc


def sgd_momentum_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor],
                      bufs: Dict[str, torch.Tensor], lr: float, momentum: float = 0.99,
                      weight_decay: float = 3e-5):
    """torch.optim.SGD semantics as configured at `train.py:178-191`
    (dampening 0, no Nesterov): g += wd*p ; buf = g (first) | m*buf + g ; p -= lr*buf."""
    for k, p in params.items():
        g = grads[k] + weight_decay * p
        if k not in bufs:
            bufs[k] = g.clone()
        else:
            bufs[k].mul_(momentum).add_(g)
        p.sub_(lr * bufs[k])
