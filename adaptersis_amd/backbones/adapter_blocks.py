"""HIP-backed adapter blocks — API / ``state_dict`` mirror of `backbones/adapter_blocks.py`.

``CAViT`` (CNN -> ViT injector, `:149-183`):  q + gamma * MSDA(LN(q), LN(feat))
``CACNN`` (ViT -> CNN extractor, `:102-147`): c + MSDA(LN(c), LN(x)) ; c + ConvFFN(LN(c))
LayerNorms write 16-bit GEMM operands directly; the residual / gamma are fused into the
output_proj / fc2 GEMM epilogues; the ConvFFN depthwise conv + GELU is one kernel.

Generalisation: the reference hard-codes n = 18*18 token grids
(`adapter_blocks.py:71`); here the three grids follow from (H, W) exactly as the reference
slices them — (2H+1, 2W+1), (H, W), (H//2, W//2) — or can be passed explicitly via ``grids``.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from .. import config, ops
from ..dinov2.layers.blocks import _Packed
from .ops.modules import MSDeformAttn
from .ops.modules.ms_deform_attn import prepare_msda_geometry


def get_reference_points(spatial_shapes, device):
    """`adapter_blocks.py:9-22` (host-side constant: cell centres in (x, y) order)."""
    reference_points_list = []
    for lvl, (H_, W_) in enumerate(spatial_shapes):
        ref_y, ref_x = torch.meshgrid(
            torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
            torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
        ref_y = ref_y.reshape(-1)[None] / H_
        ref_x = ref_x.reshape(-1)[None] / W_
        reference_points_list.append(torch.stack((ref_x, ref_y), -1))
    reference_points = torch.cat(reference_points_list, 1)
    return reference_points[:, :, None]


def deform_inputs(x, patch_size, cnn_shapes=None):
    """`adapter_blocks.py:24-38`.  ``cnn_shapes`` (the encoder's real c2/c3/c4 grid sizes) replaces the
    reference's h//8, h//16, h//32 guess, which only matches the encoder at 588."""
    bs, c, h, w = x.shape
    if cnn_shapes is None:
        cnn_shapes = [(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)]
    cnn_shapes = [tuple(int(v) for v in s) for s in cnn_shapes]
    spatial_shapes = torch.as_tensor(cnn_shapes, dtype=torch.long, device=x.device)
    level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
    reference_points = get_reference_points([(h // patch_size, w // patch_size)], x.device)
    deform_inputs1 = [reference_points, spatial_shapes, level_start_index]
    spatial_shapes = torch.as_tensor([(h // patch_size, w // patch_size)], dtype=torch.long, device=x.device)
    level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
    reference_points = get_reference_points(cnn_shapes, x.device)
    deform_inputs2 = [reference_points, spatial_shapes, level_start_index]
    return deform_inputs1, deform_inputs2


class DWConv(nn.Module):
    """`adapter_blocks.py:62-80` parameter container (depthwise 3x3, bias); compute is fused in ConvFFN."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class ConvFFN(_Packed):
    """`adapter_blocks.py:82-100`."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        if drop:
            raise ValueError("dropout is 0 on the AdapterSIS path (train.py:108)")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if hidden_features % 8:
            raise ValueError("ConvFFN hidden width must be a multiple of 8")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward16(self, n16, res, B, Ntok, grids):
        """n16 16-bit [B*Ntok, D] (normalised) -> fp32 res + fc2(gelu(dwconv(fc1 n)))."""
        dt = config.operand_dtype
        dev = n16.device
        hid = self.fc1.out_features
        h = ops.gemm(n16, self._w16("fc1", self.fc1.weight), out_f32=True, bias_n=self._f32("fc1_b", self.fc1.bias))
        key = ("grids", tuple(grids), dev)
        if self._cache.get("gkey") != key:
            sizes = [a * b for a, b in grids]
            starts = [0]
            for s in sizes[:-1]:
                starts.append(starts[-1] + s)
            self._cache["gshapes"] = torch.tensor(grids, dtype=torch.int32, device=dev)
            self._cache["gstarts"] = torch.tensor(starts, dtype=torch.int32, device=dev)
            self._cache["gkey"] = key
        w9 = self._f32_fn("dw9", self.dwconv.dwconv.weight, lambda p: p.reshape(hid, 9).t().contiguous().float())
        g = ops.dwconv_gelu(h.view(B, Ntok, hid), w9, self._f32("dwb", self.dwconv.dwconv.bias),
                            self._cache["gshapes"], self._cache["gstarts"], dt)
        return ops.gemm(g.view(B * Ntok, hid), self._w16("fc2", self.fc2.weight), out_f32=True,
                        bias_n=self._f32("fc2_b", self.fc2.bias), res=res)

    def _grid_tensors(self, grids, dev):
        key = ("grids", tuple(grids), dev)
        if self._cache.get("gkey") != key:
            sizes = [a * b for a, b in grids]
            starts = [0]
            for s in sizes[:-1]:
                starts.append(starts[-1] + s)
            self._cache["gshapes"] = torch.tensor(grids, dtype=torch.int32, device=dev)
            self._cache["gstarts"] = torch.tensor(starts, dtype=torch.int32, device=dev)
            self._cache["gkey"] = key
        return self._cache["gshapes"], self._cache["gstarts"]

    def forward16_train(self, n16, res, B, Ntok, grids):
        """``forward16`` keeping (n16, fc1 output, GELU output) for ``backward16``."""
        dt = config.operand_dtype
        hid = self.fc1.out_features
        h = ops.gemm(n16, self._w16("fc1", self.fc1.weight), out_f32=True, bias_n=self._f32("fc1_b", self.fc1.bias))
        gsh, gst = self._grid_tensors(grids, n16.device)
        w9 = self._f32_fn("dw9", self.dwconv.dwconv.weight, lambda p: p.reshape(hid, 9).t().contiguous().float())
        g = ops.dwconv_gelu(h.view(B, Ntok, hid), w9, self._f32("dwb", self.dwconv.dwconv.bias), gsh, gst, dt)
        out = ops.gemm(g.view(B * Ntok, hid), self._w16("fc2", self.fc2.weight), out_f32=True,
                       bias_n=self._f32("fc2_b", self.fc2.bias), res=res)
        return out, (n16, h, g, B, Ntok, tuple(grids))

    def backward16(self, saved, dout, inv_scale, grads, prefix):
        """dout fp32 [B*Ntok, D] (scaled) -> d n16-input fp32 [B*Ntok, D]; grads of fc1, dwconv, fc2 (overwritten)."""
        n16, h, g, B, Ntok, grids = saved
        dt = config.operand_dtype
        hid = self.fc1.out_features
        pre = prefix + "."
        d16, cs = ops.cast_colsum(dout, dt)
        self._linear_bwd(pre + "fc2", self.fc2, None, None, d16, cs, g.view(B * Ntok, hid), inv_scale, grads)
        dg = ops.gemm(d16, self._wT16("fc2T", self.fc2.weight), out_f32=True)
        gsh, gst = self._grid_tensors(grids, dout.device)
        w9 = self._f32_fn("dw9", self.dwconv.dwconv.weight, lambda p: p.reshape(hid, 9).t().contiguous().float())
        dh16, part = ops.dwconv_gelu_bwd(h.view(B, Ntok, hid), w9, self._f32("dwb", self.dwconv.dwconv.bias), gsh, gst,
                                         dg.view(B, Ntok, hid), dt)
        red = ops.reduce_rows(part.view(part.shape[0], 10 * hid), inv_scale).view(10, hid)
        grads[pre + "dwconv.dwconv.weight"].view(hid, 9).copy_(red[:9].t())
        grads[pre + "dwconv.dwconv.bias"].copy_(red[9])
        dh2 = dh16.view(B * Ntok, hid)
        self._linear_bwd(pre + "fc1", self.fc1, None, None, dh2, ops.colsum(dh2), n16, inv_scale, grads)
        return ops.gemm(dh2, self._wT16("fc1T", self.fc1.weight), out_f32=True)

    def _f32_fn(self, key, param, fn):
        from ..dinov2.layers.blocks import _pack
        return _pack(self._cache, key, param, fn)


class _AdapterBase(_Packed):
    def _ln16(self, name, x2d):
        ln = getattr(self, name)
        return ops.layernorm(x2d, self._f32(name + "_w", ln.weight), self._f32(name + "_b", ln.bias), ln.eps,
                             config.operand_dtype)

    def _ln_bwd(self, name, dy, x2d, inv_scale, grads, prefix, res=None):
        """LayerNorm backward: dx (+ res) and the weight / bias gradients into ``grads``."""
        ln = getattr(self, name)
        D = x2d.shape[1]
        dx, part = ops.layernorm_bwd(dy, x2d, self._f32(name + "_w", ln.weight), ln.eps, res=res)
        red = ops.reduce_rows(part.view(part.shape[0], 2 * D), inv_scale)
        grads[f"{prefix}{name}.weight"].copy_(red[:D]); grads[f"{prefix}{name}.bias"].copy_(red[D:])
        return dx

    @staticmethod
    def _flat(x):
        B, N, D = x.shape
        x2 = x.reshape(B * N, D)
        if x2.dtype != torch.float32 or not x2.is_contiguous():
            x2 = x2.float().contiguous()
        return x2, B, N, D


class CACNN(_AdapterBase):
    def __init__(self, dim, num_heads=6, n_points=4, n_levels=1, deform_ratio=1.0, with_cffn=True, cffn_ratio=0.25,
                 drop=0.0, drop_path=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6), with_cp=False):
        super().__init__()
        if drop or drop_path:
            raise ValueError("drop / drop_path are 0 on the AdapterSIS path (train.py:108-109)")
        self.query_norm = norm_layer(dim)
        self.feat_norm = norm_layer(dim)
        self.attn = MSDeformAttn(d_model=dim, n_levels=n_levels, n_heads=num_heads, n_points=n_points, ratio=deform_ratio)
        self.with_cffn = with_cffn
        self.with_cp = with_cp
        if with_cffn:
            self.ffn = ConvFFN(in_features=dim, hidden_features=int(dim * cffn_ratio), drop=drop)
            self.ffn_norm = norm_layer(dim)
            self.drop_path = nn.Identity()

    def forward(self, query, reference_points, feat, spatial_shapes, level_start_index, H, W, grids=None):
        q2, B, Lq, D = self._flat(query)
        f2, _, Lin, _ = self._flat(feat)
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == Lin
        ref, shapes_i32, starts_i32 = prepare_msda_geometry(reference_points, spatial_shapes, level_start_index, Lq,
                                                            query.device)
        out = self.attn.forward16(self._ln16("query_norm", q2), self._ln16("feat_norm", f2), ref, shapes_i32,
                                  starts_i32, B, Lq, Lin, res=q2)
        if self.with_cffn:
            if grids is None:  # the reference's slicing (adapter_blocks.py:72-74) for its (H, W) = (h//16, w//16)
                grids = [(H * 2 + 1, W * 2 + 1), (H, W), (H // 2, W // 2)]
            grids = [tuple(int(v) for v in g) for g in grids]
            if sum(a * b for a, b in grids) != Lq:
                raise ValueError(f"CACNN: token grids {grids} do not cover {Lq} tokens; pass grids= explicitly")
            out = self.ffn.forward16(self._ln16("ffn_norm", out), out, B, Lq, grids)
        return out.view(B, Lq, D)


    # ---- `train_adapters` mode --------------------------------------------------------------------------------------
    def forward16_train(self, c2, x2, g, B, Lq, Lin, grids):
        """c2 fp32 [B*Lq, D] (pyramid tokens = query), x2 fp32 [B*Lin, D] (ViT tokens = feat), g = engine geometry dict
        (ref2 / shapes2 / starts2) -> (out fp32 [B*Lq, D], saved)."""
        qn, fn = self._ln16("query_norm", c2), self._ln16("feat_norm", x2)
        out1, s_attn = self.attn.forward16_train(qn, fn, g["ref2"], g["shapes2"], g["starts2"], B, Lq, Lin, res=c2)
        s_ffn = None
        out = out1
        if self.with_cffn:
            out, s_ffn = self.ffn.forward16_train(self._ln16("ffn_norm", out1), out1, B, Lq, grids)
        return out, (c2, x2, out1, s_attn, s_ffn)

    def backward16(self, saved, dout, inv_scale, grads, prefix=""):
        """dout fp32 [B*Lq, D] -> (d c2, d x2) fp32; all parameter gradients of the module into ``grads``."""
        c2, x2, out1, s_attn, s_ffn = saved
        pre = prefix + "." if prefix else ""
        d1 = dout
        if self.with_cffn:
            dn = self.ffn.backward16(s_ffn, dout, inv_scale, grads, pre + "ffn")
            d1 = self._ln_bwd("ffn_norm", dn, out1, inv_scale, grads, pre, res=dout)
        dq, dfeat = self.attn.backward16(s_attn, d1, inv_scale, grads, pre + "attn")
        dc = self._ln_bwd("query_norm", dq, c2, inv_scale, grads, pre, res=d1)
        dx = self._ln_bwd("feat_norm", dfeat, x2, inv_scale, grads, pre)
        return dc, dx


class CAViT(_AdapterBase):
    def __init__(self, dim, num_heads=6, n_points=4, n_levels=1, deform_ratio=1.0,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=0.0, with_cp=False):
        super().__init__()
        self.with_cp = with_cp
        self.query_norm = norm_layer(dim)
        self.feat_norm = norm_layer(dim)
        self.attn = MSDeformAttn(d_model=dim, n_levels=n_levels, n_heads=num_heads, n_points=n_points, ratio=deform_ratio)
        self.gamma = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)

    def forward(self, query, reference_points, feat, spatial_shapes, level_start_index):
        q2, B, Lq, D = self._flat(query)
        f2, _, Lin, _ = self._flat(feat)
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == Lin
        ref, shapes_i32, starts_i32 = prepare_msda_geometry(reference_points, spatial_shapes, level_start_index, Lq,
                                                            query.device)
        out = self.attn.forward16(self._ln16("query_norm", q2), self._ln16("feat_norm", f2), ref, shapes_i32,
                                  starts_i32, B, Lq, Lin, res=q2, scale_n=self._f32("gamma", self.gamma))
        return out.view(B, Lq, D)

    # ---- `train_adapters` mode --------------------------------------------------------------------------------------
    def forward16_train(self, x2, c2, g, B, Lq, Lin):
        """x2 fp32 [B*Lq, D] (ViT tokens = query), c2 fp32 [B*Lin, D] (pyramid tokens = feat) -> (out, saved)."""
        qn, fn = self._ln16("query_norm", x2), self._ln16("feat_norm", c2)
        out, s_attn = self.attn.forward16_train(qn, fn, g["ref1"], g["shapes1"], g["starts1"], B, Lq, Lin, res=x2,
                                                scale_n=self._f32("gamma", self.gamma))
        return out, (x2, c2, s_attn)

    def backward16(self, saved, dout, inv_scale, grads, prefix=""):
        """dout fp32 [B*Lq, D] -> (d x2, d c2) fp32; parameter gradients (incl. ``gamma``) into ``grads``."""
        x2, c2, s_attn = saved
        pre = prefix + "." if prefix else ""
        dq, dfeat = self.attn.backward16(s_attn, dout, inv_scale, grads, pre + "attn", gamma=self.gamma,
                                         gamma_name=pre + "gamma")
        dx = self._ln_bwd("query_norm", dq, x2, inv_scale, grads, pre, res=dout)
        dc = self._ln_bwd("feat_norm", dfeat, c2, inv_scale, grads, pre)
        return dx, dc
