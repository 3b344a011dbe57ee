"""HIP-backed ``MSDeformAttn`` — API / ``state_dict`` mirror of
`backbones/ops/modules/ms_deform_attn.py:63-185`.

value_proj, (sampling_offsets | attention_weights) and output_proj are MFMA GEMMs; the softmax
over L*P, the sampling-location arithmetic and the bilinear gather are one HIP kernel
(``asis_msda_fwd``).  Unlike the reference, whose ``MSDeformAttnFunction`` has no backward
(`ms_deform_attn.py:17-30`), this module is forward-only *by declaration*: gradients for the
trainable-adapter mode are a later row of SURVEY.md §8.
"""
from __future__ import annotations

import math
import warnings
from typing import Optional

import torch
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from .... import config, ops
from ....dinov2.layers.blocks import _Packed, _pack


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(_Packed):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, ratio=1.0):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if ratio != 1.0:
            raise ValueError("only ratio=1.0 is used on the AdapterSIS path (train.py:92,107)")
        _d_per_head = d_model // n_heads
        if _d_per_head % 8:
            raise ValueError(f"head dim {_d_per_head} must be a multiple of 8 (16-byte channel chunks)")
        if n_levels * n_points > 16:
            raise ValueError("n_levels*n_points must be <= 16")
        if not _is_power_of_2(_d_per_head):
            warnings.warn("You'd better set d_model in MSDeformAttn to make the dimension of each attention head a "
                          "power of 2 which is more efficient in our CUDA implementation.")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points, self.ratio = d_model, n_levels, n_heads, n_points, ratio
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, int(d_model * ratio))
        self.output_proj = nn.Linear(int(d_model * ratio), d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """`ms_deform_attn.py:99-118`."""
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(
            1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    # ---- fused path used by CAViT / CACNN: 16-bit, already-normalised inputs ------------------
    def forward16(self, q16, feat16, ref, shapes_i32, starts_i32, B, Lq, Lin, *, res=None, scale_n=None):
        """q16 [B*Lq, D], feat16 [B*Lin, D] (16-bit) -> fp32 [B*Lq, D] = res + scale_n * output_proj(msda)."""
        dt = config.operand_dtype
        M, L, P = self.n_heads, self.n_levels, self.n_points

        srcs = (self.sampling_offsets.weight, self.attention_weights.weight, self.sampling_offsets.bias,
                self.attention_weights.bias)
        tag = tuple((t.data_ptr(), t._version, getattr(t, "_asis_gen", 0)) for t in srcs) + (dt,)
        if self._cache.get("oa_tag") != tag:  # offsets and attention logits come from ONE GEMM: [M*L*P*3, D]
            with torch.no_grad():
                w = torch.cat([srcs[0].detach(), srcs[1].detach()], 0).float().contiguous()
                self._cache["w_oa"] = ops.cast_pad(w, dtype=dt)
                self._cache["b_oa"] = torch.cat([srcs[2].detach(), srcs[3].detach()]).float().contiguous()
            self._cache["oa_tag"] = tag
        w_oa = self._cache["w_oa"]
        if config.precise_adapters_on():
            # 16 significant bits where tests/precision_probe.py puts the adapters' error in bf16 (weights 9.2e-4 and the sampled
            # rows 6.2e-4 of 1.43e-3 on the logits): the three weights as hi + lo halves, the sampled rows as a split A operand
            if self._cache.get("oa_lo_tag") != tag:
                with torch.no_grad():
                    w = torch.cat([srcs[0].detach(), srcs[1].detach()], 0).float().contiguous()
                    self._cache["w_oa_lo"] = ops.cast_pad(w, dtype=dt, part=1)
                self._cache["oa_lo_tag"] = tag
            lo = lambda key, p_: _pack(self._cache, key + ".lo", p_, lambda t: ops.cast_pad(t.reshape(t.shape[0], -1).contiguous().float(), dtype=dt, part=1))
            value = ops.gemm(feat16, self._w16("wv", self.value_proj.weight), bias_n=self._f32("bv", self.value_proj.bias),
                             b_lo=lo("wv", self.value_proj.weight))
            offaw = ops.gemm(q16, w_oa, out_f32=True, bias_n=self._cache["b_oa"], b_lo=self._cache["w_oa_lo"])
            samp, samp_lo = ops.msda_fwd(value.view(B, Lin, self.d_model), offaw, ref, shapes_i32, starts_i32, B, Lq, M, L, P, split=True)
            return ops.gemm(samp, self._w16("wo", self.output_proj.weight), out_f32=True, bias_n=self._f32("bo", self.output_proj.bias),
                            scale_n=scale_n, res=res, a_lo=samp_lo, b_lo=lo("wo", self.output_proj.weight))
        value = ops.gemm(feat16, self._w16("wv", self.value_proj.weight), bias_n=self._f32("bv", self.value_proj.bias))
        offaw = ops.gemm(q16, w_oa, out_f32=True, bias_n=self._cache["b_oa"])
        samp = ops.msda_fwd(value.view(B, Lin, self.d_model), offaw, ref, shapes_i32, starts_i32, B, Lq, M, L, P)
        return ops.gemm(samp, self._w16("wo", self.output_proj.weight), out_f32=True,
                        bias_n=self._f32("bo", self.output_proj.bias), scale_n=scale_n, res=res)

    # ---- training path (`train_adapters` mode): forward that keeps what the backward needs ------------------------
    def forward16_train(self, q16, feat16, ref, shapes_i32, starts_i32, B, Lq, Lin, *, res=None, scale_n=None):
        """Same arithmetic as ``forward16`` -> (out fp32 [B*Lq, D], saved)."""
        M, L, P = self.n_heads, self.n_levels, self.n_points
        srcs = (self.sampling_offsets.weight, self.attention_weights.weight, self.sampling_offsets.bias,
                self.attention_weights.bias)
        tag = tuple((t.data_ptr(), t._version, getattr(t, "_asis_gen", 0)) for t in srcs) + (config.operand_dtype,)
        if self._cache.get("oa_tag") != tag:
            with torch.no_grad():
                w = torch.cat([srcs[0].detach(), srcs[1].detach()], 0).float().contiguous()
                self._cache["w_oa"] = ops.cast_pad(w, dtype=config.operand_dtype)
                self._cache["b_oa"] = torch.cat([srcs[2].detach(), srcs[3].detach()]).float().contiguous()
            self._cache["oa_tag"] = tag
        value = ops.gemm(feat16, self._w16("wv", self.value_proj.weight), bias_n=self._f32("bv", self.value_proj.bias))
        offaw = ops.gemm(q16, self._cache["w_oa"], out_f32=True, bias_n=self._cache["b_oa"])
        samp = ops.msda_fwd(value.view(B, Lin, self.d_model), offaw, ref, shapes_i32, starts_i32, B, Lq, M, L, P)
        out = ops.gemm(samp, self._w16("wo", self.output_proj.weight), out_f32=True,
                       bias_n=self._f32("bo", self.output_proj.bias), scale_n=scale_n, res=res)
        return out, (q16, feat16, value, offaw, samp, ref, shapes_i32, starts_i32, B, Lq, Lin)

    def backward16(self, saved, dout: torch.Tensor, inv_scale: float, grads: dict, prefix: str,
                   gamma: Optional[torch.nn.Parameter] = None, gamma_name: Optional[str] = None):
        """dout fp32 [B*Lq, D] = loss_scale * dL/d(out) -> (d q16-input fp32 [B*Lq, D], d feat16-input fp32 [B*Lin, D]);
        parameter gradients (value_proj, sampling_offsets, attention_weights, output_proj, and the caller's ``gamma``
        when the output projection carries a per-channel scale) are written into ``grads`` (overwritten)."""
        q16, feat16, value, offaw, samp, ref, shapes_i32, starts_i32, B, Lq, Lin = saved
        dt = config.operand_dtype
        D = self.d_model
        M, L, P = self.n_heads, self.n_levels, self.n_points
        n_off, n_aw = M * L * P * 2, M * L * P
        pre = prefix + "."
        # output_proj (with the caller's LayerScale-like gamma folded in)
        d16, cs = ops.cast_colsum(dout, dt)
        self._linear_bwd(pre + "output_proj", self.output_proj, gamma, gamma_name, d16, cs, samp, inv_scale, grads)
        dsamp = ops.gemm(d16, self._wT16("woT", self.output_proj.weight, gamma), out_f32=True)
        # sampling core
        dvalue, doffaw = ops.msda_bwd(value.view(B, Lin, D), offaw, ref, shapes_i32, starts_i32, dsamp, B, Lq, M, L, P)
        # offsets | attention logits = q16 W_oa^T + b_oa
        doa16, cs_oa = ops.cast_colsum(doffaw, dt)
        R = q16.shape[0]
        G = ops.wgrad(doa16.view(1, R, 1, n_off + n_aw), q16.view(1, R, 1, D), n_off + n_aw, 1, 1, 1, 0, inv_scale)
        G = G.view(n_off + n_aw, D)
        grads[pre + "sampling_offsets.weight"].copy_(G[:n_off]); grads[pre + "attention_weights.weight"].copy_(G[n_off:])
        cso = ops.reduce_rows(cs_oa, inv_scale)
        grads[pre + "sampling_offsets.bias"].copy_(cso[:n_off]); grads[pre + "attention_weights.bias"].copy_(cso[n_off:])
        w_oaT = self._pack2("w_oaT", self.sampling_offsets.weight, self.attention_weights.weight, lambda: ops.cast_pad(
            torch.cat([self.sampling_offsets.weight.detach(), self.attention_weights.weight.detach()], 0).float().t().contiguous(),
            dtype=dt))
        dq = ops.gemm(doa16, w_oaT, out_f32=True)
        # value_proj
        dv16, cs_v = ops.cast_colsum(dvalue.view(B * Lin, D), dt)
        self._linear_bwd(pre + "value_proj", self.value_proj, None, None, dv16, cs_v, feat16, inv_scale, grads)
        dfeat = ops.gemm(dv16, self._wT16("wvT", self.value_proj.weight), out_f32=True)
        return dq, dfeat

    # ---- reference-shaped entry point ---------------------------------------------------------------
    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """`ms_deform_attn.py:120-185`: fp32 (N, Lq, C) / (N, Lin, C) tensors in, fp32 (N, Lq, C) out."""
        if input_padding_mask is not None:
            raise ValueError("input_padding_mask is always None on the AdapterSIS path (adapter_blocks.py:133,173)")
        N, Len_q, C = query.shape
        _, Len_in, _ = input_flatten.shape
        assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == Len_in
        ref, shapes_i32, starts_i32 = prepare_msda_geometry(reference_points, input_spatial_shapes,
                                                            input_level_start_index, Len_q, query.device)
        dt = config.operand_dtype
        q16 = ops.cast_pad(query.reshape(N * Len_q, C).float().contiguous(), dtype=dt)
        f16 = ops.cast_pad(input_flatten.reshape(N * Len_in, C).float().contiguous(), dtype=dt)
        return self.forward16(q16, f16, ref, shapes_i32, starts_i32, N, Len_q, Len_in).view(N, Len_q, C)


def prepare_msda_geometry(reference_points, spatial_shapes, level_start_index, Lq, device):
    """(1|N, Lq, 1|L, 2) reference points -> fp32 [Lq, 2]; int32 shapes / starts on the device."""
    if reference_points.shape[-1] != 2:
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]) if reference_points.shape[-1] != 4 else
            "reference boxes (last dim 4) are not used on the AdapterSIS path")
    rp = reference_points
    if rp.dim() != 4 or rp.shape[1] != Lq:
        raise ValueError(f"reference_points must be (N, {Lq}, n_levels, 2), got {tuple(rp.shape)}")
    if rp.shape[0] != 1 and not bool((rp == rp[:1]).all()):
        raise ValueError("per-sample reference points are not used on the AdapterSIS path (adapter_blocks.py:9-22)")
    if rp.shape[2] != 1 and not bool((rp == rp[:, :, :1]).all()):
        raise ValueError("per-level reference points are not used on the AdapterSIS path (adapter_blocks.py:21)")
    ref = rp[0, :, 0, :].to(device=device, dtype=torch.float32).contiguous()
    shapes_i32 = spatial_shapes.to(device=device, dtype=torch.int32).contiguous()
    starts_i32 = level_start_index.to(device=device, dtype=torch.int32).contiguous()
    return ref, shapes_i32, starts_i32
