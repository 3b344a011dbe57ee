"""HIP-backed DINOv2 layers: same class names, ctor arguments and ``state_dict`` keys as
`dinov2/layers/{patch_embed,attention,mlp,swiglu_ffn,layer_scale,block}.py`; forward runs the
gfx950 kernels of libasis_hip.so (no eager fallback).

Data flow of one eval-mode block (`block.py:111-113`), residual stream fp32 [B*N, D]:

    LN1 (fp32 stats) -> 16-bit xn
    QK  GEMM  xn W_qk^T + b           -> 16-bit [B*N, 2D]
    V^T GEMM  W_v xn_b^T + b (per image) -> 16-bit [B, D, ldvt]   (already transposed for PV)
    fused softmax attention            -> 16-bit o [B*N, D]
    proj GEMM, epilogue x + ls1*(.+b)  -> fp32 residual
    LN2 -> fc1 GEMM + bias + erf-GELU (16-bit) -> fc2 GEMM, epilogue x + ls2*(.+b) -> fp32
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Callable, Optional, Tuple, Union

import torch
from torch import nn

from ... import config, ops


_VT_STREAMS = {}


def _vt_side_stream(device) -> "torch.cuda.Stream":
    """one side HIP stream per device for the V^T GEMMs of the stacked attention (``Attention.attend_rows``)"""
    key = (device.type, device.index)
    if key not in _VT_STREAMS:
        _VT_STREAMS[key] = torch.cuda.Stream(device=device)
    return _VT_STREAMS[key]


_PRECISE_FOLD_Q = os.environ.get("ASIS_PRECISE_FOLD_Q", "1") not in ("0", "")   # lab: the precise_level-2 path with an unfolded q (round <= 4)
_ATTN_MX_OUT = os.environ.get("ASIS_ATTN_MX_OUT", "1") not in ("0", "")          # lab: 0 = (o, o_lo) + absmax16 + mx_from_pair
_LS_POW2_REFRESH = 64   # Block._ls_pow2: parameter changes between two reads of max|gamma|


def _pack(cache: dict, key: str, param: torch.Tensor, fn: Callable[[torch.Tensor], torch.Tensor]):
    """Cache a derived (16-bit / re-laid-out) copy of a parameter until the parameter changes."""
    tag = (param.data_ptr(), param._version, getattr(param, "_asis_gen", 0), config.operand_dtype, param.device)
    hit = cache.get(key)
    if hit is None or hit[0] != tag:
        with torch.no_grad():
            cache[key] = (tag, fn(param.detach()))
    return cache[key][1]


class _Packed(nn.Module):
    def __init__(self):
        super().__init__()
        self._cache = {}

    def _w16(self, key: str, param: torch.Tensor, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """16-bit [out_features, K] operand of a weight; ``rows``: its rows in that order (the SwiGLU epilogue's interleave),
        cached under ``key`` + ".sg" """
        if rows is not None:
            return _pack(self._cache, key + ".sg", param,
                         lambda p: ops.cast_pad(p.reshape(p.shape[0], -1).float()[rows].contiguous(), dtype=config.operand_dtype))
        return _pack(self._cache, key, param,
                     lambda p: ops.cast_pad(p.reshape(p.shape[0], -1).contiguous().float(), dtype=config.operand_dtype))

    def _w16lo(self, key: str, param: torch.Tensor) -> Optional[torch.Tensor]:
        """rounding residual of ``_w16`` (same layout) when ``config.precise_attention`` is on, else None"""
        if not config.precise_attention:
            return None
        return _pack(self._cache, key + ".lo", param,
                     lambda p: ops.cast_pad(p.reshape(p.shape[0], -1).contiguous().float(), dtype=config.operand_dtype, part=1))

    def _f32(self, key: str, param: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if param is None:
            return None
        return _pack(self._cache, key, param, lambda p: p.float().contiguous())

    # ---- backward helpers shared by the ViT blocks and the adapter modules --------------------------------------
    def _pack2(self, key: str, w: torch.Tensor, gamma: Optional[torch.Tensor], fn):
        """cache of a 16-bit operand derived from a weight and (optionally) a LayerScale vector"""
        tag = tuple((t.data_ptr(), t._version, getattr(t, "_asis_gen", 0)) for t in (w, gamma) if t is not None) + \
            (config.operand_dtype,)
        hit = self._cache.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                self._cache[key] = (tag, fn())
        return self._cache[key][1]

    def _wT16(self, key: str, w: torch.Tensor, gamma: Optional[torch.Tensor] = None, pow2: int = 0) -> torch.Tensor:
        """B operand of an input-gradient GEMM: (2^pow2 diag(gamma) W)^T as 16-bit [in_features, out_features]"""
        def make():
            wf = w.detach().float()
            if gamma is not None:
                wf = wf * (gamma.detach().float() * (2.0 ** pow2))[:, None]
            return ops.cast_pad(wf.t().contiguous(), dtype=config.operand_dtype)
        return self._pack2(key + (f"@{pow2}" if pow2 else ""), w, gamma, make)

    def _ls_pow2(self, key: str, gamma: Optional[torch.Tensor]) -> int:
        """k with max|gamma| 2^k in [1, 2): the power-of-two scale a LayerScale'd branch gradient carries through its 16-bit
        tensors.  The reference's init is gamma = 1e-5 (`ssl_default_config.yaml:75`): gamma W ~ 2e-7 and every gradient behind
        it sit in fp16's subnormal range even under the 2^16 loss scale, so the branch runs on 2^k gamma and the factor is
        taken out again, exactly, where its gradients are written in fp32 (Block.backward)."""
        if gamma is None:
            return 0
        # reading max|gamma| is a device-to-host sync: a frozen gamma pays it once; a TRAINED one (``optimize_backbone``) moves a
        # little every step, and k is only a range choice (any nearby power of two is exact), so it is re-read every
        # _LS_POW2_REFRESH changes of the parameter instead of 2 syncs per block per step inside the overlapped backward
        tag = (gamma.data_ptr(), gamma.device, config.operand_dtype)
        gen = (gamma._version, getattr(gamma, "_asis_gen", 0))
        ent = self._cache.get(key)
        if ent is not None and ent[0] == tag:
            st = ent[1]                       # [k, generation seen at the last read, changes since]
            if st[1] == gen:
                return st[0]
            if st[2] + 1 < _LS_POW2_REFRESH:
                st[1], st[2] = gen, st[2] + 1
                return st[0]
        m = float(gamma.detach().abs().max())
        k = 0 if not (m > 0.0 and m < float("inf")) else -int(math.floor(math.log2(m)))
        if ent is not None and ent[0] == tag and ent[1][0] != k:
            for stale in [c for c in self._cache if c.endswith(f"@{ent[1][0]}") or c.endswith(f"@{-ent[1][0]}")]:
                del self._cache[stale]        # the 2^k-scaled weight packs of the old k
        self._cache[key] = (tag, [k, gen, 0])
        return k

    def _nw_pow2(self, key: str, w: torch.Tensor, pow2: int) -> torch.Tensor:
        """fp32 LayerNorm weight times 2^pow2 (exact): takes a branch gradient's power-of-two scale out in the LayerNorm backward"""
        if pow2 == 0:
            return self._f32(key, w)
        return _pack(self._cache, f"{key}@{pow2}", w, lambda p: (p.float() * (2.0 ** pow2)).contiguous())

    def _linear_bwd(self, prefix: str, lin: nn.Linear, gamma: Optional[nn.Parameter], gname: Optional[str], dy16, dy_cs,
                    a16, inv_scale: float, grads: Optional[dict]):
        """parameter gradients of ``gamma * (A W^T + b)`` (gamma None: plain Linear) into ``grads`` (fp32, unscaled)."""
        if grads is None:
            return
        N, K = lin.weight.shape
        R = a16.shape[0]
        G = ops.wgrad(dy16.view(1, R, 1, N), a16.view(1, R, 1, K), N, 1, 1, 1, 0, 1.0).view(N, K)
        cs = ops.reduce_rows(dy_cs)
        ops.ls_linear_finish(G, lin.weight.detach() if gamma is not None else None,
                             lin.bias.detach() if lin.bias is not None else None,
                             gamma.detach() if gamma is not None else None, cs, inv_scale, grads[prefix + ".weight"],
                             grads.get(prefix + ".bias") if lin.bias is not None else None,
                             grads[gname] if gamma is not None else None)



def make_2tuple(x):
    if isinstance(x, tuple):
        assert len(x) == 2
        return x
    assert isinstance(x, int)
    return (x, x)


class PatchEmbed(_Packed):
    """`dinov2/layers/patch_embed.py:26-81`: Conv2d(k=s=P) as im2col + MFMA GEMM -> (B, N, D) fp32."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None,
                 flatten_embedding=True):
        super().__init__()
        image_HW, patch_HW = make_2tuple(img_size), make_2tuple(patch_size)
        if in_chans != 3 or patch_HW[0] != patch_HW[1]:
            raise ValueError("PatchEmbed: only 3-channel images and square patches are supported")
        self.img_size, self.patch_size = image_HW, patch_HW
        self.patches_resolution = (image_HW[0] // patch_HW[0], image_HW[1] // patch_HW[1])
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim, self.flatten_embedding = in_chans, embed_dim, flatten_embedding
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_HW, stride=patch_HW)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()
        if norm_layer:
            raise ValueError("PatchEmbed: norm_layer is not used on this path (vision_transformer.py:104)")

    def tokens(self, x: torch.Tensor, out: Optional[torch.Tensor] = None):
        """-> (tokens fp32 (B, N, D) — written into ``out`` fp32 [B*N, D] when given —, a16 = the 16-bit im2col operand (hi
        half) for the weight gradient).  With
        ``config.split_conv`` the conv runs on hi + lo halves of pixels and weights (three MFMA passes over K = 588 -> 640):
        its 16-bit operand rounding is the largest single error term of the step on the features (tests/precision_probe.py:
        2.7e-4 of 3.3e-4 on the adapter stream) because it enters BOTH ViT passes at their very first layer, while the conv
        is 0.07 % of the step's FLOPs."""
        B, _, H, W = x.shape
        P = self.patch_size[0]
        K = 3 * P * P
        dt = config.operand_dtype
        bias = self._f32("b", self.proj.bias)
        if config.split_conv:
            ldk = (K + 63) // 64 * 64
            flat = lambda p: p.reshape(p.shape[0], -1).contiguous().float()
            w_hi = _pack(self._cache, "w64", self.proj.weight, lambda p: ops.cast_pad(flat(p), ldk, dt))
            w_lo = _pack(self._cache, "w64lo", self.proj.weight, lambda p: ops.cast_pad(flat(p), ldk, dt, part=1))
            a, a_lo = ops.im2col_patch(x.contiguous().float(), P, ldk, dt, split=True)
            if out is None:
                out = torch.empty((a.shape[0], self.embed_dim), device=x.device, dtype=torch.float32)
            ops.gemm_split(a, a_lo, w_hi, w_lo, out=out, bias_n=bias)
        else:
            ldk = (K + 7) // 8 * 8
            w16 = _pack(self._cache, "w", self.proj.weight,
                        lambda p: ops.cast_pad(p.reshape(p.shape[0], -1).contiguous().float(), ldk, dt))
            a = ops.im2col_patch(x.contiguous().float(), P, ldk, dt)
            out = ops.gemm(a, w16, out=out, out_f32=True, bias_n=bias)
        return out.view(B, (H // P) * (W // P), self.embed_dim), a

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, _, H, W = x.shape
        P = self.patch_size[0]
        assert H % P == 0, f"Input image height {H} is not a multiple of patch height {P}"
        assert W % P == 0, f"Input image width {W} is not a multiple of patch width: {P}"
        out, _ = self.tokens(x)
        if not self.flatten_embedding:
            out = out.reshape(-1, H // P, W // P, self.embed_dim)
        return out


class LayerScale(nn.Module):
    """`dinov2/layers/layer_scale.py:16-27`; fused into the GEMM epilogue inside ``Block``."""

    def __init__(self, dim: int, init_values: Union[float, torch.Tensor] = 1e-5, inplace: bool = False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class Attention(_Packed):
    """`dinov2/layers/attention.py:33-69` (p_drop = 0 on this path)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        if dim % num_heads or dim // num_heads != 64:
            raise ValueError("attention kernel supports head dim 64 (every DINOv2 arch): "
                             f"got dim={dim}, heads={num_heads}")
        if attn_drop or proj_drop:
            raise ValueError("dropout is 0 on the AdapterSIS path (models/__init__.py:14-29)")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)

    def _qkv_folded(self):
        """(16-bit qkv weight, fp32 bias, rounding residual | None) with the q rows multiplied by scale * log2(e) in fp32
        BEFORE the 16-bit rounding (`attention.py:58-60`: q * scale; the log2(e) turns exp into exp2): q' leaves the GEMM
        with one rounding, the attention kernel needs neither the multiply nor the subtraction per score."""
        c = self.scale * 1.4426950408889634
        D = self.qkv.weight.shape[1]

        def scaled(p):
            w = p.reshape(p.shape[0], -1).float().clone()
            w[:D] *= c
            return w.contiguous()
        w = _pack(self._cache, "qkv.fold", self.qkv.weight, lambda p: ops.cast_pad(scaled(p), dtype=config.operand_dtype))
        wlo = None
        if config.precise_attention:
            wlo = _pack(self._cache, "qkv.fold.lo", self.qkv.weight,
                        lambda p: ops.cast_pad(scaled(p), dtype=config.operand_dtype, part=1))
        bias = None
        if self.qkv.bias is not None:
            bias = _pack(self._cache, "qkv_b.fold", self.qkv.bias, lambda p: scaled(p).reshape(-1))
        return w, bias, wlo

    def _qkv_ln_folded(self, norm: nn.LayerNorm):
        """qkv weight with diag(norm.weight) folded in (and the q rows' scale * log2(e) when config.fold_attn_scale), for the
        LayerNorm-fold chain (config.ln_fold): -> (W' 16-bit [3D, D], b' fp32 [3D] = b + W norm.bias, cs fp32 [3D] = row sums
        of the ROUNDED W' — what the MFMA really multiplies the mean with)."""
        c = self.scale * 1.4426950408889634 if config.fold_attn_scale else 1.0
        D = self.qkv.weight.shape[1]
        ps = [p for p in (self.qkv.weight, self.qkv.bias, norm.weight, norm.bias) if p is not None]
        tag = tuple((t.data_ptr(), t._version, getattr(t, "_asis_gen", 0)) for t in ps) + (config.operand_dtype, c)
        hit = self._cache.get("qkv.lnfold")
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                w = self.qkv.weight.detach().float().clone()
                w[:D] *= c
                b = self.qkv.bias.detach().float().clone() if self.qkv.bias is not None else torch.zeros(3 * D, device=w.device)
                b[:D] *= c
                b = b + w @ norm.bias.detach().float()
                w16 = ops.cast_pad((w * norm.weight.detach().float()[None, :]).contiguous(), dtype=config.operand_dtype)
                self._cache["qkv.lnfold"] = (tag, (w16, b.contiguous(), w16.float().sum(1).contiguous()))
        return self._cache["qkv.lnfold"][1]

    def attend(self, xn: torch.Tensor, B: int, N: int) -> torch.Tensor:
        """xn: 16-bit [B*N, D] (already normalised) -> 16-bit attention output [B*N, D] (before proj)."""
        return self.attend_rows(xn, [(B, N)])[0]

    def attend_rows(self, xn: torch.Tensor, segs, ln=None):
        """Several token batches stacked along the rows (``segs`` = [(B, N), ...], e.g. the cls+pos pass and the raw
        patch-token pass of `train.py:287,300-302`): one q|k GEMM over all rows, attention per batch.
        -> (o, o_lo): o_lo = the rounding residual of o (config.split_attn_out) or None.
        ``ln`` = (norm1, mr): ``xn`` is the hi plane of the UN-normalised stream and mr its per-row (mean, rstd): the
        LayerNorm is folded into the projection weights and undone in the GEMM epilogues (config.ln_fold)."""
        D = xn.shape[1]
        if ln is not None:
            norm, mr = ln
            w, bias, cs = self._qkv_ln_folded(norm)
            scale = None if config.fold_attn_scale else self.scale
            o = torch.empty((xn.shape[0], D), device=xn.device, dtype=xn.dtype)
            o_lo = torch.empty_like(o) if config.split_attn_out else None
            one_launch = len(segs) == 2
            ldv_all = (max(n for _, n in segs) + 63) // 64 * 64
            vt_all = torch.empty((sum(b for b, _ in segs), D, ldv_all), device=xn.device, dtype=xn.dtype)
            spare = xn.untyped_storage().nbytes() // xn.element_size() - (xn.storage_offset() + xn.shape[0] * D)
            side = None
            if one_launch and config.vt_stream:
                main = torch.cuda.current_stream()
                side = _vt_side_stream(xn.device)
                side.wait_stream(main)
            else:
                qk = ops.gemm(xn, w[: 2 * D], bias_n=bias[: 2 * D], ln=(mr, cs[: 2 * D], False))
            r0 = b0 = 0
            for B, N in segs:
                r1 = r0 + B * N
                vt = vt_all[b0:b0 + B]
                N4 = (N + 7) // 8 * 8      # the 16-bit epilogue that undoes the LayerNorm stores 8 columns per lane
                if spare < (N4 - N) * D:
                    raise ValueError("attend_rows(ln=...): the hi plane needs 8 spare rows behind it")
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    ops.gemm(w[2 * D:], xn[r0:r1].as_strided((B, N4, D), (N * D, D, 1)),
                             out=vt.as_strided((B, D, N4), (D * ldv_all, ldv_all, 1)), bias_m=bias[2 * D:], ln=(mr[r0:], cs[2 * D:], True))
                if not one_launch:
                    ops.attention_fwd(qk[r0:r1, :D], qk[r0:r1, D:], vt, B, self.num_heads, N, scale, out=o[r0:r1],
                                      out_lo=None if o_lo is None else o_lo[r0:r1])
                r0, b0 = r1, b0 + B
            if side is not None:
                qk = ops.gemm(xn, w[: 2 * D], bias_n=bias[: 2 * D], ln=(mr, cs[: 2 * D], False))
                main.wait_stream(side)
                mr.record_stream(side)
            if one_launch:
                (B1, N1), (B2, N2) = segs
                ops.attention_fwd_seg(qk[:, :D], qk[:, D:], vt_all, B1, N1, B2, N2, self.num_heads, scale, out=o, out_lo=o_lo)
            if r0 != xn.shape[0]:
                raise ValueError("attend_rows: segments do not cover the rows")
            return o, o_lo
        if config.fold_attn_scale:
            w, bias, wlo = self._qkv_folded()       # q rows carry scale * log2(e): asis_attention_fwd_prescaled
            scale = None
        else:
            w = self._w16("qkv", self.qkv.weight)  # [3D, D] rows q | k | v
            bias = self._f32("qkv_b", self.qkv.bias)
            wlo = self._w16lo("qkv", self.qkv.weight)
            scale = self.scale
        o = torch.empty((xn.shape[0], D), device=xn.device, dtype=xn.dtype)
        o_lo = torch.empty_like(o) if config.split_attn_out else None   # rounding residual of o: proj's split A operand
        if config.fused_qkv and len(segs) <= 2:
            # `attention.py:58`: one qkv GEMM; the attention kernel reads V row-major out of it (asis_attention_fwd_qkv)
            if sum(b * n for b, n in segs) != xn.shape[0]:
                raise ValueError("attend_rows: segments do not cover the rows")
            qkv = ops.gemm(xn, w, bias_n=bias, b_lo=wlo)
            ops.attention_fwd_qkv(qkv, list(segs), self.num_heads, scale, o, out_lo=o_lo)
            return o, o_lo
        one_launch = len(segs) == 2
        ldv_all = (max(n for _, n in segs) + 63) // 64 * 64
        vt_all = torch.empty((sum(b for b, _ in segs), D, ldv_all), device=xn.device, dtype=xn.dtype) if one_launch else None
        spare = xn.untyped_storage().nbytes() // xn.element_size() - (xn.storage_offset() + xn.shape[0] * D)
        # stacked form: the V^T GEMMs go to a side stream, the q|k GEMM stays on the compute stream (config.vt_stream)
        side = None
        if one_launch and config.vt_stream and xn.is_cuda:
            main = torch.cuda.current_stream()
            side = _vt_side_stream(xn.device)
            side.wait_stream(main)      # xn, and whatever read the recycled blocks of vt_all
        else:
            qk = ops.gemm(xn, w[: 2 * D], bias_n=None if bias is None else bias[: 2 * D], b_lo=None if wlo is None else wlo[: 2 * D])
        r0 = b0 = 0
        for B, N in segs:
            r1 = r0 + B * N
            ldvt = ldv_all if one_launch else (N + 63) // 64 * 64
            vt = vt_all[b0:b0 + B] if one_launch else torch.empty((B, D, ldvt), device=xn.device, dtype=xn.dtype)
            # N rounded up to 4 keeps the vector epilogue (N = 1765 fell to the scalar one: 101 vs 69 us); the extra
            # columns land in V^T's pad region, which the attention kernel zeroes in registers, and their operand rows
            # are the first tokens of the next image (the last image reads the spare rows behind ``xn``)
            N4 = (N + 3) // 4 * 4 if spare >= 4 * D else N
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                ops.gemm(w[2 * D:], xn[r0:r1].as_strided((B, N4, D), (N * D, D, 1)),
                         out=vt.as_strided((B, D, N4), (D * ldvt, ldvt, 1)), bias_m=None if bias is None else bias[2 * D:],
                         a_lo=None if wlo is None else wlo[2 * D:])
            if not one_launch:
                ops.attention_fwd(qk[r0:r1, :D], qk[r0:r1, D:], vt, B, self.num_heads, N, scale, out=o[r0:r1],
                                  out_lo=None if o_lo is None else o_lo[r0:r1])
            r0, b0 = r1, b0 + B
        if side is not None:
            qk = ops.gemm(xn, w[: 2 * D], bias_n=None if bias is None else bias[: 2 * D], b_lo=None if wlo is None else wlo[: 2 * D])
            main.wait_stream(side)
        if one_launch:  # both token batches in one launch (fewer partial rounds of workgroups)
            (B1, N1), (B2, N2) = segs
            ops.attention_fwd_seg(qk[:, :D], qk[:, D:], vt_all, B1, N1, B2, N2, self.num_heads, scale, out=o, out_lo=o_lo)
        if r0 != xn.shape[0]:
            raise ValueError("attend_rows: segments do not cover the rows")
        return o, o_lo

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("Attention is driven by Block.forward (LayerNorm / LayerScale / residual are fused around it)")


class MemEffAttention(Attention):
    """`attention.py:72-89`: same maths as ``Attention``; here both are the fused flash kernel."""


class Mlp(_Packed):
    """`dinov2/layers/mlp.py:17-40` (erf GELU fused into the fc1 epilogue)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)


class SwiGLUFFN(_Packed):
    """`dinov2/layers/swiglu_ffn.py:14-34`."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0.0, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)

    # the SwiGLU epilogue of the w12 GEMM (ops.ACT_SILU_MUL) reads w12 / its bias with x1 and x2 rows interleaved in groups of 16
    def sg_rows(self) -> torch.Tensor:
        return _pack(self._cache, "w12.sgrows", self.w12.weight, lambda p: ops.swiglu_rows(p.shape[0] // 2, p.device))

    def sg_bias(self) -> Optional[torch.Tensor]:
        if self.w12.bias is None:
            return None
        rows = self.sg_rows()
        return _pack(self._cache, "w12_b.sg", self.w12.bias, lambda p: p.float()[rows].contiguous())


class SwiGLUFFNFused(SwiGLUFFN):
    """`swiglu_ffn.py:54-72`: hidden = (int(h*2/3)+7)//8*8."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0.0, bias=True):
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        hidden_features = (int(hidden_features * 2 / 3) + 7) // 8 * 8
        super().__init__(in_features, hidden_features, out_features, bias=bias)


class Block(_Packed):
    """`dinov2/layers/block.py:38-114`, eval branch (`:111-113`); drop_path is 0 on this path."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, drop=0.0,
                 attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 attn_class=Attention, ffn_layer=Mlp):
        super().__init__()
        if drop_path or drop:
            raise ValueError("stochastic depth / dropout are 0 on the AdapterSIS path")
        self.norm1 = norm_layer(dim)
        self.attn = attn_class(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = ffn_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, bias=ffn_bias)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.sample_drop_ratio = drop_path

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x fp32 (B, N, D) -> fp32 (B, N, D); out of place like the reference."""
        B, N, D = x.shape
        x2 = x.reshape(B * N, D)
        if x2.dtype != torch.float32 or not x2.is_contiguous():
            x2 = x2.float().contiguous()
        return self.forward_rows(x2, [(B, N)]).view(B, N, D)

    def _w16lo_any(self, owner, key: str, w: torch.Tensor, rows: Optional[torch.Tensor] = None):
        """rounding residual of ``owner._w16(key, w, rows)`` (same layout), whatever config.precise_attention says (precise_level 2)"""
        sel = (lambda t: t) if rows is None else (lambda t: t[rows])
        return _pack(owner._cache, key + (".sg" if rows is not None else "") + ".lo_any", w,
                     lambda p: ops.cast_pad(sel(p.reshape(p.shape[0], -1).float()).contiguous(), dtype=config.operand_dtype, part=1))

    def _w16mx_any(self, owner, key: str, w: torch.Tensor, rows: Optional[torch.Tensor] = None):
        """(MX plane of the weight's (hi, lo) pair — weight side: (lo8, hi8) per element —, its absolute maximum [1])"""
        return _pack(owner._cache, key + (".sg" if rows is not None else "") + ".mx_any", w,
                     lambda p: ops.mx_from_pair(owner._w16(key, w, rows), self._w16lo_any(owner, key, w, rows), wside=True))

    def _mx_ok(self, owner, key: str, lin: nn.Linear, M: int) -> bool:
        N, K = lin.weight.shape
        return config.mx_dense_on() and K % 64 == 0 and N >= 256 and M >= 256 and N * K < (1 << 31)

    def _split_lin(self, owner, key: str, lin: nn.Linear, a_hi, a_lo, a_mx=None, rows=None, **kw):
        """one linear layer of precise_level 2: A W^T on hi + lo operands — three 16-bit K parts, or (config.mx_dense_on) the
        16-bit part + ONE block-scaled fp8 pass over the MX planes of both operands.  ``a_mx`` = (plane, amax) made by the
        producer (LayerNorm), else the plane is made here from the stored (hi, lo) pair.  ``rows``: the weight's rows in that
        order (w12 for the SwiGLU epilogue)."""
        w = owner._w16(key, lin.weight, rows)
        if self._mx_ok(owner, key, lin, a_hi.shape[0]):
            a_pl, amax_a = a_mx if a_mx is not None else ops.mx_from_pair(a_hi, a_lo)
            w_mx, amax_w = self._w16mx_any(owner, key, lin.weight, rows)
            return ops.gemm(a_hi, w, a_lo=a_pl, b_lo=w_mx, mx=(amax_a, amax_w), **kw)
        return ops.gemm(a_hi, w, a_lo=a_lo, b_lo=self._w16lo_any(owner, key, lin.weight, rows), **kw)

    def _ln_split(self, key: str, norm: nn.LayerNorm, x2: torch.Tensor, mx: bool):
        """LayerNorm output as a split-precision A operand: (hi, lo, None), or (hi, None, (MX plane, amax bound)) in ONE pass"""
        dt = config.operand_dtype
        w, b = self._f32(key + "w", norm.weight), self._f32(key + "b", norm.bias)
        if mx:
            # |LN(x)_i| <= sqrt(D - 1) |w_i| + |b_i|: a bound, not the maximum (asis_layernorm_mx: costs nothing at e4m3's range)
            amax = _pack(self._cache, key + ".amax", norm.weight,
                         lambda p: ((x2.shape[1] - 1) ** 0.5 * p.float().abs().max() + norm.bias.detach().float().abs().max()).reshape(1).contiguous())
            hi, pl = ops.layernorm_mx(x2, w, b, norm.eps, dt, amax)
            return hi, None, (pl, amax)
        xn32 = ops.layernorm(x2, w, b, norm.eps, torch.float32)
        D = x2.shape[1]
        return ops.cast_pad(xn32, D, dt), ops.cast_pad(xn32, D, dt, part=1), None

    def _forward_rows_precise(self, x2: torch.Tensor, segs) -> torch.Tensor:
        """``forward_rows`` with every linear layer on split-precision operands (config.precise_level 2): LayerNorm / GELU / SwiGLU /
        attention outputs and all weights as hi + lo halves; the fused attention keeps single 16-bit q, k, v, P."""
        D = x2.shape[1]
        R = x2.shape[0]
        dt = config.operand_dtype
        a, m = self.attn, self.mlp
        g1 = self._f32("g1", self.ls1.gamma) if isinstance(self.ls1, LayerScale) else None
        g2 = self._f32("g2", self.ls2.gamma) if isinstance(self.ls2, LayerScale) else None
        parts = config.precise_parts      # which of the four linear layers run split (default: all)
        n1w, n1b = self._f32("n1w", self.norm1.weight), self._f32("n1b", self.norm1.bias)
        fold_q = config.fold_attn_scale and "qkv" not in parts and _PRECISE_FOLD_Q
        if "qkv" in parts:
            xn, xn_lo, xn_mx = self._ln_split("n1", self.norm1, x2, self._mx_ok(a, "qkv", a.qkv, R))
            qkv = self._split_lin(a, "qkv", a.qkv, xn, xn_lo, xn_mx, bias_n=a._f32("qkv_b", a.qkv.bias))     # 16-bit out: q, k, v operands
        elif fold_q:
            qw, qb, _ = a._qkv_folded()      # q rows carry scale * log2(e) (folded before the 16-bit rounding): the folded attention kernel
            qkv = ops.gemm(ops.layernorm(x2, n1w, n1b, self.norm1.eps, dt), qw, bias_n=qb)
        else:
            qkv = ops.gemm(ops.layernorm(x2, n1w, n1b, self.norm1.eps, dt), a._w16("qkv", a.qkv.weight), bias_n=a._f32("qkv_b", a.qkv.bias))
        o = torch.empty((R, D), device=x2.device, dtype=dt)
        o_lo = torch.empty_like(o) if "proj" in parts else None
        if sum(B * N for B, N in segs) != R:
            raise ValueError("forward_rows: segments do not cover the rows")
        o_mx = None
        if fold_q and len(segs) <= 2:
            # V row-major out of the one qkv GEMM, both stacked token batches in one launch (asis_attention_fwd_qkv); when proj
            # takes MX operands the kernel writes o's lo half in the MX form itself, scaled by max |v| >= max |o| (o is a convex
            # combination of V rows): no absmax + conversion passes over (o, o_lo)
            if o_lo is not None and _ATTN_MX_OUT and self._mx_ok(a, "proj", a.proj, R):
                o_mx = ops.absmax16(qkv[:, 2 * D:3 * D])
            ops.attention_fwd_qkv(qkv, list(segs), a.num_heads, None, o, out_lo=o_lo, mx_amax=o_mx)
        else:
            r0 = 0
            for B, N in segs:
                r1 = r0 + B * N
                vt = ops.transpose_tokens(qkv[r0:r1, 2 * D:], B, N)
                ops.attention_fwd(qkv[r0:r1, :D], qkv[r0:r1, D:2 * D], vt, B, a.num_heads, N, None if fold_q else a.scale,
                                  out=o[r0:r1], out_lo=None if o_lo is None else o_lo[r0:r1])
                r0 = r1
        pkw = dict(out_f32=True, bias_n=a._f32("proj_b", a.proj.bias), scale_n=g1, res=x2)
        if o_mx is not None:
            x1 = self._split_lin(a, "proj", a.proj, o, None, a_mx=(o_lo, o_mx), **pkw)
        else:
            x1 = self._split_lin(a, "proj", a.proj, o, o_lo, **pkw) if o_lo is not None else ops.gemm(o, a._w16("proj", a.proj.weight), **pkw)
        n2w, n2b = self._f32("n2w", self.norm2.weight), self._f32("n2b", self.norm2.bias)
        lin_in, k_in = (m.fc1, "fc1") if isinstance(m, Mlp) else (m.w12, "w12")
        lin_out, k_out = (m.fc2, "fc2") if isinstance(m, Mlp) else (m.w3, "w3")
        ikw = dict(out_f32=True, bias_n=m._f32(k_in + "_b", lin_in.bias))
        split_out = "fc2" in parts
        mx_in = "fc1" in parts and self._mx_ok(m, k_in, lin_in, R)
        # SwiGLU in the w12 GEMM's epilogue (ops.ACT_SILU_MUL: rows of w12 interleaved, the fp32 [R, 2 Hd] pre-activation never
        # written) wherever the launch form has it and the gate's output is a single 16-bit operand
        sg = (not isinstance(m, Mlp)) and not split_out and \
            ops.swiglu_fused_ok(R, lin_in.out_features // 2, lin_in.in_features, "fc1" in parts, mx_in)
        ge = isinstance(m, Mlp) and not split_out          # likewise erf-GELU with a single 16-bit output: the GEMM's own epilogue
        if sg:
            rows = m.sg_rows()
            ikw = dict(bias_n=m.sg_bias(), act=ops.ACT_SILU_MUL, rows=rows)
        elif ge:
            ikw = dict(bias_n=m._f32(k_in + "_b", lin_in.bias), act=ops.ACT_GELU)
        if "fc1" in parts:
            xn2, xn2_lo, xn2_mx = self._ln_split("n2", self.norm2, x1, mx_in)
            hpre = self._split_lin(m, k_in, lin_in, xn2, xn2_lo, xn2_mx, **ikw)
        else:
            rows = ikw.pop("rows", None)
            hpre = ops.gemm(ops.layernorm(x1, n2w, n2b, self.norm2.eps, dt), m._w16(k_in, lin_in.weight, rows), **ikw)
        if sg or ge:
            h, h_lo = hpre, None
        elif isinstance(m, Mlp):
            h, h_lo = ops.gelu_split(hpre, dt, split=split_out)
        else:
            h, h_lo = ops.swiglu(hpre, dt, split=True) if split_out else (ops.swiglu(hpre, dt), None)
        okw = dict(out_f32=True, bias_n=m._f32(k_out + "_b", lin_out.bias), scale_n=g2, res=x1)
        if split_out:
            return self._split_lin(m, k_out, lin_out, h, h_lo, **okw)
        return ops.gemm(h, m._w16(k_out, lin_out.weight), **okw)

    def _lin_ln_folded(self, key: str, lin: nn.Linear, norm: nn.LayerNorm):
        """(W diag(norm.weight) as 16 bits, b + W norm.bias, row sums of the rounded W') of a linear layer behind a LayerNorm"""
        ps = [p for p in (lin.weight, lin.bias, norm.weight, norm.bias) if p is not None]
        tag = tuple((t.data_ptr(), t._version, getattr(t, "_asis_gen", 0)) for t in ps) + (config.operand_dtype,)
        hit = self._cache.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                w = lin.weight.detach().float()
                b = (lin.bias.detach().float() if lin.bias is not None else torch.zeros(w.shape[0], device=w.device)) + w @ norm.bias.detach().float()
                w16 = ops.cast_pad((w * norm.weight.detach().float()[None, :]).contiguous(), dtype=config.operand_dtype)
                self._cache[key] = (tag, (w16, b.contiguous(), w16.float().sum(1).contiguous()))
        return self._cache[key][1]

    def fold_ok(self, R: int, segs) -> bool:
        """the LayerNorm-fold chain (config.ln_fold) applies to this block on ``R`` stacked rows: every GEMM involved lands on
        a kernel that implements the epilogue fields (ops.ln_fold_supported)"""
        if not (config.ln_fold and config.operand_dtype == torch.float16 and config.precise_level < 2 and not config.precise_attention
                and not config.fused_qkv and isinstance(self.mlp, Mlp) and isinstance(self.ls1, LayerScale)
                and isinstance(self.ls2, LayerScale) and isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm)):
            return False
        D, Hd = self.attn.qkv.weight.shape[1], self.mlp.fc1.weight.shape[0]
        if config.split_attn_out:      # the producer epilogue lives on the one-tile-per-workgroup form, which takes no split operands
            return False
        ok = ops.ln_fold_supported(R, 2 * D, D) and ops.ln_fold_supported(R, D, D, producer=True)
        ok = ok and ops.ln_fold_supported(R, Hd, D) and ops.ln_fold_supported(R, D, Hd, producer=True)
        return ok and all(ops.ln_fold_supported(D, (N + 7) // 8 * 8, D, batch=B) for B, N in segs) and D % 64 == 0

    def forward_rows_fold(self, xin, segs, planes_out: bool):
        """``forward_rows`` on the LayerNorm-fold chain.  ``xin``: fp32 [R, D] (norm1 runs as a kernel, as in ``forward_rows``)
        or the planes (hi, lo, mr) a previous block left; -> fp32 [R, D], or (hi, lo, mr) when ``planes_out`` (mr = per-row
        statistics for the NEXT block's norm1: same eps, `vision_transformer.py:89`)."""
        dt = config.operand_dtype
        a, m = self.attn, self.mlp
        g1, g2 = self._f32("g1", self.ls1.gamma), self._f32("g2", self.ls2.gamma)
        if isinstance(xin, tuple):
            hi, lo, mr = xin
            R, D = hi.shape
            o, o_lo = a.attend_rows(hi, segs, ln=(self.norm1, mr))
            res_kw = dict(res16=(hi, lo))
        else:
            R, D = xin.shape
            xn = torch.empty((R + 8, D), device=xin.device, dtype=dt)[:R]
            ops.layernorm(xin, self._f32("n1w", self.norm1.weight), self._f32("n1b", self.norm1.bias), self.norm1.eps, dt, out=xn)
            o, o_lo = a.attend_rows(xn, segs)
            res_kw = dict(res=xin)
        G = (D + 63) // 64
        x1h = torch.empty((R, D), device=o.device, dtype=dt)
        x1l = torch.empty((R, D), device=o.device, dtype=dt)
        st = torch.empty((R, G, 2), device=o.device, dtype=torch.float32)
        ops.gemm(o, a._w16("proj", a.proj.weight), out=x1h, out_lo=x1l, rowstats=st, bias_n=a._f32("proj_b", a.proj.bias), scale_n=g1,
                 a_lo=o_lo, **res_kw)
        mr1 = ops.ln_stats_finalize(st, D, self.norm2.eps)
        w1, b1, cs1 = self._lin_ln_folded("fc1.lnfold", m.fc1, self.norm2)
        h = ops.gemm(x1h, w1, bias_n=b1, act=ops.ACT_GELU, ln=(mr1, cs1, False))
        w2, b2 = m._w16("fc2", m.fc2.weight), m._f32("fc2_b", m.fc2.bias)
        if not planes_out:
            return ops.gemm(h, w2, out_f32=True, bias_n=b2, scale_n=g2, res16=(x1h, x1l))
        x3h = torch.empty((R + 8, D), device=o.device, dtype=dt)[:R]      # spare rows: the next block's V^T GEMM reads past the end
        x3l = torch.empty((R, D), device=o.device, dtype=dt)
        st3 = torch.empty((R, G, 2), device=o.device, dtype=torch.float32)
        ops.gemm(h, w2, out=x3h, out_lo=x3l, rowstats=st3, bias_n=b2, scale_n=g2, res16=(x1h, x1l))
        return x3h, x3l, ops.ln_stats_finalize(st3, D, self.norm1.eps)

    def forward_rows(self, x2: torch.Tensor, segs) -> torch.Tensor:
        """The block on several token batches stacked along the rows of one fp32 [R, D] matrix (``segs`` = [(B, N), ..]).
        Everything but the attention itself is row-wise, so the two ViT passes of the training step (cls + pos-embed
        tokens and raw patch tokens, same frozen weights: `train.py:287,300-302`) share every GEMM launch: twice the
        rows per launch fill the 512 tile slots of the chip in 2.6 instead of 2 x 1.3 (-> 2 x 2) rounds on the
        N = 1024 GEMMs, and the weights are read once."""
        if config.precise_level >= 2:
            return self._forward_rows_precise(x2, segs)
        D = x2.shape[1]
        dt = config.operand_dtype
        g1 = self._f32("g1", self.ls1.gamma) if isinstance(self.ls1, LayerScale) else None
        g2 = self._f32("g2", self.ls2.gamma) if isinstance(self.ls2, LayerScale) else None
        xn = torch.empty((x2.shape[0] + 8, D), device=x2.device, dtype=dt)[: x2.shape[0]]   # 8 spare rows: see attend_rows
        ops.layernorm(x2, self._f32("n1w", self.norm1.weight), self._f32("n1b", self.norm1.bias), self.norm1.eps, dt, out=xn)
        o, o_lo = self.attn.attend_rows(xn, segs)
        x1 = ops.gemm(o, self.attn._w16("proj", self.attn.proj.weight), out_f32=True,
                      bias_n=self.attn._f32("proj_b", self.attn.proj.bias), scale_n=g1, res=x2, a_lo=o_lo,
                      b_lo=self.attn._w16lo("proj", self.attn.proj.weight))
        xn2 = ops.layernorm(x1, self._f32("n2w", self.norm2.weight), self._f32("n2b", self.norm2.bias),
                            self.norm2.eps, dt)
        m = self.mlp
        if isinstance(m, Mlp):
            h = ops.gemm(xn2, m._w16("fc1", m.fc1.weight), bias_n=m._f32("fc1_b", m.fc1.bias), act=ops.ACT_GELU)
            x3 = ops.gemm(h, m._w16("fc2", m.fc2.weight), out_f32=True, bias_n=m._f32("fc2_b", m.fc2.bias),
                          scale_n=g2, res=x1)
        else:
            if ops.swiglu_fused_ok(xn2.shape[0], m.w12.out_features // 2, m.w12.in_features, False, False):
                h = ops.gemm(xn2, m._w16("w12", m.w12.weight, m.sg_rows()), bias_n=m.sg_bias(), act=ops.ACT_SILU_MUL)
            else:
                h12 = ops.gemm(xn2, m._w16("w12", m.w12.weight), out_f32=True, bias_n=m._f32("w12_b", m.w12.bias))
                h = ops.swiglu(h12, dt)
            x3 = ops.gemm(h, m._w16("w3", m.w3.weight), out_f32=True, bias_n=m._f32("w3_b", m.w3.bias),
                          scale_n=g2, res=x1)
        return x3

    # ---- training path (block.py:89-114 under autograd; BASELINE config 4 / north_star "forward/backward") ----------
    def forward_train(self, x: torch.Tensor):
        """x fp32 (B, N, D) -> (out fp32 (B, N, D), saved activations for ``backward``).  Same arithmetic as
        ``forward`` except that fc1 keeps its 16-bit pre-activation (GELU runs as its own pass) and q, k, v come from
        one [3D] GEMM (V^T by a token transpose)."""
        B, N, D = x.shape
        x2 = x.reshape(B * N, D)
        if x2.dtype != torch.float32 or not x2.is_contiguous():
            x2 = x2.float().contiguous()
        out, saved = self.forward_train_rows(x2, [(B, N)])
        return out.view(B, N, D), saved

    def forward_train_rows(self, x2: torch.Tensor, segs):
        """``forward_train`` on several token batches stacked along the rows of one fp32 [R, D] matrix (``segs`` =
        [(B, N), ...], see ``forward_rows``): every GEMM / LayerNorm / GELU launch — and in the backward every weight-gradient
        GEMM — covers all rows, so the parameter gradients of the two ViT passes of the unfrozen adapter step (BASELINE
        config 4) are summed by construction; only the attention runs per batch."""
        dt = config.operand_dtype
        D = x2.shape[1]
        m = self.mlp
        g1 = self._f32("g1", self.ls1.gamma) if isinstance(self.ls1, LayerScale) else None
        g2 = self._f32("g2", self.ls2.gamma) if isinstance(self.ls2, LayerScale) else None
        a = self.attn
        xn = ops.layernorm(x2, self._f32("n1w", self.norm1.weight), self._f32("n1b", self.norm1.bias), self.norm1.eps, dt)
        qkv = ops.gemm(xn, a._w16("qkv", a.qkv.weight), bias_n=a._f32("qkv_b", a.qkv.bias), b_lo=a._w16lo("qkv", a.qkv.weight))
        o = torch.empty((x2.shape[0], D), device=x2.device, dtype=dt)
        o_lo = torch.empty_like(o) if config.split_attn_out else None
        if sum(B * N for B, N in segs) != x2.shape[0]:
            raise ValueError("forward_train_rows: segments do not cover the rows")
        # the log-sum-exp of every stacked batch in ONE buffer ([B1, H, N1] then [B2, H, N2]: what attention_bwd_rows reads)
        lse = torch.empty(x2.shape[0] * a.num_heads, device=x2.device, dtype=torch.float32)
        if len(segs) <= 2 and config.attn_rows:
            # V row-major straight out of the qkv GEMM (transposing LDS reads in the kernel), both batches in one launch
            ops.attention_fwd_qkv(qkv, segs, a.num_heads, a.scale, o, out_lo=o_lo, lse=lse)
        else:
            r0 = l0 = 0
            for B, N in segs:
                r1, l1 = r0 + B * N, l0 + B * a.num_heads * N
                vt = ops.transpose_tokens(qkv[r0:r1, 2 * D:], B, N)
                ops.attention_fwd(qkv[r0:r1, :D], qkv[r0:r1, D:2 * D], vt, B, a.num_heads, N, a.scale, out=o[r0:r1],
                                  lse=lse[l0:l1].view(B, a.num_heads, N), out_lo=None if o_lo is None else o_lo[r0:r1])
                r0, l0 = r1, l1
        x1 = ops.gemm(o, a._w16("proj", a.proj.weight), out_f32=True, bias_n=a._f32("proj_b", a.proj.bias), scale_n=g1,
                      res=x2, a_lo=o_lo, b_lo=a._w16lo("proj", a.proj.weight))
        xn2 = ops.layernorm(x1, self._f32("n2w", self.norm2.weight), self._f32("n2b", self.norm2.bias), self.norm2.eps, dt)
        if isinstance(m, Mlp):
            hpre = ops.gemm(xn2, m._w16("fc1", m.fc1.weight), bias_n=m._f32("fc1_b", m.fc1.bias))
            hpost = ops.gelu16(hpre)
            x3 = ops.gemm(hpost, m._w16("fc2", m.fc2.weight), out_f32=True, bias_n=m._f32("fc2_b", m.fc2.bias), scale_n=g2,
                          res=x1)
        else:  # SwiGLU (ViT-g): hpre = the fp32 [x1 | x2] of w12, hpost = silu(x1) * x2
            hpre = ops.gemm(xn2, m._w16("w12", m.w12.weight), out_f32=True, bias_n=m._f32("w12_b", m.w12.bias))
            hpost = ops.swiglu(hpre, dt)
            x3 = ops.gemm(hpost, m._w16("w3", m.w3.weight), out_f32=True, bias_n=m._f32("w3_b", m.w3.bias), scale_n=g2,
                          res=x1)
        return x3, (x2, xn, qkv, o, lse, x1, xn2, hpre, hpost, list(segs))

    def backward(self, saved, dres: torch.Tensor, inv_scale: float, grads: Optional[dict] = None, prefix: str = "") -> torch.Tensor:
        """dres fp32 [B*N, D] = loss_scale * dL/d(block output) -> loss_scale * dL/d(block input).  With ``grads``
        (name -> fp32 tensor, names as in ``state_dict`` under ``prefix``) the parameter gradients are written too;
        without it only the input gradient is produced (frozen backbone, adapters training)."""
        x2, xn, qkv, o, lse, x1, xn2, hpre, hpost, segs = saved
        dt = config.operand_dtype
        D = x2.shape[1]
        a, m = self.attn, self.mlp
        ls1 = self.ls1.gamma if isinstance(self.ls1, LayerScale) else None
        ls2 = self.ls2.gamma if isinstance(self.ls2, LayerScale) else None
        pre = prefix + "." if prefix else ""
        # per-branch power-of-two scales of the LayerScale'd gradients (see ``_ls_pow2``): dh / dO and everything computed from
        # them inside the branch carry 2^k; the factor leaves through inv_scale (weight / bias / norm gradients) and through the
        # LayerNorm backward's weight (input gradient)
        k1, k2 = self._ls_pow2("ls1.k", ls1), self._ls_pow2("ls2.k", ls2)
        inv1, inv2 = inv_scale * 2.0 ** -k1, inv_scale * 2.0 ** -k2
        # ---- MLP branch: out = x1 + ls2 * fc2(gelu(fc1(LN2(x1)))) ----
        d16, cs = ops.cast_colsum(dres, dt) if grads is not None else (ops.cast_pad(dres, D, dt), None)
        if isinstance(m, Mlp):
            lin_out, lin_in, n_out, n_in, act_bwd = m.fc2, m.fc1, "mlp.fc2", "mlp.fc1", ops.gelu16
        else:
            lin_out, lin_in, n_out, n_in, act_bwd = m.w3, m.w12, "mlp.w3", "mlp.w12", ops.swiglu_bwd
        self._linear_bwd(pre + n_out, lin_out, ls2, pre + "ls2.gamma", d16, cs, hpost, inv_scale, grads)
        R = d16.shape[0]
        if isinstance(m, Mlp) and R >= 256 and lin_out.in_features >= 128 and lin_out.in_features % 4 == 0 and D % 32 == 0:
            # fc2's input gradient with GELU's backward in the GEMM epilogue: d pre = (d16 W) * gelu'(pre)
            dh = ops.gemm(d16, self._wT16("fc2T", lin_out.weight, ls2, k2), act=ops.ACT_GELU_GRAD, aux=hpre)
        else:
            dh = ops.gemm(d16, self._wT16("fc2T", lin_out.weight, ls2, k2))        # 16-bit [R, hidden], times 2^k2
            dh = act_bwd(hpre, dh)                                                 # GELU' / SwiGLU gate backward
        self._linear_bwd(pre + n_in, lin_in, None, None, dh, ops.colsum(dh) if grads is not None else None, xn2,
                         inv2, grads)
        dln = ops.gemm(dh, self._wT16("fc1T", lin_in.weight), out_f32=True)        # fp32 [R, D], times 2^k2
        dx1, part = ops.layernorm_bwd(dln, x1, self._nw_pow2("n2w", self.norm2.weight, -k2), self.norm2.eps, res=dres)
        if grads is not None:
            red = ops.reduce_rows(part.view(part.shape[0], 2 * D), inv2)
            grads[pre + "norm2.weight"].copy_(red[:D]); grads[pre + "norm2.bias"].copy_(red[D:])
        # ---- attention branch: x1 = x + ls1 * proj(attn(LN1(x))) ----
        d16, cs = ops.cast_colsum(dx1, dt) if grads is not None else (ops.cast_pad(dx1, D, dt), None)
        self._linear_bwd(pre + "attn.proj", a.proj, ls1, pre + "ls1.gamma", d16, cs, o, inv_scale, grads)
        dO = ops.gemm(d16, self._wT16("projT", a.proj.weight, ls1, k1))            # 16-bit [R, D], times 2^k1
        dqkv = torch.empty((x2.shape[0], 3 * D), device=x2.device, dtype=dt)
        if isinstance(lse, (list, tuple)):   # per-batch [B, H, N] tensors (the MaskTransformer block's forward)
            lse = torch.cat([l.reshape(-1) for l in lse])
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        for i in range(0, len(segs), 2):     # attention backward, two stacked token batches per launch
            part = segs[i:i + 2]
            r0 = sum(B * N for B, N in segs[:i])
            r1 = r0 + sum(B * N for B, N in part)
            ops.attention_bwd_rows(q[r0:r1], k[r0:r1], v[r0:r1], o[r0:r1], dO[r0:r1], lse[r0 * a.num_heads:r1 * a.num_heads],
                                   part, a.num_heads, a.scale, dqkv=dqkv[r0:r1])
        self._linear_bwd(pre + "attn.qkv", a.qkv, None, None, dqkv, ops.colsum(dqkv) if grads is not None else None, xn,
                         inv1, grads)
        dln = ops.gemm(dqkv, self._wT16("qkvT", a.qkv.weight), out_f32=True)
        dx, part = ops.layernorm_bwd(dln, x2, self._nw_pow2("n1w", self.norm1.weight, -k1), self.norm1.eps, res=dx1)
        if grads is not None:
            red = ops.reduce_rows(part.view(part.shape[0], 2 * D), inv1)
            grads[pre + "norm1.weight"].copy_(red[:D]); grads[pre + "norm1.bias"].copy_(red[D:])
        return dx


def run_blocks(blocks, x: torch.Tensor, segs) -> torch.Tensor:
    """fp32 [R, D] through a run of frozen blocks -> fp32 [R, D].  Where ``Block.fold_ok`` says so the residual stream stays in
    the two-plane form between the blocks (config.ln_fold): the first block of the run reads fp32 (its norm1 is a kernel), the
    last one writes fp32, no LayerNorm kernel in between."""
    state = x
    R = x.shape[0]
    n = len(blocks)
    for i, blk in enumerate(blocks):
        if blk.fold_ok(R, segs):
            state = blk.forward_rows_fold(state, segs, planes_out=i + 1 < n and blocks[i + 1].fold_ok(R, segs))
        else:
            state = blk.forward_rows(state, segs)
    return state


class NestedTensorBlock(Block):
    """`block.py:165-254`: list (nested-tensor) inputs need xformers in the reference; tensors only here."""

    def forward(self, x_or_x_list):
        if isinstance(x_or_x_list, torch.Tensor):
            return super().forward(x_or_x_list)
        raise AssertionError("xFormers is required for using nested tensors")
