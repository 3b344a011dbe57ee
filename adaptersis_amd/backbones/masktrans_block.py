"""HIP-backed MaskTransformer decode head (Segmenter-style) — API / ``state_dict`` mirror of `backbones/masktrans_block.py`
(``FeedForward``, ``Attention``, ``Block``) and of the ``MaskTransformer`` of `eval/eval_dinov2_masktrans.py:400-477`.

    x = proj_dec(tokens)                       Linear(d_encoder, d_model) on the ViT patch tokens          `:445`
    x = cat(x, cls_emb)                        n_cls learnable class tokens behind the N patches         `:446-447`
    x = blocks(x) ; x = decoder_norm(x)        pre-norm transformer blocks, LayerNorm(eps 1e-5)           `:448-450`
    patches = x[:, :-n_cls] @ proj_patch ; cls = x[:, -n_cls:] @ proj_classes                             `:452-454`
    patches /= ||patches|| ; cls /= ||cls|| ; masks = patches @ cls^T ; masks = mask_norm(masks)          `:456-460`
    -> (B, n_cls, H/patch, W/patch)                                                                        `:461`

A block of `backbones/masktrans_block.py:75-89` is the plain pre-norm transformer block: LayerNorm -> multi-head attention
(qkv with bias, softmax(q k^T / sqrt(d)), proj) -> residual -> LayerNorm -> Linear / GELU / Linear -> residual.  That is the
ViT block of this package without LayerScale, so ``Block`` here IS `dinov2.layers.blocks.Block` (fused attention forward +
backward, GEMM epilogues, weight-gradient GEMMs — the config-4 machinery) under the reference's constructor; the parameter
names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2) are the same.  The tail runs on `csrc/maskhead.hip`.
Tokens of all images are rows of one fp32 matrix, batch b = rows b*(N+n_cls) ..; both D x D projections run over all rows
(one GEMM each: the class projection of the patch rows is 0.1 TFLOP of waste that buys uniform row indexing).
Precision: ``mask_norm`` (LayerNorm over n_cls values) amplifies the error of the cosines 2.2-2.8x, so under
``config.split_conv`` (default) every linear layer of the head runs on split-precision operands (masks 2.5e-4 .. 6.8e-4 against the
golden; 1.4e-3 .. 2.7e-3 with single-pass 16-bit operands).

Dropout: the reference script builds the head with ``dropout=0.1`` (`:139`; `masktrans_block.py:19,43-45,66,70,82`: nn.Dropout on
the attention probabilities, the projection output, behind GELU and behind fc2).  In training mode with ``dropout > 0`` a
block runs ``_fwd_dropout`` / ``_bwd_dropout``: counter-based masks (Philox keyed by step seed, dropout-layer number and element
index: csrc/dropout.hip — forward, backward and the mask export regenerate them, nothing is stored), an UNFUSED attention for
the probability dropout (scores / probabilities materialised per head through batched GEMMs + two row kernels; the fused flash
kernels serve every path without dropout), the other three sites as one elementwise kernel each.  ``drop_path > 0``
(`drop_path_rate=0.0` in the script) is not implemented and raises in training mode.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .. import config, ops
from ..dinov2.layers import blocks as L
from ..dinov2.layers.blocks import _Packed, _pack


class FeedForward(L.Mlp):
    """`masktrans_block.py:11-31` — fc1 / GELU / fc2 (parameter container; runs inside ``Block``)."""

    def __init__(self, dim, hidden_dim, dropout, out_dim=None):
        super().__init__(in_features=dim, hidden_features=hidden_dim, out_features=out_dim or dim, bias=True)
        self.dropout = float(dropout)

    @property
    def unwrapped(self):
        return self


class Block(L.Block):
    """`masktrans_block.py:75-89`: ``Block(dim, heads, mlp_dim, dropout, drop_path)``."""

    def __init__(self, dim, heads, mlp_dim, dropout, drop_path):
        if dim % heads or dim // heads != 64:
            raise ValueError("masktrans Block: the fused attention kernels are built for head dim 64 (the script uses dim // 64 heads)")
        super().__init__(dim, heads, mlp_ratio=mlp_dim / dim, qkv_bias=True, proj_bias=True, ffn_bias=True, init_values=None,
                         norm_layer=nn.LayerNorm, attn_class=L.MemEffAttention)
        self.dropout, self.drop_path_rate = float(dropout), float(drop_path)

        self._drop = None          # (step seed, first dropout-layer number of this block): set by MaskTransformer per forward

    def check_train(self):
        if self.drop_path_rate != 0.0:
            raise NotImplementedError("masktrans Block: drop_path > 0 is not implemented on the HIP path (the script builds the "
                                      "head with drop_path_rate=0.0)")
        if not (0.0 <= self.dropout < 1.0):
            raise ValueError("masktrans Block: dropout must be in [0, 1)")

    # ---- training forward / backward with dropout ----------------------------------------------------------------------
    SITE_ATTN, SITE_PROJ, SITE_GELU, SITE_FC2 = 0, 1, 2, 3

    def _fwd_dropout(self, x2: torch.Tensor, segs):
        """`masktrans_block.py:75-89` in training mode with dropout p > 0 (see the module docstring); saves for ``_bwd_dropout``."""
        dt = config.operand_dtype
        seed, site0 = self._drop
        p = self.dropout
        R, D = x2.shape
        a, m = self.attn, self.mlp
        H = a.num_heads
        xn32 = ops.layernorm(x2, self._f32("n1w", self.norm1.weight), self._f32("n1b", self.norm1.bias), self.norm1.eps, torch.float32)
        xn, xn_lo = ops.cast_pad(xn32, D, dt), ops.cast_pad(xn32, D, dt, part=1)
        qkv32 = self._split_linear(xn, xn_lo, a._w16("qkv", a.qkv.weight), self._lo(a, "qkv", a.qkv.weight), a._f32("qkv_b", a.qkv.bias))
        qkv = ops.cast_pad(qkv32, 3 * D, dt)
        o = torch.empty((R, D), device=x2.device, dtype=dt)
        if len(segs) != 1:
            raise ValueError("masktrans Block: the dropout path takes one token batch (the head's tokens are one batch)")
        (B, N), = segs
        ld = ops.token_ld(N)
        q3 = qkv.view(B, N, 3 * D)
        vt = ops.transpose_tokens(qkv[:, 2 * D:], B, N)                       # [B, D, ld], zero padded
        S = torch.empty((H, B, N, ld), device=x2.device, dtype=torch.float32)
        for h in range(H):                                                    # scores per head: q_h k_h^T, one launch per head over the batch
            ops.gemm(q3[:, :, h * 64:(h + 1) * 64], q3[:, :, D + h * 64:D + (h + 1) * 64], out=S[h, :, :, :N], out_f32=True)
        P, Pd = ops.softmax_dropout_fwd(S.view(H * B * N, ld), N, a.scale, seed, site0 + self.SITE_ATTN, p, dt)
        del S
        o3 = o.view(B, N, D)
        Pd4 = Pd.view(H, B, N, ld)
        for h in range(H):
            ops.gemm(Pd4[h], vt[:, h * 64:(h + 1) * 64, :], out=o3[:, :, h * 64:(h + 1) * 64])
        y = ops.gemm(o, a._w16("proj", a.proj.weight), out_f32=True, bias_n=a._f32("proj_b", a.proj.bias),
                     b_lo=self._lo(a, "proj", a.proj.weight))
        x1 = ops.dropout_f32(y, seed, site0 + self.SITE_PROJ, p, res=x2)
        xn2_32 = ops.layernorm(x1, self._f32("n2w", self.norm2.weight), self._f32("n2b", self.norm2.bias), self.norm2.eps, torch.float32)
        xn2, xn2_lo = ops.cast_pad(xn2_32, D, dt), ops.cast_pad(xn2_32, D, dt, part=1)
        hpre32 = self._split_linear(xn2, xn2_lo, m._w16("fc1", m.fc1.weight), self._lo(m, "fc1", m.fc1.weight), m._f32("fc1_b", m.fc1.bias))
        hpost, hpost_lo = ops.gelu_split(hpre32, dt)
        hsaved = ops.dropout_t16(hpost.clone(), seed, site0 + self.SITE_GELU, p, rescale=True)      # drop(gelu) as fc2's weight gradient sees it
        # the operand pair of fc2 is only ZEROED (hi + lo stays exact); its 1 / (1 - p) and the bias go into the next kernel
        ops.dropout_t16(hpost, seed, site0 + self.SITE_GELU, p, x_lo=hpost_lo, rescale=False)
        f2 = self._split_linear(hpost, hpost_lo, m._w16("fc2", m.fc2.weight), self._lo(m, "fc2", m.fc2.weight), None)
        x3 = ops.dropout_f32(f2, seed, site0 + self.SITE_FC2, p, res=x1, alpha=1.0 / (1.0 - p), bias_n=m._f32("fc2_b", m.fc2.bias))
        hpre = ops.cast_pad(hpre32, hpre32.shape[1], dt)
        return x3, dict(x2=x2, xn=xn, qkv=qkv, o=o, P=P, Pd=Pd, x1=x1, xn2=xn2, hpre=hpre, hpost=hsaved, segs=list(segs),
                        drop=(seed, site0, p))

    def _bwd_dropout(self, sv, dres: torch.Tensor, inv_scale: float, grads, prefix: str) -> torch.Tensor:
        """Backward of ``_fwd_dropout`` (structure of ``dinov2.layers.blocks.Block.backward``; every dropout mask regenerated)."""
        dt = config.operand_dtype
        seed, site0, p = sv["drop"]
        x2, xn, qkv, o, x1, xn2, hpre, hpost = (sv[k] for k in ("x2", "xn", "qkv", "o", "x1", "xn2", "hpre", "hpost"))
        R, D = x2.shape
        a, m = self.attn, self.mlp
        H = a.num_heads
        pre = prefix + "." if prefix else ""
        # ---- MLP branch: out = x1 + drop(fc2(drop(gelu(fc1(LN2 x1))))) ----
        dy = ops.dropout_f32(dres, seed, site0 + self.SITE_FC2, p)
        d16, cs = ops.cast_colsum(dy, dt)
        self._linear_bwd(pre + "mlp.fc2", m.fc2, None, None, d16, cs, hpost, inv_scale, grads)
        dh = ops.gemm(d16, self._wT16("fc2T", m.fc2.weight, None))
        ops.dropout_t16(dh, seed, site0 + self.SITE_GELU, p, rescale=True)
        dh = ops.gelu16(hpre, dh)
        self._linear_bwd(pre + "mlp.fc1", m.fc1, None, None, dh, ops.colsum(dh), xn2, inv_scale, grads)
        dln = ops.gemm(dh, self._wT16("fc1T", m.fc1.weight), out_f32=True)
        dx1, part = ops.layernorm_bwd(dln, x1, self._f32("n2w", self.norm2.weight), self.norm2.eps, res=dres)
        red = ops.reduce_rows(part.view(part.shape[0], 2 * D), inv_scale)
        grads[pre + "norm2.weight"].copy_(red[:D]); grads[pre + "norm2.bias"].copy_(red[D:])
        # ---- attention branch: x1 = x + drop(proj(drop(softmax(q k^T)) v)) ----
        dy = ops.dropout_f32(dx1, seed, site0 + self.SITE_PROJ, p)
        d16, cs = ops.cast_colsum(dy, dt)
        self._linear_bwd(pre + "attn.proj", a.proj, None, None, d16, cs, o, inv_scale, grads)
        dO = ops.gemm(d16, self._wT16("projT", a.proj.weight, None))
        (B, N), = sv["segs"]
        ld = ops.token_ld(N)
        q3, dO3 = qkv.view(B, N, 3 * D), dO.view(B, N, D)
        dqkv = torch.empty((R, 3 * D), device=x2.device, dtype=dt)
        dq3 = dqkv.view(B, N, 3 * D)
        qt, kt, dOt = ops.transpose_tokens(qkv[:, :D], B, N), ops.transpose_tokens(qkv[:, D:2 * D], B, N), ops.transpose_tokens(dO, B, N)
        P4, Pd4 = sv["P"].view(H, B, N, ld), sv["Pd"].view(H, B, N, ld)
        dPd = torch.empty((H, B, N, ld), device=x2.device, dtype=torch.float32)
        for h in range(H):                      # d(dropped probabilities) = dO_h v_h^T
            ops.gemm(dO3[:, :, h * 64:(h + 1) * 64], q3[:, :, 2 * D + h * 64:2 * D + (h + 1) * 64], out=dPd[h, :, :, :N], out_f32=True)
        dS = ops.softmax_dropout_bwd(sv["P"], sv["Pd"], dPd.view(H * B * N, ld), N, a.scale, seed, site0 + self.SITE_ATTN, p)
        del dPd
        dS4 = dS.view(H, B, N, ld)
        for h in range(H):
            hs = slice(h * 64, (h + 1) * 64)
            PdT = ops.transpose_tokens(Pd4[h].reshape(B * N, ld), B, N)[:, :N, :]       # [B, keys, ld(queries)]
            ops.gemm(PdT, dOt[:, hs, :], out=dq3[:, :, 2 * D + h * 64:2 * D + (h + 1) * 64])          # dV = Pd^T dO
            ops.gemm(dS4[h], kt[:, hs, :], out=dq3[:, :, hs])                                          # dQ = dS K   (scale inside dS)
            dST = ops.transpose_tokens(dS4[h].reshape(B * N, ld), B, N)[:, :N, :]
            ops.gemm(dST, qt[:, hs, :], out=dq3[:, :, D + h * 64:D + (h + 1) * 64])                    # dK = dS^T Q
        self._linear_bwd(pre + "attn.qkv", a.qkv, None, None, dqkv, ops.colsum(dqkv), xn, inv_scale, grads)
        dln = ops.gemm(dqkv, self._wT16("qkvT", a.qkv.weight), out_f32=True)
        dx, part = ops.layernorm_bwd(dln, x2, self._f32("n1w", self.norm1.weight), self.norm1.eps, res=dx1)
        red = ops.reduce_rows(part.view(part.shape[0], 2 * D), inv_scale)
        grads[pre + "norm1.weight"].copy_(red[:D]); grads[pre + "norm1.bias"].copy_(red[D:])
        return dx

    def backward(self, saved, dres, inv_scale, grads=None, prefix=""):
        if isinstance(saved, dict):
            return self._bwd_dropout(saved, dres, inv_scale, grads, prefix)
        return super().backward(saved, dres, inv_scale, grads, prefix)

    # ---- split-precision forward ---------------------------------------------------------------------------------------
    # The head's output goes through mask_norm, a LayerNorm over only n_cls values that amplifies a relative error of the
    # cosines 2.2-2.8x (MaskTransformer._forward_core).  With the three GEMMs outside the blocks on hi + lo operands, the 16-bit
    # rounding of the blocks' GEMM operands is what is left (2.9e-4 on the tokens after two blocks at the reference's init
    # scales -> 1.4e-3 on the masks), so under ``config.split_conv`` the head's blocks run their four linear layers on
    # split-precision operands too (LayerNorm / GELU outputs as hi + lo pairs, weights as hi + lo pairs); only the fused
    # attention keeps 16-bit q, k, v, P and its 16-bit output.  Two layers of d_model: 3x the GEMM passes of a head that is a
    # few per cent of the backbone.  The saved activations keep the 16-bit layout ``Block.backward`` expects.
    def _lo(self, owner, key: str, w: torch.Tensor):
        return _pack(owner._cache, key + ".lo2", w,
                     lambda q: ops.cast_pad(q.reshape(q.shape[0], -1).contiguous().float(), dtype=config.operand_dtype, part=1))

    @staticmethod
    def _split_linear(a_hi, a_lo, w_hi, w_lo, bias, res=None):
        out = torch.empty((a_hi.shape[0], w_hi.shape[0]), device=a_hi.device, dtype=torch.float32)
        ops.gemm_split(a_hi, a_lo, w_hi, w_lo, out=out, bias_n=bias)
        if res is not None:
            o3 = out.view(1, *out.shape)
            ops.add_f32(o3, res.view(1, *res.shape), out=o3)
        return out

    def _fwd_precise(self, x2: torch.Tensor, segs, save: bool):
        dt = config.operand_dtype
        R, D = x2.shape
        a, m = self.attn, self.mlp
        xn32 = ops.layernorm(x2, self._f32("n1w", self.norm1.weight), self._f32("n1b", self.norm1.bias), self.norm1.eps, torch.float32)
        xn, xn_lo = ops.cast_pad(xn32, D, dt), ops.cast_pad(xn32, D, dt, part=1)
        qkv32 = self._split_linear(xn, xn_lo, a._w16("qkv", a.qkv.weight), self._lo(a, "qkv", a.qkv.weight), a._f32("qkv_b", a.qkv.bias))
        qkv = ops.cast_pad(qkv32, 3 * D, dt)
        o = torch.empty((R, D), device=x2.device, dtype=dt)
        lse = []
        r0 = 0
        for B, N in segs:
            r1 = r0 + B * N
            vt = ops.transpose_tokens(qkv[r0:r1, 2 * D:], B, N)
            l = torch.empty((B, a.num_heads, N), device=x2.device, dtype=torch.float32) if save else None
            ops.attention_fwd(qkv[r0:r1, :D], qkv[r0:r1, D:2 * D], vt, B, a.num_heads, N, a.scale, out=o[r0:r1], lse=l)
            lse.append(l)
            r0 = r1
        if r0 != R:
            raise ValueError("masktrans Block: segments do not cover the rows")
        x1 = ops.gemm(o, a._w16("proj", a.proj.weight), out_f32=True, bias_n=a._f32("proj_b", a.proj.bias), res=x2,
                      b_lo=self._lo(a, "proj", a.proj.weight))
        xn2_32 = ops.layernorm(x1, self._f32("n2w", self.norm2.weight), self._f32("n2b", self.norm2.bias), self.norm2.eps, torch.float32)
        xn2, xn2_lo = ops.cast_pad(xn2_32, D, dt), ops.cast_pad(xn2_32, D, dt, part=1)
        hpre32 = self._split_linear(xn2, xn2_lo, m._w16("fc1", m.fc1.weight), self._lo(m, "fc1", m.fc1.weight), m._f32("fc1_b", m.fc1.bias))
        hpost, hpost_lo = ops.gelu_split(hpre32, dt)
        x3 = self._split_linear(hpost, hpost_lo, m._w16("fc2", m.fc2.weight), self._lo(m, "fc2", m.fc2.weight),
                                m._f32("fc2_b", m.fc2.bias), res=x1)
        if not save:
            return x3, None
        hpre = ops.cast_pad(hpre32, hpre32.shape[1], dt)
        return x3, (x2, xn, qkv, o, lse, x1, xn2, hpre, hpost, list(segs))

    def forward_rows(self, x2, segs):
        if config.split_conv:
            return self._fwd_precise(x2, segs, False)[0]
        return super().forward_rows(x2, segs)

    def forward_train_rows(self, x2, segs):
        if self.dropout > 0.0 and self._drop is not None:
            return self._fwd_dropout(x2, segs)
        if config.split_conv:
            return self._fwd_precise(x2, segs, True)
        return super().forward_train_rows(x2, segs)


def init_weights(m):
    """`eval_dinov2_masktrans.py:388-396`"""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class _MaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        logits, saved = module._forward_core(x.detach(), save=True)
        ctx.module, ctx.saved = module, saved
        ctx.names = [n for n, _ in module.named_parameters()]
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dmasks):
        m = ctx.module
        S = m.loss_scale
        B, C, h, w = dmasks.shape
        d = (dmasks.permute(0, 2, 3, 1).contiguous().float() * S).view(B * h * w, C)
        grads = {n: torch.empty_like(p) for n, p in m.named_parameters()}
        m._backward_core(ctx.saved, d, 1.0 / S, grads)
        ctx.saved = None
        return (None, None) + tuple(grads[n] for n in ctx.names)


class MaskTransformer(_Packed):
    def __init__(self, n_cls, patch_size, d_encoder, n_layers, n_heads, d_model, d_ff, drop_path_rate, dropout):
        super().__init__()
        if not (0 < n_cls <= 16):
            raise ValueError("MaskTransformer: 1..16 classes (csrc/maskhead.hip keeps one accumulator per class)")
        if d_model % 8 or d_encoder % 8:
            raise ValueError("MaskTransformer: d_model and d_encoder must be multiples of 8")
        self.d_encoder, self.patch_size, self.n_layers, self.n_cls = d_encoder, patch_size, n_layers, n_cls
        self.d_model, self.d_ff = d_model, d_ff
        self.num_classes = n_cls
        self.scale = d_model ** -0.5
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, n_layers)]
        self.blocks = nn.ModuleList([Block(d_model, n_heads, d_ff, dropout, dpr[i]) for i in range(n_layers)])
        self.cls_emb = nn.Parameter(torch.randn(1, n_cls, d_model))
        self.proj_dec = nn.Linear(d_encoder, d_model)
        self.proj_patch = nn.Parameter(self.scale * torch.randn(d_model, d_model))
        self.proj_classes = nn.Parameter(self.scale * torch.randn(d_model, d_model))
        self.decoder_norm = nn.LayerNorm(d_model)
        self.mask_norm = nn.LayerNorm(n_cls)
        # Loss scale of this head's 16-bit backward operands.  mask_norm's backward multiplies the mask gradients by up to
        # 1 / sqrt(eps) = 316, so the package-wide 2^16 (config.loss_scale) overflows fp16 here (the optimizer's guard then
        # skips the step); 2^10 keeps the largest operand around 1e4 and the smallest useful ones far above fp16's 6e-8.
        self.loss_scale = 1024.0
        # dropout (training mode only): every forward draws the masks of step `drop_step` of stream `drop_seed`
        self.dropout = float(dropout)
        self.drop_seed, self.drop_step = 0x5EED, 0
        self.apply(init_weights)
        nn.init.trunc_normal_(self.cls_emb, std=0.02)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"cls_emb"}

    GRAD_ORDER = ("mask_norm", "proj_patch", "proj_classes", "decoder_norm", "blocks", "cls_emb", "proj_dec")

    # ---- functional core ---------------------------------------------------------------------------------------------
    def _right(self, key: str, p: nn.Parameter, transpose: bool, part: int = 0):
        """16-bit B operand of ``x @ p`` (transpose: rows = output features) or of its input gradient ``dy @ p^T``;
        part 1 = the rounding residual of part 0."""
        dt = config.operand_dtype
        return _pack(self._cache, key, p,
                     lambda q: ops.cast_pad((q.float().t() if transpose else q.float()).contiguous(), dtype=dt, part=part))

    def _lo(self, key: str, w: torch.Tensor):
        return _pack(self._cache, key, w,
                     lambda q: ops.cast_pad(q.reshape(q.shape[0], -1).contiguous().float(), dtype=config.operand_dtype, part=1))

    def _forward_core(self, tok: torch.Tensor, save: bool):
        """tok fp32 [B, N, d_encoder] -> masks fp32 NHWC [B, GS, GS, n_cls] (GS = sqrt(N)), saved state."""
        dt = config.operand_dtype
        B, N, De = tok.shape
        if De != self.d_encoder:
            raise ValueError(f"MaskTransformer: tokens have {De} features, built for d_encoder={self.d_encoder}")
        GS = int(round(math.sqrt(N)))
        if GS * GS != N:
            raise ValueError("MaskTransformer: the reference reshapes the N patches to a square (h = w = H // patch_size)")
        if save:
            for blk in self.blocks:
                blk.check_train()
        C, D = self.n_cls, self.d_model
        RB = N + C
        R = B * RB
        t_all = torch.zeros((B, RB, De), device=tok.device, dtype=torch.float32)
        t_all[:, :N] = tok.float()
        a16 = ops.cast_pad(t_all.view(R, De), De, dt)
        pd = self.proj_dec
        sp = config.split_conv    # hi + lo operands for the three GEMMs outside the blocks (see the note on mask_norm below)
        x = torch.empty((R, D), device=tok.device, dtype=torch.float32)
        if sp:
            ops.gemm_split(a16, ops.cast_pad(t_all.view(R, De), De, dt, part=1), self._w16("pd", pd.weight),
                           self._lo("pd_lo", pd.weight), out=x, bias_n=self._f32("pd_b", pd.bias))
        else:
            ops.gemm(a16, self._w16("pd", pd.weight), out=x, bias_n=self._f32("pd_b", pd.bias))
        x.view(B, RB, D)[:, N:] = self.cls_emb.detach().float()           # the class tokens behind every image's patches
        saves = []
        drop = save and self.training and self.dropout > 0.0
        if drop:
            self.drop_step += 1
        for i, blk in enumerate(self.blocks):
            blk._drop = (self._step_seed(), 4 * i) if drop else None
            if save:
                x, sv = blk.forward_train_rows(x, [(B, RB)])
                saves.append(sv)
            else:
                x = blk.forward_rows(x, [(B, RB)])
        dn = self.decoder_norm
        # mask_norm is a LayerNorm over only n_cls values (eps 1e-5): for two classes nearly a sign function of the cosine
        # difference, and it amplifies a relative error of the cosines 2.2-2.8x (scripts/masktrans_probe.py).  The GEMMs that
        # feed the cosines directly therefore run on split-precision operands (three small GEMMs of the head).
        Pall = torch.empty((R, D), device=tok.device, dtype=torch.float32)
        Call = torch.empty((R, D), device=tok.device, dtype=torch.float32)
        if sp:
            xd32 = ops.layernorm(x, self._f32("dn_w", dn.weight), self._f32("dn_b", dn.bias), dn.eps, torch.float32)
            xd16, xd_lo = ops.cast_pad(xd32, D, dt), ops.cast_pad(xd32, D, dt, part=1)
            ops.gemm_split(xd16, xd_lo, self._right("pp_f", self.proj_patch, True), self._right("pp_flo", self.proj_patch, True, 1),
                           out=Pall)
            ops.gemm_split(xd16, xd_lo, self._right("pc_f", self.proj_classes, True), self._right("pc_flo", self.proj_classes, True, 1),
                           out=Call)
        else:
            xd16 = ops.layernorm(x, self._f32("dn_w", dn.weight), self._f32("dn_b", dn.bias), dn.eps, dt)
            ops.gemm(xd16, self._right("pp_f", self.proj_patch, True), out=Pall)
            ops.gemm(xd16, self._right("pc_f", self.proj_classes, True), out=Call)
        chat, inv_c = ops.cls_l2norm(Call, B, N, C)
        mn = self.mask_norm
        logits, cosm, inv_p = ops.mask_logits_fwd(Pall, chat, self._f32("mn_w", mn.weight), self._f32("mn_b", mn.bias), mn.eps, N)
        saved = None
        if save:
            saved = dict(a16=a16, saves=saves, x=x, xd16=xd16, Pall=Pall, chat=chat, inv_c=inv_c, cosm=cosm, inv_p=inv_p,
                         geom=(B, N, C, D, De))
        return logits.view(B, GS, GS, C), saved

    def _backward_core(self, saved, dlogits: torch.Tensor, inv_scale: float, grads, stage_done=None):
        """dlogits fp32 [B*N, n_cls] = loss_scale * dL/dmasks -> every parameter gradient (unscaled) into ``grads``."""
        dt = config.operand_dtype
        B, N, C, D, De = saved["geom"]
        RB = N + C
        R = B * RB

        def done():
            if stage_done is not None:
                stage_done()

        mn, dn = self.mask_norm, self.decoder_norm
        dP = torch.zeros((R, D), device=dlogits.device, dtype=torch.float32)
        dcos, part = ops.mask_logits_bwd(dlogits, saved["cosm"], saved["inv_p"], saved["Pall"], saved["chat"],
                                         self._f32("mn_w", mn.weight), mn.eps, dP, N)
        red = ops.reduce_rows(part.view(part.shape[0], 2 * C), inv_scale)
        grads["mask_norm.weight"].copy_(red[:C]); grads["mask_norm.bias"].copy_(red[C:])
        done()
        dchat = ops.mask_dchat(dcos, saved["Pall"], saved["inv_p"], B, N, C)
        dCl = torch.zeros((R, D), device=dlogits.device, dtype=torch.float32)
        ops.cls_l2norm_bwd(dchat, saved["chat"], saved["inv_c"], dCl, N)
        dP16, dC16 = ops.cast_pad(dP, D, dt), ops.cast_pad(dCl, D, dt)
        xd16 = saved["xd16"]
        # d(x @ p) / dp = x^T dy: the 1x1 "weight gradient" with x in the dy role -> [D_in, D_out] = the parameter's layout
        ops.wgrad(xd16.view(1, R, 1, D), dP16.view(1, R, 1, D), D, 1, 1, 1, 0, inv_scale, out=grads["proj_patch"].view(D, D, 1, 1))
        done()
        ops.wgrad(xd16.view(1, R, 1, D), dC16.view(1, R, 1, D), D, 1, 1, 1, 0, inv_scale, out=grads["proj_classes"].view(D, D, 1, 1))
        done()
        dxd = ops.gemm(dP16, self._right("pp_b", self.proj_patch, False), out_f32=True)
        dxd = ops.gemm(dC16, self._right("pc_b", self.proj_classes, False), out_f32=True, res=dxd)
        dx, part = ops.layernorm_bwd(dxd, saved["x"], self._f32("dn_w", dn.weight), dn.eps)
        red = ops.reduce_rows(part.view(part.shape[0], 2 * D), inv_scale)
        grads["decoder_norm.weight"].copy_(red[:D]); grads["decoder_norm.bias"].copy_(red[D:])
        done()
        for i in range(len(self.blocks) - 1, -1, -1):
            dx = self.blocks[i].backward(saved["saves"][i], dx, inv_scale, grads, f"blocks.{i}")
        done()
        # class rows -> cls_emb (summed over the batch); all rows -> proj_dec (the class rows of its operand are zero)
        dcls = dx.view(B, RB, D)[:, N:].reshape(B, C * D).contiguous()
        ops.reduce_rows(dcls, inv_scale, grads["cls_emb"].view(C * D))
        done()
        d16, cs = ops.cast_colsum(dx, dt)
        pd = self.proj_dec
        gw = ops.wgrad(d16.view(1, R, 1, D), saved["a16"].view(1, R, 1, De), D, 1, 1, 1, 0, inv_scale)
        grads["proj_dec.weight"].copy_(gw.view(D, De))
        ops.reduce_rows(cs, inv_scale, grads["proj_dec.bias"])
        grads["proj_dec.bias"].sub_(grads["cls_emb"].view(C, D).sum(0))          # the bias does not reach the class rows
        done()

    def _step_seed(self) -> int:
        return ((int(self.drop_seed) & 0xFFFFFFFF) << 32) | (int(self.drop_step) & 0xFFFFFFFF)

    def dropout_masks(self, B: int, N: int, device):
        """Keep masks of the LAST training forward (test infrastructure: the oracle replays them) for ``B`` images of ``N``
        patches: per block a dict attn bool [B, H, T, T], proj [B, T, D], gelu [B, T, d_ff], fc2 [B, T, D], T = N + n_cls."""
        T = N + self.n_cls
        ld = ops.token_ld(T)
        seed, p, out = self._step_seed(), self.dropout, []
        for i, blk in enumerate(self.blocks):
            H, D, F_ = blk.attn.num_heads, self.d_model, blk.mlp.fc1.out_features
            s0 = 4 * i
            am = ops.dropout_mask(H * B * T * ld, seed, s0 + Block.SITE_ATTN, p, device).view(H, B, T, ld)[..., :T].permute(1, 0, 2, 3)
            out.append(dict(attn=am.bool(), proj=ops.dropout_mask(B * T * D, seed, s0 + Block.SITE_PROJ, p, device).view(B, T, D).bool(),
                            gelu=ops.dropout_mask(B * T * F_, seed, s0 + Block.SITE_GELU, p, device).view(B, T, F_).bool(),
                            fc2=ops.dropout_mask(B * T * D, seed, s0 + Block.SITE_FC2, p, device).view(B, T, D).bool()))
        return out

    # ---- reference-shaped entry point ----------------------------------------------------------------------------------
    def forward(self, x, im_size):
        """`eval_dinov2_masktrans.py:441-462`: tokens (B, N, d_encoder), im_size (H, W) -> masks (B, n_cls, H/P, W/P)."""
        H, W = im_size
        GS = H // self.patch_size
        if x.dim() != 3 or x.shape[1] != GS * GS:
            raise ValueError(f"MaskTransformer: expected (B, {GS * GS}, d_encoder) tokens for im_size {tuple(im_size)}")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _MaskFn.apply(self, x, *list(self.parameters()))
        masks, _ = self._forward_core(x.detach(), save=False)
        return masks.permute(0, 3, 1, 2)


class MaskTransEngine(nn.Module):
    """The training step of `eval/eval_dinov2_masktrans.py:262-331` with a frozen ViT:

        tokens = cat of the patch tokens of the last n blocks along the feature axis (`:274-277`, no_grad)
        masks = seg_decoder(tokens, (H, W)) ; bilinear resize to the image (`:296-298`)
        loss = CrossEntropy(weight [0.1, 10]) + (1 - dice of the arg-max prediction)      (`:303-312`; the second term is a
               constant of the step — arg-max has no gradient — and is reported, for two classes, exactly as the script does)
        backward through the head ; SGD on the head (`:211-218`)
    """

    def __init__(self, model, seg_decoder: MaskTransformer, *, n_last_blocks: int = 1, lr: float = 0.01, momentum: float = 0.9,
                 weight_decay: float = 0.0, ce_weight=(0.1, 10.0), process_group=None):
        super().__init__()
        from ..optim import SGD, FlatBucket
        from ..parallel import StageReducer
        self.model, self.seg_decoder, self.n, self.process_group = model, seg_decoder, n_last_blocks, process_group
        for p in model.parameters():
            p.requires_grad_(False)
        if seg_decoder.d_encoder != n_last_blocks * model.embed_dim:
            raise ValueError("MaskTransEngine: d_encoder must be n_last_blocks * embed_dim (`:276-277` concatenates the layers)")
        named = dict(seg_decoder.named_parameters())
        order = list(seg_decoder.GRAD_ORDER)
        match = lambda n, pre: n == pre or n.startswith(pre + ".")
        ordered = [(n, named[n]) for pre in order for n in named if match(n, pre)]
        assert len(ordered) == len(named)
        self.bucket = FlatBucket(ordered)
        self.stage_ranges = [self.bucket.range_of([n for n in named if match(n, pre)]) for pre in order]
        self.optimizer = SGD([self.bucket], lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.reducer = StageReducer(self.bucket.grad, self.stage_ranges, process_group)
        dev = next(seg_decoder.parameters()).device
        self.register_buffer("ce_weight", torch.tensor(ce_weight, dtype=torch.float32, device=dev) if ce_weight is not None else None)

    @torch.no_grad()
    def tokens(self, inp: torch.Tensor) -> torch.Tensor:
        outs = self.model.get_intermediate_layers(inp, self.n)
        return outs[0] if len(outs) == 1 else torch.cat(list(outs), dim=-1)

    @torch.no_grad()
    def train_step(self, inp: torch.Tensor, target: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        from ..parallel import world_size
        dec = self.seg_decoder
        S = dec.loss_scale
        tok = self.tokens(inp)
        logits, saved = dec._forward_core(tok, save=True)
        B, gs, _, C = logits.shape
        target = target.long().contiguous()
        cw = self.ce_weight
        loss, coef, _ = ops.seg_loss_fwd(logits, target, 0, ops.LOSS_NONE, 0.0, 1, cw, S)
        dz = ops.seg_loss_bwd(logits, target, coef, 0, ops.LOSS_NONE, 1, cw)
        dl, _ = ops.resize_bilinear_bwd(dz, gs, gs, torch.float32)
        inv = 1.0 / (S * world_size(self.process_group))
        self.reducer.begin()
        dec._backward_core(saved, dl.view(B * gs * gs, C), inv, self.bucket.views, stage_done=self.reducer.stage_done)
        self.reducer.finish()
        self.optimizer.step(1.0)
        loss = loss.view(())
        if C == 2:   # + dice_loss(preds, target) of the hard prediction (`:83-92,306-311`), from the arg-max pixel counts
            _, cnt = ops.ce_acc(logits, target, cw, counts=True)
            c1 = cnt[1].float()
            loss = loss + (1.0 - (2.0 * c1[2] + 1e-7) / (c1[1] + c1[0] + 1e-7))
        if taps is not None:
            taps.update(logits=logits, tokens=tok)
        return loss
