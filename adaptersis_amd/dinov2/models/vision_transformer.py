"""HIP-backed ``DinoVisionTransformer`` — API and ``state_dict`` mirror of
`dinov2/models/vision_transformer.py:42-357` for the frozen-backbone path of AdapterSIS
(`train.py:287,300-302`): ``patch_embed``, ``blocks[i](x)``, ``get_intermediate_layers``,
``forward_features`` / ``forward(is_training=True)``.

Only what the hot path uses is implemented: ``block_chunks=0`` (flat ``blocks.{i}`` keys, as in
`dinov2/configs/ssl_default_config.yaml`), no masks, no register tokens, drop_path 0.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from ... import config, ops
from ..layers.blocks import _pack
from ..layers import MemEffAttention, Mlp, NestedTensorBlock as Block, PatchEmbed, SwiGLUFFNFused


class DinoVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, ffn_bias=True, proj_bias=True, drop_path_rate=0.0,
                 drop_path_uniform=False, init_values=None, embed_layer=PatchEmbed, act_layer=nn.GELU,
                 block_fn=Block, ffn_layer="mlp", block_chunks=0, num_register_tokens=0,
                 interpolate_antialias=False, interpolate_offset=0.1):
        super().__init__()
        if block_chunks not in (0, None):
            raise ValueError("block_chunks must be 0 on this path (dinov2/configs/ssl_default_config.yaml)")
        if drop_path_rate or num_register_tokens or interpolate_antialias:
            raise ValueError("drop_path / register tokens / antialias are not part of the AdapterSIS path")
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.n_blocks = depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.interpolate_offset = interpolate_offset
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        if ffn_layer == "mlp":
            ffn = Mlp
        elif ffn_layer in ("swiglufused", "swiglu"):
            ffn = SwiGLUFFNFused
        else:
            raise NotImplementedError(ffn_layer)
        self.blocks = nn.ModuleList([
            block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, proj_bias=proj_bias,
                     ffn_bias=ffn_bias, drop_path=0.0, norm_layer=norm_layer, act_layer=act_layer, ffn_layer=ffn,
                     init_values=init_values)
            for _ in range(depth)])
        self.chunked_blocks = False
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        self._pos_cache = {}

    # -- vision_transformer.py:164-188 ---------------------------------------------------------
    def interpolate_pos_encoding(self, x, w, h):
        return self._pos_for(x.shape[1] - 1, w, h)

    def _pos_for(self, npatch, w, h):
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        key = (w, h, self.pos_embed.data_ptr(), self.pos_embed._version, self.pos_embed.device)
        if key in self._pos_cache:
            return self._pos_cache[key]
        # Constant per (H, W): computed once and cached.  Keeps the reference's
        # scale_factor quirk: the bicubic grid uses (w0+0.1)/sqrt(N), not w0/sqrt(N).
        with torch.no_grad():
            pos_embed = self.pos_embed.float()
            class_pos_embed = pos_embed[:, 0]
            patch_pos_embed = pos_embed[:, 1:]
            dim = self.embed_dim
            w0, h0 = w // self.patch_size + self.interpolate_offset, h // self.patch_size + self.interpolate_offset
            s = int(math.sqrt(N))
            patch_pos_embed = nn.functional.interpolate(
                patch_pos_embed.reshape(1, s, s, dim).permute(0, 3, 1, 2),
                scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
            assert int(w0) == patch_pos_embed.shape[-2] and int(h0) == patch_pos_embed.shape[-1]
            patch_pos_embed = patch_pos_embed.permute(0, 2, 3, 1).reshape(1, -1, dim)
            out = torch.cat((class_pos_embed.unsqueeze(0), patch_pos_embed), dim=1).contiguous()
        self._pos_cache = {key: out}
        return out

    # -- vision_transformer.py:190-199 ---------------------------------------------------------
    def prepare_tokens_with_masks(self, x, masks=None):
        if masks is not None:
            raise ValueError("masks are a DINOv2 pre-training feature, not part of the AdapterSIS path")
        B, nc, w, h = x.shape
        t = self.patch_embed(x)
        pos = self._pos_for(t.shape[1], w, h)
        return ops.add_cls_pos(t, self.cls_token.detach().reshape(-1).float().contiguous(),
                               pos.detach().reshape(-1, self.embed_dim).float().contiguous())

    def _final_norm(self, x: torch.Tensor) -> torch.Tensor:
        return ops.layernorm(x, self.norm.weight.detach().float(), self.norm.bias.detach().float(), self.norm.eps,
                             torch.float32)

    # -- vision_transformer.py:212-235 ---------------------------------------------------------
    def forward_features(self, x, masks=None):
        x = self.prepare_tokens_with_masks(x, masks)
        for blk in self.blocks:
            x = blk(x)
        x_norm = self._final_norm(x)
        return {"x_norm_clstoken": x_norm[:, 0], "x_norm_patchtokens": x_norm[:, 1:], "x_prenorm": x, "masks": masks}

    def patch_tokens_train(self, x: torch.Tensor):
        """PatchEmbed keeping its im2col operand: -> (tokens fp32 (B, N, D), a16 16-bit [B*N, ldk]) for the weight gradient."""
        return self.patch_embed.tokens(x)

    # -- training path: forward_features under autograd (eval_dinov2_setr_cross_ete.py:145-148,318-321) -----------
    def forward_train(self, x: torch.Tensor):
        """images (B,3,H,W) -> (x_norm_patchtokens fp32 (B,N,D) view, saved activations for ``backward``)."""
        B, nc, w, h = x.shape
        t, a16 = self.patch_tokens_train(x)
        pos = self._pos_for(t.shape[1], w, h)
        xx = ops.add_cls_pos(t, self.cls_token.detach().reshape(-1).float().contiguous(),
                             pos.detach().reshape(-1, self.embed_dim).float().contiguous())
        saved = []
        for blk in self.blocks:
            xx, s = blk.forward_train(xx)
            saved.append(s)
        x_norm = self._final_norm(xx)
        return x_norm[:, 1:], (a16, xx, saved, (B, t.shape[1], w, h))

    def _pos_resize_matrix(self, w: int, h: int, dev):
        """The bicubic pos-embed resize of ``_pos_for`` as a dense matrix Mt [N_orig, N_new] (it is a fixed linear map
        per image size): d pos_embed[1:] = Mt @ d pos_interp[1:].  Built once per size with the same F.interpolate
        call, split into 16-bit hi/lo halves for the split-precision GEMM."""
        key = (w, h, dev, config.operand_dtype)
        cache = self.__dict__.setdefault("_posm_cache", {})
        hit = cache.get(key)
        if hit is None:
            N = self.pos_embed.shape[1] - 1
            s = int(math.sqrt(N))
            w0, h0 = w // self.patch_size + self.interpolate_offset, h // self.patch_size + self.interpolate_offset
            with torch.no_grad():
                eye = torch.eye(N).reshape(N, 1, s, s)
                Mt = nn.functional.interpolate(eye, scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
                Mt = Mt.reshape(N, -1).contiguous().to(dev)
            ld = (Mt.shape[1] + 7) // 8 * 8
            hit = (ops.cast_pad(Mt, ld, config.operand_dtype), ops.cast_pad(Mt, ld, config.operand_dtype, part=1))
            cache[key] = hit
        return hit

    def backward(self, saved, dtok: torch.Tensor, inv_scale: float, grads: dict, block_done=None) -> None:
        """dtok fp32 (B,N,D) = loss_scale * dL/d x_norm_patchtokens -> ``grads[name]`` (fp32, unscaled) for every
        parameter of the model (``mask_token`` is unused on this path: zero).  ``block_done(i)`` is called when block
        i's gradients are enqueued (i = depth .. 0, depth = final norm) so a reducer can start that bucket."""
        a16, x_last, bsaved, (B, N, w, h) = saved
        D = self.embed_dim
        dev = dtok.device
        dt = config.operand_dtype
        dxn = torch.zeros((B, N + 1, D), device=dev, dtype=torch.float32)
        ops.copy_channels(dtok.reshape(B, N * D), dxn.view(B, (N + 1) * D)[:, D:])
        dx, part = ops.layernorm_bwd(dxn.view(-1, D), x_last.reshape(-1, D), self.norm.weight.detach().float().contiguous(),
                                     self.norm.eps)
        red = ops.reduce_rows(part.view(part.shape[0], 2 * D), inv_scale)
        grads["norm.weight"].copy_(red[:D]); grads["norm.bias"].copy_(red[D:])
        if block_done is not None:
            block_done(len(self.blocks))
        for i in range(len(self.blocks) - 1, -1, -1):
            dx = self.blocks[i].backward(bsaved[i], dx, inv_scale, grads, f"blocks.{i}")
            bsaved[i] = None
            if block_done is not None:
                block_done(i)
        self.embed_backward(a16, dx.view(B, N + 1, D), None, w, h, inv_scale, grads)
        if block_done is not None:
            block_done(-1)

    def embed_backward(self, a16, dxa: torch.Tensor, dtb: Optional[torch.Tensor], w: int, h: int, inv_scale: float,
                       grads: dict) -> None:
        """Backward of ``prepare_tokens_with_masks`` + PatchEmbed: dxa fp32 (B, N+1, D) = gradient of the cls + pos-embed
        token matrix; dtb fp32 [B*N, D] or None = gradient that reaches the raw patch tokens directly (pass B of the
        adapter flow, `train.py:300`, which takes ``patch_embed(inp)`` without cls / pos-embed).  Writes cls_token,
        pos_embed, mask_token (unused: zero) and patch_embed.proj.{weight,bias} into ``grads``."""
        B, N1, D = dxa.shape
        N = N1 - 1
        dev = dxa.device
        dt = config.operand_dtype
        dx = dxa
        # ---- prepare_tokens: x = cat(cls, patch_embed(img)) + pos ----
        psum = ops.reduce_rows(ops.colsum(dx.view(B, (N + 1) * D)), inv_scale).view(N + 1, D)   # sum over the batch
        grads["cls_token"].view(-1).copy_(psum[0])
        gp = grads["pos_embed"].view(-1, D)
        gp[0].copy_(psum[0])
        if gp.shape[0] - 1 == N and w == h:
            gp[1:].copy_(psum[1:])
        else:
            m_hi, m_lo = self._pos_resize_matrix(w, h, dev)
            pt = psum[1:].t().contiguous()                                                      # [D, N]
            ld = m_hi.shape[1]
            ops.gemm_split(m_hi, m_lo, ops.cast_pad(pt, ld, dt), ops.cast_pad(pt, ld, dt, part=1), out=gp[1:])
        grads["mask_token"].zero_()
        dtk = torch.empty((B * N, D), device=dev, dtype=torch.float32)
        ops.copy_channels(dx.view(B, (N + 1) * D)[:, D:], dtk.view(B, N * D))
        if dtb is not None:
            ops.add_f32(dtk.view(B, N, D), dtb.view(B, N, D), out=dtk.view(B, N, D))
        d16 = ops.cast_pad(dtk, D, dt)
        K = grads["patch_embed.proj.weight"][0].numel()
        gw = ops.wgrad(d16.view(1, B * N, 1, D), a16.view(1, B * N, 1, a16.shape[1]), D, 1, 1, 1, 0, inv_scale)
        grads["patch_embed.proj.weight"].view(D, K).copy_(gw.view(D, -1)[:, :K])
        ops.reduce_rows(ops.colsum(dtk), inv_scale, grads["patch_embed.proj.bias"])

    # -- vision_transformer.py:237-247 ---------------------------------------------------------
    def _get_intermediate_layers_not_chunked(self, x, n=1):
        x = self.prepare_tokens_with_masks(x)
        output, total_block_len = [], len(self.blocks)
        blocks_to_take = range(total_block_len - n, total_block_len) if isinstance(n, int) else n
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i in blocks_to_take:
                output.append(x)
        assert len(output) == len(blocks_to_take), f"only {len(output)} / {len(blocks_to_take)} blocks found"
        return output

    # -- vision_transformer.py:263-287 ---------------------------------------------------------
    def get_intermediate_layers(self, x: torch.Tensor, n: Union[int, Sequence] = 1, reshape: bool = False,
                                return_class_token: bool = False, norm=True) -> Tuple:
        outputs = self._get_intermediate_layers_not_chunked(x, n)
        if norm:
            outputs = [self._final_norm(out) for out in outputs]
        class_tokens = [out[:, 0] for out in outputs]
        outputs = [out[:, 1:] for out in outputs]
        if reshape:
            B, _, w, h = x.shape
            outputs = [out.reshape(B, w // self.patch_size, h // self.patch_size, -1).permute(0, 3, 1, 2).contiguous()
                       for out in outputs]
        if return_class_token:
            return tuple(zip(outputs, class_tokens))
        return tuple(outputs)

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        if is_training:
            return ret
        return self.head(ret["x_norm_clstoken"])


def vit_tiny_test(patch_size=14, **kwargs):
    """Test-only geometry (D=128, 4 blocks, 2 heads of 64)."""
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=128, depth=4, num_heads=2, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_tiny_swiglu(patch_size=14, **kwargs):
    """Test-only: SwiGLU FFN (ViT-g style) at D=128."""
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=128, depth=4, num_heads=2, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_large_d4(patch_size=14, **kwargs):
    """Test-only: ViT-L width with 4 blocks."""
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=1024, depth=4, num_heads=16, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_base_d4(patch_size=14, **kwargs):
    """Test-only: ViT-B width with 4 blocks."""
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=768, depth=4, num_heads=12, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_giant2_d4(patch_size=14, **kwargs):
    """Test-only: ViT-g width (SwiGLU FFN when built with ffn_layer="swiglufused") with 4 blocks."""
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=1536, depth=4, num_heads=24, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_small(patch_size=16, **kwargs):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_base(patch_size=16, **kwargs):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_large(patch_size=16, **kwargs):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)


def vit_giant2(patch_size=16, **kwargs):
    """embed-dim 1536, 24 heads of 64 (`vision_transformer.py:344-357`)."""
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=1536, depth=40, num_heads=24, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), **kwargs)
