"""HIP-backed OR-UNet multi-scale fuse head — API / ``state_dict`` mirror of the ``UNet`` of
`eval/eval_dinov2_or_unet_fuse.py:426-486` with its ``FusionModel`` (`:502-510`) and ``FCUUp`` (`:511-530`; the same class
ships in `backbones/decoders.py:276-296`).  The reference script imports it from a module ``or_unet`` that the repository
does not contain; the classes live in the script itself.

    x1 = inc(img)                      DoubleConv(3, 64) at H x W
    x1 = fuser(x1, expand_block_4(x_t2, H, W))       ViT map of the image at scale 1.5
    x2 = fuser(down1(x1), expand_block_3(x_o, ...))  ViT map at scale 1
    x3 = fuser(down2(x2), expand_block_2(x_d2, ...)) ViT map at scale 0.5
    x4 = down3(x3);  x5 = down4(x4)
    y  = up1(x5, x4) -> up2(., x3) -> up3(., x2) -> up4(., x1);  logits = outc(y)

FCUUp = 1x1 conv (one GEMM over the ViT tokens, which ARE the NHWC map) -> BatchNorm2d(eps=1e-6, train statistics) -> ReLU
-> F.interpolate(size=(H, W)) in its default 'nearest' mode; FusionModel = add -> ReLU.  The resize, the add and the ReLU
are one in-place kernel on the UNet level's split-precision map (`asis_nearest_add_relu`); its backward is a pass-through
to the UNet branch and a block sum to the FCUUp branch (`asis_nearest_sum`).  Everything else is the machinery of the
config-2 UNet head (`unet_parts.py`): implicit-GEMM 3x3 convs with BatchNorm statistics from the epilogue, MaxPool2d with
arg-max bytes, ConvTranspose2d as GEMM + pixel shuffle into the concat buffer.  The first conv (3 input channels) is the
direct fp32 kernel of the CNN encoder's stem.  The ViT maps are inputs without gradient (`:281-306`: ``torch.no_grad()``).
"""
from __future__ import annotations

from functools import partial
from typing import Optional

import torch
import torch.nn as nn

from .. import config, ops, parallel
from ..dinov2.layers.blocks import _Packed, _pack
from . import _bn
from . import unet_parts as P


class FusionModel(nn.Module):
    """`eval_dinov2_or_unet_fuse.py:502-510` — no parameters; runs fused inside ``UNet`` (asis_nearest_add_relu)."""

    def __init__(self):
        super().__init__()
        self.activation = nn.ReLU()

    def forward(self, x, x1):
        raise RuntimeError("FusionModel runs fused with FCUUp inside or_unet.UNet (asis_nearest_add_relu); call the UNet")


class FCUUp(nn.Module):
    """Transformer patch embeddings -> CNN feature maps (`eval_dinov2_or_unet_fuse.py:511-530`) — parameter container."""

    def __init__(self, inplanes, outplanes, up_stride, act_layer=nn.ReLU, norm_layer=partial(nn.BatchNorm2d, eps=1e-6)):
        super().__init__()
        self.up_stride = up_stride
        self.conv_project = nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=1, padding=0)
        self.bn = norm_layer(outplanes)
        self.act = act_layer()

    def forward(self, x_r, H, W):
        raise RuntimeError("FCUUp runs fused with FusionModel inside or_unet.UNet; call the UNet")


class _FuseFn(torch.autograd.Function):
    """autograd bridge: parameters get gradients, the image and the ViT maps do not (reference: no_grad features)."""

    @staticmethod
    def forward(ctx, module, x, x_o, x_t2, x_d2, *params):
        logits, saved = module._forward_core(x, [module._map16(t) for t in (x_o, x_t2, x_d2)], save=True)
        ctx.module, ctx.saved = module, saved
        ctx.names = [n for n, _ in module.named_parameters()]
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.module
        dt = config.operand_dtype
        S = config.loss_scale
        B, C, h, w = dlogits.shape
        d = dlogits.permute(0, 2, 3, 1).contiguous().float().view(B * h * w, C)
        CP = (C + 7) // 8 * 8
        d16 = ops.cast_pad(d, CP, dt, scale=S).view(B, h, w, CP)
        d_lo = ops.cast_pad(d, CP, dt, scale=S, part=1).view(B, h, w, CP) if config.split_conv else None
        grads = {n: torch.empty_like(p) for n, p in m.named_parameters()}
        m._backward_core(ctx.saved, d16, None, 1.0 / S, grads, dlogits_f32=d, d_lo=d_lo)
        ctx.saved = None
        if d16.is_cuda:
            parallel.join_grad_streams()     # weight gradients computed on the side stream (config.wgrad_stream)
        return (None, None, None, None, None) + tuple(grads[n] for n in ctx.names)


class UNet(P.UNet):
    """``UNet(n_channels=3, n_classes=2, outplanes=1024, embed_dim=384, dw_stride=1, bilinear=False)`` of the reference script;
    ``base`` (64 in the reference) scales every width for small test geometries."""

    def __init__(self, n_channels=3, n_classes=2, outplanes=1024, embed_dim=384, dw_stride=1, bilinear=False, base=64):
        _Packed.__init__(self)
        if bilinear:
            raise NotImplementedError("or_unet.UNet(bilinear=True) is never used by the reference script")
        if n_channels != 3:
            raise ValueError("or_unet.UNet: the first conv is the 3-channel image kernel")
        if dw_stride != 1:
            raise NotImplementedError("or_unet.UNet: the reference script builds it with dw_stride=1 only")
        if base % 8 or embed_dim % 8:
            raise ValueError("or_unet.UNet: base and embed_dim must be multiples of 8 (16-byte vector kernels)")
        self.n_channels, self.n_classes, self.bilinear, self.dw_stride = n_channels, n_classes, bilinear, dw_stride
        self.num_classes = n_classes
        self.embed_dim = embed_dim
        c = [base, 2 * base, 4 * base, 8 * base, 16 * base]
        self.inc = P.DoubleConv(n_channels, c[0])
        self.down1, self.down2, self.down3, self.down4 = (P.Down(c[i], c[i + 1]) for i in range(4))
        self.up1, self.up2, self.up3, self.up4 = (P.Up(c[4 - i], c[3 - i]) for i in range(4))
        self.outc = P.OutConv(c[0], n_classes)
        self.fuser = FusionModel()
        self.expand_block_2 = FCUUp(inplanes=embed_dim, outplanes=c[2], up_stride=dw_stride)
        self.expand_block_3 = FCUUp(inplanes=embed_dim, outplanes=c[1], up_stride=dw_stride)
        self.expand_block_4 = FCUUp(inplanes=embed_dim, outplanes=c[0], up_stride=dw_stride)
        self.sync_bn = False

    GRAD_ORDER = ("outc", "up4", "up3", "up2", "up1", "down4", "down3", "expand_block_2", "down2", "expand_block_3", "down1",
                  "expand_block_4", "inc")

    # ---- FCUUp + FusionModel ---------------------------------------------------------------------------------------
    def _map16(self, x):
        """ViT map fp32 NCHW [B, D, h, w] (`:283-306` rearrange) -> (hi, lo|None) 16-bit NHWC."""
        if x.dim() != 4 or x.shape[1] != self.embed_dim:
            raise ValueError(f"or_unet.UNet: ViT map must be [B, {self.embed_dim}, h, w], got {tuple(x.shape)}")
        return self._to_nhwc16(x)

    def _fcu_fuse_fwd(self, key: str, blk: FCUUp, tok, level, save: bool, training: bool):
        """level (hi, lo|None) [B,H,W,C] <- relu(level + nearest(relu(bn(conv1x1(tok))))) in place; returns the saved state."""
        dt = config.operand_dtype
        th, tl = tok
        B, h, w, D = th.shape
        _, H, W, C = level[0].shape
        split = tl is not None
        P_ = B * h * w
        cp = blk.conv_project
        w_hi = self._w16(key + ".w", cp.weight)
        bias = self._f32(key + ".b", cp.bias)
        z = torch.empty((P_, C), device=th.device, dtype=torch.float32)
        if split:
            w_lo = _pack(self._cache, key + ".wlo", cp.weight,
                         lambda p: ops.cast_pad(p.reshape(p.shape[0], -1).contiguous().float(), dtype=dt, part=1))
            ops.gemm_split(th.view(P_, D), tl.view(P_, D), w_hi, w_lo, out=z, bias_n=bias)
        else:
            ops.gemm(th.view(P_, D), w_hi, out=z, bias_n=bias)
        if training:
            scale, shift, mean, invstd, count = _bn.finalize(ops.colstats(z), P_, blk.bn, self.sync_bn)
        else:
            scale, shift = ops.bn_eval_affine(blk.bn)
            mean = invstd = count = None
        z = z.view(B, h, w, C)
        split_lvl = level[1] is not None
        r = ops.bn_act(z, scale, shift, True, dt, split_lvl)
        r_hi, r_lo = r if split_lvl else (r, None)
        ys, y0 = ops.nearest_tables(h, H * blk.up_stride, th.device)
        xs, x0 = ops.nearest_tables(w, W * blk.up_stride, th.device)
        ops.nearest_add_relu(level[0], level[1], r_hi, r_lo, ys, xs)
        if not (save and training):
            return None
        return dict(tok=th, z=z, scale=scale, shift=shift, mean=mean, invstd=invstd, count=count, y0=y0, x0=x0)

    def _fcu_fuse_bwd(self, key: str, blk: FCUUp, sv, g, inv_scale, grads, prefix: str):
        """g fp32 [B,H,W,C] = gradient of the fused map (the UNet branch takes it unchanged) -> FCUUp parameter gradients."""
        import torch.distributed as dist
        dt = config.operand_dtype
        z = sv["z"]
        B, h, w, C = z.shape
        D = sv["tok"].shape[-1]
        dr = ops.nearest_sum(g, h, w, sv["y0"], sv["x0"])
        gz, partial_ = ops.upsample_bn_relu_bwd(dr, z, sv["scale"], sv["shift"], sv["mean"], sv["invstd"], 1)
        red = ops.reduce_rows(partial_.view(partial_.shape[0], 2 * C))
        local = red
        if self.sync_bn and parallel.bn_collectives_on():
            local = red.clone()
            dist.all_reduce(red)
        dz16, bpart = ops.bn_bwd_apply(gz, z, sv["mean"], sv["invstd"], self._f32(key + ".g", blk.bn.weight), red[C:], red[:C],
                                       sv["count"], dt)
        ops.reduce_rows(local[:C].view(1, C), inv_scale, grads[prefix + ".bn.bias"])
        ops.reduce_rows(local[C:].view(1, C), inv_scale, grads[prefix + ".bn.weight"])
        ops.reduce_rows(bpart, inv_scale, grads[prefix + ".conv_project.bias"])
        gw = ops.wgrad(dz16.view(1, B * h * w, 1, C), sv["tok"].view(1, B * h * w, 1, D), C, 1, 1, 1, 0, inv_scale)
        grads[prefix + ".conv_project.weight"].view(C, D).copy_(gw.view(C, D))

    # ---- first DoubleConv: 3-channel direct conv + the shared stage ---------------------------------------------------
    def _inc_fwd(self, img, save: bool, training: bool):
        from .decoders import conv_bn_relu_up_forward as stage
        dt = config.operand_dtype
        sp = config.split_conv
        seq = self.inc.double_conv
        img = img.detach().float().contiguous()
        B, _, H, W = img.shape
        raw0 = ops.conv3x3_c3(img, self._f32("inc0", seq[0].weight), 1, 1)
        if training:
            scale, shift, mean, invstd, count = _bn.finalize(ops.colstats(raw0), B * H * W, seq[1], self.sync_bn)
        else:
            scale, shift = ops.bn_eval_affine(seq[1])
            mean = invstd = count = None
        a = ops.bn_act(raw0, scale, shift, True, dt, sp)
        a = a if sp else (a, None)
        sv0 = None
        if save and training:
            x8 = ops.cast_pad(img.permute(0, 2, 3, 1).contiguous().view(B * H * W, 3), 8, dt).view(B, H, W, 8)
            sv0 = (x8, raw0, scale, shift, mean, invstd, count)
        x1, sv1 = stage(self, "incb", a[0], a[1], seq[3], seq[4], 1, self.sync_bn, save, training)
        return x1, (sv0, sv1)

    def _inc_bwd(self, sv, g, inv_scale, grads):
        import torch.distributed as dist
        from .decoders import conv_bn_relu_up_backward as stage_bwd
        dt = config.operand_dtype
        seq = self.inc.double_conv
        p = "inc.double_conv"
        d0 = stage_bwd(self, "incb", sv[1], g, seq[3], seq[4], inv_scale, grads, p, True, self.sync_bn, conv_name=p + ".3",
                       bn_name=p + ".4")
        x8, raw0, scale, shift, mean, invstd, count = sv[0]
        C0 = raw0.shape[-1]
        gz, partial_ = ops.upsample_bn_relu_bwd(d0, raw0, scale, shift, mean, invstd, 1)
        red = ops.reduce_rows(partial_.view(partial_.shape[0], 2 * C0))
        local = red
        if self.sync_bn and parallel.bn_collectives_on():
            local = red.clone()
            dist.all_reduce(red)
        dx16, _ = ops.bn_bwd_apply(gz, raw0, mean, invstd, self._f32("inc1.g", seq[1].weight), red[C0:], red[:C0], count, dt)
        ops.reduce_rows(local[:C0].view(1, C0), inv_scale, grads[p + ".1.bias"])
        ops.reduce_rows(local[C0:].view(1, C0), inv_scale, grads[p + ".1.weight"])
        gw = ops.wgrad(dx16, x8, C0, 3, 3, 1, 1, inv_scale)                 # [C0, 8, 3, 3]
        grads[p + ".0.weight"].copy_(gw[:, :3])

    # ---- functional core -------------------------------------------------------------------------------------------
    def _forward_core(self, img, maps, save: bool, training: Optional[bool] = None):
        """img fp32 [B,3,H,W]; maps = [(hi, lo|None) NHWC of x_o, x_t2, x_d2] -> logits fp32 NHWC [B,H,W,classes], saved."""
        training = self.training if training is None else training
        m_o, m_t2, m_d2 = maps
        sv = {}
        x1, sv["inc"] = self._inc_fwd(img, save, training)
        sv["f4"] = self._fcu_fuse_fwd("eb4", self.expand_block_4, m_t2, x1, save, training)
        p, pl, sv["idx1"] = ops.maxpool2_fwd(x1[0], x1[1], save_idx=save)
        x2, sv["down1"] = self._dconv_fwd("d1", self.down1.maxpool_conv[1], (p, pl), save, training)
        sv["f3"] = self._fcu_fuse_fwd("eb3", self.expand_block_3, m_o, x2, save, training)
        p, pl, sv["idx2"] = ops.maxpool2_fwd(x2[0], x2[1], save_idx=save)
        x3, sv["down2"] = self._dconv_fwd("d2", self.down2.maxpool_conv[1], (p, pl), save, training)
        sv["f2"] = self._fcu_fuse_fwd("eb2", self.expand_block_2, m_d2, x3, save, training)
        p, pl, sv["idx3"] = ops.maxpool2_fwd(x3[0], x3[1], save_idx=save)
        x4, sv["down3"] = self._dconv_fwd("d3", self.down3.maxpool_conv[1], (p, pl), save, training)
        p, pl, sv["idx4"] = ops.maxpool2_fwd(x4[0], x4[1], save_idx=save)
        x5, sv["down4"] = self._dconv_fwd("d4", self.down4.maxpool_conv[1], (p, pl), save, training)
        y = x5
        for i, skip in enumerate((x4, x3, x2, x1)):
            up = getattr(self, f"up{i + 1}")
            y, sv[f"up{i + 1}.up"] = self._up_fwd(f"u{i + 1}", up.up, y, skip, save)
            y, sv[f"up{i + 1}"] = self._dconv_fwd(f"u{i + 1}c", up.conv, y, save, training)
        sv["skip_shapes"] = [tuple(t[0].shape) for t in (x4, x3, x2, x1)]
        logits = self._outc_fwd(y)
        sv["x_last"] = y[0] if save else None
        return logits, sv

    def _backward_core(self, saved, d16, bias_partial, inv_scale, grads, dlogits_f32=None, stage_done=None, d_lo=None):
        def done():
            if stage_done is not None:
                stage_done()

        dU = self._outc_bwd(saved["x_last"], d16, bias_partial, inv_scale, grads, dlogits_f32, d_lo)
        done()
        skips = []
        for i in (4, 3, 2, 1):                       # up4 (skip x1) ... up1 (skip x4)
            up = getattr(self, f"up{i}")
            dcat = self._dconv_bwd(f"u{i}c", up.conv, saved[f"up{i}"], dU, inv_scale, grads, f"up{i}.conv", True)
            dU = self._up_bwd(f"u{i}", up.up, saved[f"up{i}.up"], dcat, inv_scale, grads, f"up{i}.up")
            Bs, Hs, Ws, Cs = saved["skip_shapes"][i - 1]
            g = torch.empty((Bs, Hs, Ws, Cs), device=d16.device, dtype=torch.float32)
            ops.copy_channels(dcat.view(-1, dcat.shape[-1])[:, :Cs], g.view(-1, Cs))
            skips.append(g)
            done()
        g1, g2, g3, g4 = skips                        # gradients of x1 .. x4 from their skip connections
        dP = self._dconv_bwd("d4", self.down4.maxpool_conv[1], saved["down4"], dU, inv_scale, grads, "down4.maxpool_conv.1", True)
        ops.maxpool2_bwd(dP, saved["idx4"], g4)
        done()
        dP = self._dconv_bwd("d3", self.down3.maxpool_conv[1], saved["down3"], g4, inv_scale, grads, "down3.maxpool_conv.1", True)
        ops.maxpool2_bwd(dP, saved["idx3"], g3)
        done()
        self._fcu_fuse_bwd("eb2", self.expand_block_2, saved["f2"], g3, inv_scale, grads, "expand_block_2")
        done()
        dP = self._dconv_bwd("d2", self.down2.maxpool_conv[1], saved["down2"], g3, inv_scale, grads, "down2.maxpool_conv.1", True)
        ops.maxpool2_bwd(dP, saved["idx2"], g2)
        done()
        self._fcu_fuse_bwd("eb3", self.expand_block_3, saved["f3"], g2, inv_scale, grads, "expand_block_3")
        done()
        dP = self._dconv_bwd("d1", self.down1.maxpool_conv[1], saved["down1"], g2, inv_scale, grads, "down1.maxpool_conv.1", True)
        ops.maxpool2_bwd(dP, saved["idx1"], g1)
        done()
        self._fcu_fuse_bwd("eb4", self.expand_block_4, saved["f4"], g1, inv_scale, grads, "expand_block_4")
        done()
        self._inc_bwd(saved["inc"], g1, inv_scale, grads)
        done()
        return None

    # ---- reference-shaped entry point ----------------------------------------------------------------------------------
    def forward(self, x, x_o, x_t2, x_d2):
        """`eval_dinov2_or_unet_fuse.py:448-483`: image (B,3,H,W) + the three ViT maps (B,D,h,w) -> logits (B,classes,H,W)."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _FuseFn.apply(self, x, x_o, x_t2, x_d2, *list(self.parameters()))
        logits, _ = self._forward_core(x, [self._map16(t) for t in (x_o, x_t2, x_d2)], save=False)
        return logits.permute(0, 3, 1, 2)


class ORUNetFuseEngine(nn.Module):
    """The training step of `eval/eval_dinov2_or_unet_fuse.py:266-331` with a frozen ViT:

        inp_t2, inp_d2 = bilinear resize of the image by 1.5 / 0.5 (`:279-280`)
        x_o, x_t2, x_d2 = last-layer patch tokens of model(inp), model(inp_t2), model(inp_d2) as maps (`:281-306`, no_grad)
        logits = seg_decoder(inp, x_o, x_t2, x_d2) ; resize to the label size (an identity here) ; loss = CE + DC(2)
        backward through the head ; SGD on the head (`:211-218`)

    The token matrices of the three ViT passes are the NHWC maps the head's FCUUp GEMMs read — no NCHW round trip.
    Gradients live in one flat bucket in gradient-ready order, all-reduced per stage on the side stream (data parallel).
    """

    def __init__(self, model, seg_decoder: UNet, *, lr: float = 0.01, momentum: float = 0.9, weight_decay: float = 0.0,
                 process_group=None):
        super().__init__()
        from ..optim import SGD, FlatBucket
        from ..parallel import StageReducer
        self.model, self.seg_decoder, self.process_group = model, seg_decoder, process_group
        self.patch = model.patch_size
        for p in model.parameters():
            p.requires_grad_(False)
        order = list(seg_decoder.GRAD_ORDER)
        named = dict(seg_decoder.named_parameters())
        ordered = [(n, named[n]) for pre in order for n in named if n.startswith(pre + ".")]
        assert len(ordered) == len(named)
        self.bucket = FlatBucket(ordered)
        self.stage_ranges = [self.bucket.range_of([n for n in named if n.startswith(pre + ".")]) for pre in order]
        self.optimizer = SGD([self.bucket], lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.reducer = StageReducer(self.bucket.grad, self.stage_ranges, process_group)

    @staticmethod
    def _rescale(inp: torch.Tensor, s: float) -> torch.Tensor:
        """``F.interpolate(inp, scale_factor=(s, s), mode="bilinear", align_corners=False)`` (`:281,291`) on the HIP resize kernel:
        each NCHW plane is an NHWC image of one channel.  With an integral output size (588 * 1.5 = 882, 588 * 0.5 = 294) the
        scale-factor form and the kernel's size form sample the same coordinates; other sizes keep ATen's scale-factor form."""
        B, C, H, W = inp.shape
        Ho, Wo = H * s, W * s
        if Ho != int(Ho) or Wo != int(Wo) or not inp.is_cuda:
            import torch.nn.functional as F
            return F.interpolate(inp, scale_factor=(s, s), mode="bilinear", align_corners=False)
        x = inp.float().contiguous().view(B * C, H, W, 1)
        return ops.resize_bilinear_fwd(x, int(Ho), int(Wo)).view(B, C, int(Ho), int(Wo))

    @torch.no_grad()
    def vit_maps(self, inp: torch.Tensor):
        """-> [(hi, lo|None) NHWC 16-bit maps of x_o, x_t2, x_d2] (`:279-306`)"""
        dt = config.operand_dtype
        maps = []
        for s in (1.0, 1.5, 0.5):
            x = inp if s == 1.0 else self._rescale(inp, s)
            tok = self.model.get_intermediate_layers(x, 1)[0]                 # (B, h*w, D) normalised patch tokens
            B, N, D = tok.shape
            h, w = x.shape[2] // self.patch, x.shape[3] // self.patch
            t2 = tok.reshape(B * N, D).float().contiguous()
            hi = ops.cast_pad(t2, D, dt).view(B, h, w, D)
            lo = ops.cast_pad(t2, D, dt, part=1).view(B, h, w, D) if config.split_conv else None
            maps.append((hi, lo))
        return maps

    @torch.no_grad()
    def train_step(self, inp: torch.Tensor, target: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        from ..parallel import world_size
        dec = self.seg_decoder
        S = config.loss_scale
        dt = config.operand_dtype
        maps = self.vit_maps(inp)
        logits, saved = dec._forward_core(inp, maps, save=True, training=True)
        target = target.long().contiguous()
        loss, coef, _ = ops.seg_loss_fwd(logits, target, 1, ops.LOSS_DICE, 10e-20, 1, None, S)    # CE + DC(2), `:311-316`
        dz = ops.seg_loss_bwd(logits, target, coef, 1, ops.LOSS_DICE, 1, None)
        _, hh, ww, _ = logits.shape
        r = ops.resize_bilinear_bwd(dz, hh, ww, dt, config.split_conv)
        d16, d_lo, bpart = r if config.split_conv else (r[0], None, r[1])
        inv = 1.0 / (S * world_size(self.process_group))
        self.reducer.begin()
        dec._backward_core(saved, d16, bpart, inv, self.bucket.views, stage_done=self.reducer.stage_done, d_lo=d_lo)
        self.reducer.finish()
        self.optimizer.step(1.0)
        if taps is not None:
            taps.update(logits=logits, loss=loss, maps=maps)
        return loss.view(())
