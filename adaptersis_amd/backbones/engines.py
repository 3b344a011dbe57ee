"""Engine entry points of the MI355X build (the reference's `backbones/engines.py` only holds the
unused ``pre_vit`` patch-embed clone, which is kept here for API parity).

``SegEngine`` is the fused form of the step body that the reference writes inline in
`train.py:268-436` (and copies into `validate_network`, `train.py:465-612`):

    encoder -> ViT pass A (cls + pos-embed, last-4 normed features) -> ViT pass B (raw patch tokens,
    blocks[0:-3]) -> 4 x [block, CAViT, CACNN, + pass-A feature] -> decoder input assembly ->
    FeatureDecoder -> resize + softmax + DC (softmax again) -> backward of the decoder (the only
    part of the graph that receives gradients: SURVEY.md fact 1) -> gradient all-reduce (RCCL,
    launched per decoder stage on a side stream as soon as that stage's gradients exist) -> SGD.

Everything is a HIP kernel from libasis_hip.so; torch provides device memory, streams and
``torch.distributed``.  No autograd graph is built: the engine calls the same functional cores
(`FeatureDecoder._forward_core/_backward_core`) that the reference-shaped modules wrap in
``torch.autograd.Function``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn

from .. import config, ops
from ..dinov2.layers.blocks import _Packed, _pack, run_blocks
from ..optim import SGD, FlatBucket
from ..parallel import StageReducer, world_size
from .adapter_blocks import CACNN, CAViT, deform_inputs


class pre_vit(_Packed):
    """`backbones/engines.py:4-60`: (B, C, H, W) -> (B, N, D) patch embedding, Conv2d(k = s = patch).
    Unused by every reference script; implemented as an implicit-GEMM conv for completeness."""

    def __init__(self, img_size=84, patch_size=14, in_chans=256, embed_dim=384, norm_layer=None, flatten_embedding=True):
        super().__init__()
        if norm_layer is not None:
            raise ValueError("pre_vit: norm_layer is never set in the reference")
        self.img_size, self.patch_size = img_size, patch_size
        self.patches_resolution = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim, self.flatten_embedding = in_chans, embed_dim, flatten_embedding
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        P = self.patch_size
        assert H % P == 0, f"Input image height {H} is not a multiple of patch height {P}"
        assert W % P == 0, f"Input image width {W} is not a multiple of patch width: {P}"
        dt = config.operand_dtype
        x16 = ops.cast_pad(x.permute(0, 2, 3, 1).contiguous().float().view(B * H * W, C), C, dt).view(B, H, W, C)
        w16 = _pack(self._cache, "w", self.proj.weight, lambda p: ops.pack_conv_weight(p.float().contiguous(), 0, dt))
        out = ops.conv_gemm(x16, w16, P, P, P, 0, bias_n=self._f32("b", self.proj.bias))
        out = out.view(B, (H // P) * (W // P), self.embed_dim)
        if not self.flatten_embedding:
            out = out.reshape(-1, H // P, W // P, self.embed_dim)
        return out


class SegEngine(nn.Module):
    """One object owning the frozen ViT, the CNN encoder, the adapters, the decode head and the optimizer.

    mode="reference_exact" reproduces `train.py` as written: adapters/encoder run forward only (graph
    cut at `train.py:389-406`), ``level_embed`` is a fresh zero tensor (no-op), the loss applies
    softmax twice, only the decoder is optimised (momentum 0.99, wd 3e-5: `train.py:178-191`).
    """

    # loss name -> (n_region, region mode, eps / smooth, n_ce) of asis_seg_loss_fwd, all on softmax(resize(logits))
    # as the scripts apply `nn.Softmax(1)` before the loss (`train.py:424`) — except "ce_dc", which the decoder-only
    # scripts apply to the raw resized logits (`eval/eval_dinov2_unet.py:291-297`)
    LOSSES = {
        "dice": (2, ops.LOSS_DICE, 10e-20, 0),          # train.py:427-428  DC(2)(softmax(out))
        "iou": (2, ops.LOSS_IOU, 1e-6, 0),              # train_multi_class.py:390-393
        "softdice": (1, ops.LOSS_SOFTDICE, 1.0, 0),     # train.py:425 (commented alternative)
        "dc_and_ce": (1, ops.LOSS_SOFTDICE, 1.0, 2),    # train.py:426 (commented alternative)
        "tversky": (1, ops.LOSS_TVERSKY, 1.0, 0),       # train.py:50 import
        "ce_dc": (1, ops.LOSS_DICE, 10e-20, 1),         # eval/eval_dinov2_unet.py:291-297, eval_dinov2_setr_cross_ete.py:334-340
    }

    def __init__(self, model, backbone_encoder, cross_vit: CAViT, cross_cnn: CACNN, seg_decoder, *,
                 n_last_blocks: int = 4, num_classes: int = 2, lr: float = 0.01, momentum: float = 0.99,
                 weight_decay: float = 3e-5, mode: str = "reference_exact", process_group=None, loss: str = "dice",
                 train_encoder: bool = False, train_backbone: bool = False, optimize_backbone: bool = False,
                 blocks_per_bucket: int = 4, grad_compress: Optional[str] = None):
        """``seg_decoder``: ``FeatureDecoder`` -> the `train.py` flow; ``DecoderMLA`` -> the `train_mla.py` flow
        (block -> CACNN -> CAViT order, the four adapter-stream maps feed the MLA head, `blocks[-2]` is evaluated
        twice and `blocks[-1]` never: `train_mla.py:318,340`).  ``loss``: a key of ``SegEngine.LOSSES``.

        ``train_backbone`` (with mode="train_adapters"; BASELINE config 4): the `train.py` adapter flow with its
        no_grad / inference_mode regions removed (`train.py:287,300-302,389-406`), i.e. the unfreezing pattern of
        `eval/eval_dinov2_setr_cross_ete.py:145-148,307-361` applied to this flow: BOTH ViT passes run with saved
        activations, the backward goes through the head, the four adapter stages, the encoder (``train_encoder``) and
        every block evaluation of both passes with weight gradients; the 304 M backbone gradients live in a flat
        gradient bucket all-reduced in ``blocks_per_bucket``-block chunks while earlier blocks are still in their
        backward.  Like the reference script (its optimizer lists the decoder only, `:224-229`) the backbone gradients are
        computed and exchanged but not applied unless ``optimize_backbone`` is set.  ``grad_compress`` ("bf16" | "none"; default: the ``ASIS_GRAD_COMPRESS``
        environment variable, else "bf16"): the backbone bucket travels as bfloat16 (parallel.StageReducer) — half the bytes of the
        one exchange that is large enough for an xGMI ring to notice (0.6 instead of 1.2 GB); the decoder / adapter / encoder buckets
        (63 + 31 MB) always travel as fp32 like the reference's DDP."""
        super().__init__()
        if grad_compress is None:   # default since round 5: bf16 transport for the 1.2 GB backbone bucket (ASIS_GRAD_COMPRESS=none: fp32)
            grad_compress = os.environ.get("ASIS_GRAD_COMPRESS", "bf16").lower()
        if grad_compress in ("", "0", "none", "off", "fp32"):
            grad_compress = None
        if mode not in ("reference_exact", "train_adapters"):
            raise ValueError("mode must be 'reference_exact' or 'train_adapters'")
        if loss not in self.LOSSES:
            raise ValueError(f"loss must be one of {sorted(self.LOSSES)}")
        self.loss_kind = loss
        self.is_mla = type(seg_decoder).__name__ == "DecoderMLA"
        # UNet head (BASELINE config 2): fed with the adapter-stream map alone, (B, h, w, D)
        self.stream_only = type(seg_decoder).__name__ == "UNet"
        self.model, self.backbone_encoder = model, backbone_encoder
        self.cross_vit, self.cross_cnn, self.seg_decoder = cross_vit, cross_cnn, seg_decoder
        self.n_last_blocks, self.num_classes, self.mode = n_last_blocks, num_classes, mode
        self.patch = model.patch_size
        self.heads = model.num_heads
        self.process_group = process_group
        if train_encoder and mode != "train_adapters":
            raise ValueError("train_encoder needs mode='train_adapters'")
        self.train_encoder = train_encoder
        frozen = list(model.parameters()) + ([] if train_encoder else list(backbone_encoder.parameters()))
        if mode == "reference_exact":
            frozen += list(cross_vit.parameters()) + list(cross_cnn.parameters())
        for p in frozen:
            p.requires_grad_(False)  # no gradient reaches them in the reference step
        # gradient-ready order of the decoder backward: final conv first, decoder_1 last
        order = list(seg_decoder.GRAD_ORDER)
        named = dict(seg_decoder.named_parameters())
        ordered = [(n, named[n]) for pre in order for n in named if n.startswith(pre + ".")]
        assert len(ordered) == len(named)
        self.bucket = FlatBucket(ordered)
        self.stage_ranges = [self.bucket.range_of([n for n in named if n.startswith(pre + ".")]) for pre in order]
        buckets = [self.bucket]
        self.adapter_bucket = None
        if mode == "train_adapters":
            # the trainable set the reference's optimiser lists for the adapters (`train.py:178-186`) and that its
            # no_grad block (`:389-406`) and forward-only MSDeformAttnFunction keep from ever training (SURVEY facts
            # 1-2): CAViT + CACNN parameters, one flat bucket, all-reduced once the adapter backward is enqueued
            if type(seg_decoder).__name__ not in ("FeatureDecoder", "DecoderSETR", "UNet", "DecoderMLA"):
                raise NotImplementedError("train_adapters is built for the train.py adapter flow (FeatureDecoder / UNet heads) and "
                                          "the train_mla.py flow (DecoderMLA)")
            if self.is_mla and train_backbone:
                raise NotImplementedError("the train_mla.py flow trains decoder + adapters (+ encoder); the unfrozen backbone "
                                          "(BASELINE config 4) is the train.py adapter flow")
            for p in list(cross_vit.parameters()) + list(cross_cnn.parameters()):
                p.requires_grad_(True)
            named_a = [("cross_vit." + n, p) for n, p in cross_vit.named_parameters()] + \
                      [("cross_cnn." + n, p) for n, p in cross_cnn.named_parameters()]
            self.adapter_bucket = FlatBucket(named_a)
            self.adapter_reducer = StageReducer(self.adapter_bucket.grad, [(0, self.adapter_bucket.numel)], process_group)
            buckets.append(self.adapter_bucket)
            self.encoder_bucket = None
            if train_encoder:  # the last member of the reference's optimiser list (`train.py:183`): the spatial-prior CNN
                for p in backbone_encoder.parameters():
                    p.requires_grad_(True)
                self.encoder_bucket = FlatBucket([("backbone_encoder." + n, p) for n, p in backbone_encoder.named_parameters()])
                self.encoder_reducer = StageReducer(self.encoder_bucket.grad, [(0, self.encoder_bucket.numel)], process_group)
                buckets.append(self.encoder_bucket)
        self.train_backbone = bool(train_backbone)
        # precision policy of the attention output (config.split_attn_out): hi + lo halves into the projection GEMM for the
        # heads that amplify a stream error most (UNet 3.9x, MLA 3.2x; FeatureDecoder 2x) and for the unfrozen backbone
        self.split_attn_out = config.split_attn_out_policy if config.split_attn_out_policy is not None else \
            bool(self.stream_only or self.is_mla or train_backbone)
        # precision level of the ViT blocks (config.precise_level): level 2 (every linear layer on hi + lo operands) for the one
        # geometry whose full-depth stress golden needs it — the MLA head (amplifies a stream error 3.2x) behind 2 x 40 block
        # evaluations (BASELINE config 5: 1.30e-3 on single 16-bit operands, 3.0e-4 on level 2; tests/test_gpu_fulldepth.py)
        self.precise_level = config.precise_level_policy if config.precise_level_policy is not None else \
            (2 if (self.is_mla and len(model.blocks) >= 40) else 0)
        # ... and WHICH linear layers of a block run split there (config.precise_parts): the automatic level 2 of config 5 needs
        # the attention output (proj) and the LayerNorm output in front of w12 only — 8.3e-4 on the full-depth stress golden at
        # 58.9 img/s against 3.1e-4 at 48.2 with all four layers split (A/B table in config.py); a level forced by
        # ASIS_PRECISE_LEVEL keeps all four unless ASIS_PRECISE_PARTS names a subset
        all_parts = frozenset(("qkv", "proj", "fc1", "fc2"))
        self.precise_parts = config.precise_parts_policy if config.precise_parts_policy is not None else \
            (frozenset(("proj", "fc1")) if (config.precise_level_policy is None and self.precise_level == 2) else all_parts)
        self.vit_bucket = None
        if train_backbone:
            if mode != "train_adapters":
                raise ValueError("train_backbone needs mode='train_adapters' (the unfrozen variant of the adapter flow)")
            self.vit_bucket, self.vit_reducer, self._fire_at = make_vit_bucket(model, blocks_per_bucket, process_group,
                                                                               momentum=optimize_backbone,
                                                                               min_first_blocks=n_last_blocks,
                                                                               compress=grad_compress)
            if optimize_backbone:
                buckets.append(self.vit_bucket)
        self.optimizer = SGD(buckets, lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.reducer = StageReducer(self.bucket.grad, self.stage_ranges, process_group)
        self._geom = {}

    # ------------------------------------------------------------------------------------------
    def _geometry(self, H, W, shapes, dev):
        key = (H, W, tuple(shapes), dev)
        g = self._geom.get(key)
        if g is None:
            d1, d2 = deform_inputs(torch.zeros(1, 3, H, W), self.patch, shapes)
            g = {"ref1": d1[0][0, :, 0, :].contiguous().to(dev), "shapes1": d1[1].to(torch.int32).to(dev),
                 "starts1": d1[2].to(torch.int32).to(dev), "ref2": d2[0][0, :, 0, :].contiguous().to(dev),
                 "shapes2": d2[1].to(torch.int32).to(dev), "starts2": d2[2].to(torch.int32).to(dev)}
            self._geom = {key: g}
        return g

    def _encoder_on_side_stream(self, inp: torch.Tensor):
        """The spatial-prior CNN (≈ 30 small, latency-bound launches + 6 SyncBatchNorm all-reduces of 2C doubles) is
        independent of the ViT until the first adapter stage: run it on a side HIP stream so it fills the gaps of the
        GEMM-bound block loop and — on N > 1 ranks — its six blocking statistic exchanges wait for each other there, not
        on the compute stream.  -> (c tokens, level shapes, event recorded behind the last encoder kernel)."""
        main = torch.cuda.current_stream()
        if getattr(self, "_enc_stream", None) is None:
            self._enc_stream = torch.cuda.Stream()
        side = self._enc_stream
        side.wait_stream(main)          # the input batch, and everything of the previous step that read recycled blocks
        with torch.cuda.stream(side):
            _, c, shapes = self.backbone_encoder.forward_tokens(inp, need_c1=not config.elide_c1)
            done = torch.cuda.Event()
            done.record(side)
        c.record_stream(main)           # allocated on the side stream, consumed on the compute stream
        return c, shapes, done

    def _trunk_dual(self, xcat: torch.Tensor, Ra: int, blocks, segs) -> torch.Tensor:
        """The trunk blocks with the two token batches on two HIP streams (config.dual_stream; ASIS_TRUNK_STREAMS=4 cuts each
        batch in two image groups: four streams) -> the stacked [Ra + Rb, D]."""
        main = torch.cuda.current_stream()
        nper = 2 if config.trunk_streams >= 4 and min(b for b, _ in segs) >= 2 else 1
        if getattr(self, "_dual_streams", None) is None or len(self._dual_streams) != 2 * nper:
            self._dual_streams = tuple(torch.cuda.Stream() for _ in range(2 * nper))
        parts = []                                   # (row slice, [(B, N)]) per stream, in row order
        r0 = 0
        for (B, N) in segs:
            cuts = [0, B] if nper == 1 else [0, B // 2, B]
            for a, b in zip(cuts[:-1], cuts[1:]):
                parts.append((slice(r0 + a * N, r0 + b * N), [(b - a, N)]))
            r0 += B * N
        xs = []
        for st, (sl, _) in zip(self._dual_streams, parts):
            st.wait_stream(main)
            xcat.record_stream(st)
            xs.append(xcat[sl])
        # block by block, alternating between the streams (the launch order the streams' kernels interleave in); inside the run
        # the residual stream stays in the two-plane form where Block.fold_ok allows (blocks.run_blocks semantics)
        nblk = len(blocks)
        for bi, blk in enumerate(blocks):
            for i, (st, (_, sg)) in enumerate(zip(self._dual_streams, parts)):
                with torch.cuda.stream(st):
                    R_i = xs[i][0].shape[0] if isinstance(xs[i], tuple) else xs[i].shape[0]
                    if blk.fold_ok(R_i, sg):
                        xs[i] = blk.forward_rows_fold(xs[i], sg, planes_out=bi + 1 < nblk and blocks[bi + 1].fold_ok(R_i, sg))
                    else:
                        xs[i] = blk.forward_rows(xs[i], sg)
        for st, x in zip(self._dual_streams, xs):
            main.wait_stream(st)
            x.record_stream(main)
        return torch.cat(xs, 0)

    def _cavit(self, x2, c2, g, B, Lq, Lin):
        cv = self.cross_vit
        return cv.attn.forward16(cv._ln16("query_norm", x2), cv._ln16("feat_norm", c2), g["ref1"], g["shapes1"],
                                 g["starts1"], B, Lq, Lin, res=x2, scale_n=cv._f32("gamma", cv.gamma))

    def _cacnn(self, c2, x2, g, B, Lq, Lin, grids):
        cn = self.cross_cnn
        out = cn.attn.forward16(cn._ln16("query_norm", c2), cn._ln16("feat_norm", x2), g["ref2"], g["shapes2"],
                                g["starts2"], B, Lq, Lin, res=c2)
        return cn.ffn.forward16(cn._ln16("ffn_norm", out), out, B, Lq, grids) if cn.with_cffn else out

    @torch.no_grad()
    def features(self, inp: torch.Tensor, taps: Optional[dict] = None, adapter_saves: Optional[list] = None):
        """`train.py:275-406`: image batch -> decoder input, NHWC 16-bit [B, h, w, 3D] as (hi, lo|None).
        ``adapter_saves`` (train_adapters mode): list that receives, per stage, the saved activations of the frozen
        block on pass B and of CAViT / CACNN."""
        config.split_attn_out = self.split_attn_out
        config.precise_level = self.precise_level
        config.precise_parts = self.precise_parts
        m = self.model
        B, _, H, W = inp.shape
        inp = inp.float().contiguous()
        D = m.embed_dim
        h, w = H // self.patch, W // self.patch
        N = h * w
        nb = len(m.blocks)
        nl = self.n_last_blocks
        enc_done = None
        if self.train_encoder and adapter_saves is not None:
            c, shapes, esaved = self.backbone_encoder.forward_tokens_train(inp)
            self._esaved = (esaved, shapes)
        elif config.encoder_stream and inp.is_cuda:
            c, shapes, enc_done = self._encoder_on_side_stream(inp)
        else:
            _, c, shapes = self.backbone_encoder.forward_tokens(inp, need_c1=not config.elide_c1)
        c_orig = c
        Lc = c.shape[1]
        g = self._geometry(H, W, shapes, inp.device)
        # ---- pass A (train.py:287): cls + pos-embed, all blocks, final norm on the last n outputs ----
        # ---- both passes stacked along the rows: every block evaluation of pass A (cls + pos-embed tokens, all
        # blocks, `train.py:287`) has a pass-B partner on the same frozen weights (raw patch tokens through
        # blocks[0:-3], `train.py:300-302`, then one more block per adapter stage), so each launch carries 2x the rows.
        # The patch embedding (shared by both passes: same conv on the same input) writes pass B's rows of the stacked
        # buffer directly, the cls / pos-embed form goes into pass A's rows: no concatenation copy.
        Ra, Rb = B * (N + 1), B * N
        segs = [(B, N + 1), (B, N)]
        xcat = torch.empty((Ra + Rb, D), device=inp.device, dtype=torch.float32)
        tokens, _ = m.patch_embed.tokens(inp, out=xcat[Ra:])
        if not config.share_patch_embed:   # the reference's second evaluation (pass A's own, `vision_transformer.py:190`): same values
            tokens, _ = m.patch_embed.tokens(inp)
        pos = m._pos_for(N, H, W)
        ops.add_cls_pos(tokens, m.cls_token.detach().reshape(-1).float().contiguous(),
                        pos.detach().reshape(-1, D).float().contiguous(), out=xcat[:Ra].view(B, N + 1, D))
        feats = []
        trunk = list(m.blocks[: nb - (nl - 1)])
        if config.dual_stream and inp.is_cuda and getattr(self, "_dual_warm", False) and nb - nl == len(trunk) - 1:
            xcat = self._trunk_dual(xcat, Ra, trunk, segs)
            feats.append(m._final_norm(xcat[:Ra].view(B, N + 1, D))[:, 1:])
            trunk = []
        self._dual_warm = True      # the first step runs in order: it fills the per-module operand caches on one stream
        if trunk:   # only the LAST trunk block's output is read as fp32 (final norm + adapter stage 0): one run
            assert nb - nl == len(trunk) - 1
            xcat = run_blocks(trunk, xcat, segs)
            feats.append(m._final_norm(xcat[:Ra].view(B, N + 1, D))[:, 1:])   # [B, N, D] view, batch stride (N+1)*D
        if enc_done is not None:
            torch.cuda.current_stream().wait_event(enc_done)   # the pyramid tokens are first needed by adapter stage 0
        if taps is not None:
            taps.update(c=c, x_b0=xcat[Ra:].view(B, N, D).clone(), shapes=shapes)
        c2d = c.view(B * Lc, D)
        train = self.mode == "train_adapters" and adapter_saves is not None
        for s in range(nl):
            bsaved = None
            if s > 0:
                blk = m.blocks[nb - (nl - 1) + s - 1]
                if train:  # pass B through the frozen block with saved activations (input gradient needed), pass A plain
                    xa_s = blk.forward_rows(xcat[:Ra], [(B, N + 1)])
                    xb_s, bsaved = blk.forward_train(xcat[Ra:].view(B, N, D))
                    xcat = torch.cat([xa_s, xb_s.reshape(Rb, D)], 0)
                else:
                    xcat = run_blocks([blk], xcat, segs)
                feats.append(m._final_norm(xcat[:Ra].view(B, N + 1, D))[:, 1:])
            # the last stage's CACNN output (`train.py:372-386`) feeds nothing: the decoder input takes c4 from the
            # ENCODER output (`:395`), so unless a caller asks for the taps it is dead code and skipped (same results)
            dead = s == nl - 1 and taps is None and config.elide_dead_cacnn
            if train:
                # CAViT keeps its query for the LayerNorm backward; the stage output below overwrites these rows of the
                # stacked buffer in place, so the saved copy must be its own tensor
                x2, s_cv = self.cross_vit.forward16_train(xcat[Ra:].clone(), c2d, g, B, N, Lc)
                s_cn = None
                if not dead:
                    c2d, s_cn = self.cross_cnn.forward16_train(c2d, x2, g, B, Lc, N, shapes)
                adapter_saves.append((bsaved, s_cv, s_cn))
            else:
                x2 = self._cavit(xcat[Ra:], c2d, g, B, N, Lc)
                if not dead:
                    c2d = self._cacnn(c2d, x2, g, B, Lc, N, shapes)
            # the stage output overwrites pass B's rows of the stacked buffer: the next block reads it in place
            x = ops.add_f32(x2.view(B, N, D), feats[s], out=xcat[Ra:].view(B, N, D))
        if taps is not None:
            taps.update(feats=feats)
        if self.stream_only:
            xs = x.reshape(B * N, D)
            hi = ops.cast_pad(xs, D, config.operand_dtype).view(B, h, w, D)
            lo = ops.cast_pad(xs, D, config.operand_dtype, part=1).view(B, h, w, D) if config.split_conv else None
            if taps is not None:
                taps.update(x_final=x, c_final=c2d.view(B, Lc, D), cat=hi)
            return hi, lo
        n4 = shapes[2][0] * shapes[2][1]
        d1 = getattr(self.seg_decoder, "decoder_1", None)     # FeatureDecoder: its first conv takes MX lo operands (config.mx_conv)
        mx = bool(config.mx_conv_on() and config.split_conv and d1 is not None and "d1" not in config.unsplit_layers and
                  ops.mx_conv_ok(B * h * w, 3 * D, d1[0].out_channels))
        cat = ops.decoder_input(x, c_orig[:, Lc - n4:], feats[-1], (h, w), shapes[2], config.operand_dtype,
                                config.split_conv, mx=mx)
        if not config.split_conv:
            cat = (cat, None)
        if taps is not None:
            taps.update(x_final=x, c_final=c2d.view(B, Lc, D), cat=cat[0])
        return cat

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def features_mla(self, inp: torch.Tensor, taps: Optional[dict] = None, adapter_saves: Optional[list] = None):
        """`train_mla.py:266-383`: -> the four MLA inputs [(hi, lo|None)] in decoder argument order
        (output_last, output_last_2, output_last_3, output_last_4), each NHWC 16-bit [B, h, w, D].
        ``adapter_saves`` (train_adapters): receives, per adapter use, (saved activations of the frozen block in front of it on
        pass B | None, CAViT save, CACNN save | None) — stage 0 is CAViT alone, stages 1..3 are block -> CACNN -> CAViT."""
        config.split_attn_out = self.split_attn_out
        config.precise_level = self.precise_level
        config.precise_parts = self.precise_parts
        m = self.model
        B, _, H, W = inp.shape
        inp = inp.float().contiguous()
        D = m.embed_dim
        h, w = H // self.patch, W // self.patch
        N = h * w
        nb = len(m.blocks)
        train = self.mode == "train_adapters" and adapter_saves is not None
        if train and self.train_encoder:
            c, shapes, esaved = self.backbone_encoder.forward_tokens_train(inp)
            self._esaved = (esaved, shapes)
        else:
            _, c, shapes = self.backbone_encoder.forward_tokens(inp, need_c1=not config.elide_c1)
        Lc = c.shape[1]
        g = self._geometry(H, W, shapes, inp.device)
        tokens = m.patch_embed(inp)
        if not config.share_patch_embed:   # pass A's own evaluation in the reference (same values)
            m.patch_embed(inp)
        # pass A (`train_mla.py:361-366`, only its last layer is used) rides along with pass B on the same weights:
        # both token batches stacked along the rows for blocks[0:-1] (see ``features``)
        xa = ops.add_cls_pos(tokens, m.cls_token.detach().reshape(-1).float().contiguous(),
                             m._pos_for(N, H, W).detach().reshape(-1, D).float().contiguous())
        Ra = B * (N + 1)
        segs = [(B, N + 1), (B, N)]
        xcat = torch.cat([xa.view(Ra, D), tokens.reshape(B * N, D)], 0)
        xcat = run_blocks(list(m.blocks[: nb - 3]), xcat, segs)
        c2d = c.view(B * Lc, D)
        if train:
            x2, s_cv = self.cross_vit.forward16_train(xcat[Ra:].clone(), c2d, g, B, N, Lc)
            adapter_saves.append((None, s_cv, None))
        else:
            x2 = self._cavit(xcat[Ra:], c2d, g, B, N, Lc)
        outs = [x2]
        for j, bi in enumerate((nb - 3, nb - 2, nb - 2)):  # the reference's repeated [-2:-1]
            bsaved = None
            if train:   # pass B through the frozen block with saved activations (its input gradient is needed), pass A plain
                if j < 2:
                    xa_s = m.blocks[bi].forward_rows(xcat[:Ra], [(B, N + 1)])
                    xb_s, bsaved = m.blocks[bi].forward_train(x2.view(B, N, D))
                    xcat = torch.cat([xa_s, xb_s.reshape(B * N, D)], 0)
                    x2 = xcat[Ra:]
                else:
                    xb_s, bsaved = m.blocks[bi].forward_train(x2.view(B, N, D))
                    x2 = xb_s.reshape(B * N, D)
            elif j < 2:
                xcat[Ra:].copy_(x2)
                xcat = m.blocks[bi].forward_rows(xcat, segs)
                x2 = xcat[Ra:]
            else:
                x2 = m.blocks[bi](x2.view(B, N, D)).view(B * N, D)
            if train:
                c2d, s_cn = self.cross_cnn.forward16_train(c2d, x2.contiguous(), g, B, Lc, N, shapes)
                x2, s_cv = self.cross_vit.forward16_train(x2.clone(), c2d, g, B, N, Lc)
                adapter_saves.append((bsaved, s_cv, s_cn))
            else:
                c2d = self._cacnn(c2d, x2, g, B, Lc, N, shapes)
                x2 = self._cavit(x2, c2d, g, B, N, Lc)
            outs.append(x2)
        xa = m.blocks[nb - 1](xcat[:Ra].view(B, N + 1, D))
        vit_last = m._final_norm(xa)[:, 1:]
        last = ops.add_f32(outs[3].view(B, N, D), vit_last)
        maps = [last.view(B * N, D), outs[2], outs[1], outs[0]]
        dt = config.operand_dtype
        res = []
        for t in maps:
            hi = ops.cast_pad(t, D, dt).view(B, h, w, D)
            lo = ops.cast_pad(t, D, dt, part=1).view(B, h, w, D) if config.split_conv else None
            res.append((hi, lo))
        if taps is not None:
            taps.update(mla_inputs=[t.view(B, N, D) for t in maps])
        return res

    @torch.no_grad()
    def train_step(self, inp: torch.Tensor, target: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """One `train.py:268-436` (or `train_mla.py:260-407`) iteration; returns the loss as a 0-dim device
        tensor (no host sync)."""
        dec = self.seg_decoder
        S = config.loss_scale
        asaves = None
        if self.is_mla:
            asaves = [] if self.mode == "train_adapters" else None
            logits, saved = dec._forward_core(self.features_mla(inp, taps, asaves), save=True, training=True)
        elif self.train_backbone:
            cat, e2e = self._features_e2e(inp, taps)
            if self.stream_only:
                logits, saved = dec._forward_core(cat[0], cat[1], save=True, training=True, need_input_grad=True)
            else:
                logits, saved = dec._forward_core(cat[0], cat[1], save=True, training=True)
        else:
            asaves = [] if self.mode == "train_adapters" else None
            cat = self.features(inp, taps, asaves)
            if self.stream_only and asaves is not None:
                logits, saved = dec._forward_core(cat[0], cat[1], save=True, training=True, need_input_grad=True)
            else:
                logits, saved = dec._forward_core(cat[0], cat[1], save=True, training=True)
        target = target.long().contiguous()
        n_region, lmode, eps, n_ce = self.LOSSES[self.loss_kind]
        loss, coef, _ = ops.seg_loss_fwd(logits, target, n_region, lmode, eps, n_ce, None, S)
        dz = ops.seg_loss_bwd(logits, target, coef, n_region, lmode, n_ce, None)
        B, hh, ww, C = logits.shape
        r = ops.resize_bilinear_bwd(dz, hh, ww, config.operand_dtype, config.split_conv)
        d16, d_lo, bpart = r if config.split_conv else (r[0], None, r[1])
        world = world_size(self.process_group)
        inv = 1.0 / (S * world)  # gradient mean over ranks folded into the un-scaling (DDP semantics)
        self.reducer.begin()
        if self.mode == "train_adapters" and self.is_mla:
            dmaps = dec._backward_core(saved, d16, bpart, inv, self.bucket.views, stage_done=self.reducer.stage_done, d_lo=d_lo,
                                       need_input_grad=True)
            self.adapter_reducer.begin()
            dc0 = self._mla_adapter_backward(asaves, dmaps, inv)
            self.adapter_reducer.stage_done()
            if self.train_encoder:
                self.encoder_reducer.begin()
                self._encoder_backward(dc0, None, inv)
                self.encoder_reducer.stage_done()
                self.encoder_reducer.finish()
            self.adapter_reducer.finish()
        elif self.mode == "train_adapters":
            dcat = dec._backward_core(saved, d16, bpart, inv, self.bucket.views, stage_done=self.reducer.stage_done, d_lo=d_lo,
                                      need_input_grad=True)
            if self.stream_only and (self.train_backbone or self.train_encoder):
                # UNet head: the decoder input is the adapter stream alone; the encoder / backbone backward walks take the
                # FeatureDecoder layout [stream | c4 (padded) | pass-A feature], whose other two gradient slices are zero here
                Bq, hq, wq, Dq = dcat.shape
                full = ops.zeros((Bq, hq, wq, 3 * Dq), dcat.device, dcat.dtype)
                ops.copy_channels(dcat.view(-1, Dq), full.view(-1, 3 * Dq)[:, :Dq])
                dcat = full
            self.adapter_reducer.begin()
            if self.train_backbone:
                self.vit_reducer.begin()
                if self.train_encoder:
                    self.encoder_reducer.begin()
                self._e2e_backward(e2e, dcat, inv)
                self.vit_reducer.finish()
                if self.train_encoder:
                    self.encoder_reducer.finish()
                self.adapter_reducer.finish()
                self.reducer.finish()
                self.optimizer.step(1.0)
                if taps is not None:
                    taps.update(logits=logits, loss=loss)
                return loss.view(())
            dc0 = self._adapter_backward(asaves, dcat, inv)
            self.adapter_reducer.stage_done()
            if self.train_encoder:
                self.encoder_reducer.begin()
                self._encoder_backward(dc0, dcat, inv)
                self.encoder_reducer.stage_done()
                self.encoder_reducer.finish()
            self.adapter_reducer.finish()
        else:
            dec._backward_core(saved, d16, bpart, inv, self.bucket.views, stage_done=self.reducer.stage_done, d_lo=d_lo)
        self.reducer.finish()
        self.optimizer.step(1.0)
        if taps is not None:
            taps.update(logits=logits, loss=loss)
        return loss.view(())

    # ---- BASELINE config 4: the adapter flow with the backbone unfrozen ------------------------------------------------
    @torch.no_grad()
    def _features_e2e(self, inp: torch.Tensor, taps: Optional[dict] = None):
        """``features`` with everything kept for the backward: both ViT passes stacked along the rows run
        ``Block.forward_train_rows`` (so each block's weight gradients are later taken over the rows of both passes in
        one GEMM), the final norm's inputs of the last ``n_last_blocks`` outputs, CAViT / CACNN activations, the encoder's
        when it trains.  -> ((cat_hi, cat_lo|None), saved)."""
        config.split_attn_out = self.split_attn_out
        config.precise_level = self.precise_level
        config.precise_parts = self.precise_parts
        m = self.model
        B, _, H, W = inp.shape
        inp = inp.float().contiguous()
        D = m.embed_dim
        h, w = H // self.patch, W // self.patch
        N = h * w
        nb, nl = len(m.blocks), self.n_last_blocks
        if self.train_encoder:
            c, shapes, esaved = self.backbone_encoder.forward_tokens_train(inp)
            self._esaved = (esaved, shapes)
        else:
            _, c, shapes = self.backbone_encoder.forward_tokens(inp, need_c1=not config.elide_c1)
        c_orig = c
        Lc = c.shape[1]
        g = self._geometry(H, W, shapes, inp.device)
        tokens, a16 = m.patch_tokens_train(inp)           # shared by both passes: its gradient is the sum of both
        xa = ops.add_cls_pos(tokens, m.cls_token.detach().reshape(-1).float().contiguous(),
                             m._pos_for(N, H, W).detach().reshape(-1, D).float().contiguous())
        Ra, Rb = B * (N + 1), B * N
        segs = [(B, N + 1), (B, N)]
        xcat = torch.cat([xa.view(Ra, D), tokens.reshape(Rb, D)], 0)
        bsaves, feats, fin, asaves = [], [], [], []
        for i, blk in enumerate(m.blocks[: nb - (nl - 1)]):
            xcat, sv = blk.forward_train_rows(xcat, segs)
            bsaves.append(sv)
            if i >= nb - nl:
                feats.append(m._final_norm(xcat[:Ra].view(B, N + 1, D))[:, 1:])
                fin.append(xcat)                           # rows [:Ra] = input of that final norm (never overwritten)
        if taps is not None:
            taps.update(c=c, x_b0=xcat[Ra:].view(B, N, D).clone(), shapes=shapes)
        c2d = c.view(B * Lc, D)
        for s in range(nl):
            if s > 0:
                xcat, sv = m.blocks[nb - nl + s].forward_train_rows(xcat, segs)
                bsaves.append(sv)
                feats.append(m._final_norm(xcat[:Ra].view(B, N + 1, D))[:, 1:])
                fin.append(xcat)
            dead = s == nl - 1 and taps is None and config.elide_dead_cacnn   # see ``features``: the last CACNN output feeds nothing
            x2, s_cv = self.cross_vit.forward16_train(xcat[Ra:].clone(), c2d, g, B, N, Lc)
            s_cn = None
            if not dead:
                c2d, s_cn = self.cross_cnn.forward16_train(c2d, x2, g, B, Lc, N, shapes)
            asaves.append((s_cv, s_cn))
            x = ops.add_f32(x2.view(B, N, D), feats[s], out=xcat[Ra:].view(B, N, D))
        if taps is not None:
            taps.update(feats=feats)
        if self.stream_only:     # UNet head (BASELINE config 2 with everything trainable): the adapter stream alone
            xs = x.reshape(B * N, D)
            hi = ops.cast_pad(xs, D, config.operand_dtype).view(B, h, w, D)
            lo = ops.cast_pad(xs, D, config.operand_dtype, part=1).view(B, h, w, D) if config.split_conv else None
            if taps is not None:
                taps.update(x_final=x, c_final=c2d.view(B, Lc, D), cat=hi)
            return (hi, lo), (a16, bsaves, fin, asaves, (B, N, H, W))
        n4 = shapes[2][0] * shapes[2][1]
        cat = ops.decoder_input(x, c_orig[:, Lc - n4:], feats[-1], (h, w), shapes[2], config.operand_dtype, config.split_conv)
        if not config.split_conv:
            cat = (cat, None)
        if taps is not None:
            taps.update(x_final=x, c_final=c2d.view(B, Lc, D), cat=cat[0])
        return cat, (a16, bsaves, fin, asaves, (B, N, H, W))

    def _block_done(self, i: int) -> None:
        if i in self._fire_at:
            self.vit_reducer.stage_done()

    def _e2e_backward(self, saved, dcat: torch.Tensor, inv: float) -> None:
        """Backward of ``_features_e2e``.  G = gradient of the stacked token matrix [pass A rows | pass B rows] at the
        current point of the reverse walk.  Per adapter stage (last to first): the stage output x = CAViT(x_B, c) + f_s
        sends its gradient to the CAViT output and to f_s = final_norm(pass-A block output)[:, 1:] (LayerNorm backward
        into the pass-A rows; the last feature also receives the pass-A slice of the decoder input); CACNN (when it fed a
        later stage) and CAViT run their backward; then the block in front of the stage takes G for both passes at once.
        ``model.norm``, CAViT and CACNN are used once per stage: per-stage gradient slabs summed in a fixed order."""
        a16, bsaves, fin, asaves, (B, N, H, W) = saved
        m, cv, cn = self.model, self.cross_vit, self.cross_cnn
        _, h, w, D3 = dcat.shape
        D = D3 // 3
        nb, nl = len(m.blocks), self.n_last_blocks
        Ra, Rb = B * (N + 1), B * N
        dev = dcat.device
        ab, vg = self.adapter_bucket, self.vit_bucket.views
        slabs = ops.zeros((nl, ab.numel), dev)
        nslab = ops.zeros((nl, 2 * D), dev)
        nw = m.norm.weight.detach().float().contiguous()
        dcat2 = dcat.view(B * N, D3)
        G = ops.zeros((Ra + Rb, D), dev)
        ops.copy_channels(dcat2[:, :D], G[Ra:])
        dvit = torch.empty((B * N, D), device=dev, dtype=torch.float32)       # pass-A slice of the decoder input
        ops.copy_channels(dcat2[:, 2 * D:], dvit)
        dc_next = None
        for s in range(nl - 1, -1, -1):
            s_cv, s_cn = asaves[s]
            gs = {n: slabs[s, o:o + p.numel()].view(p.shape) for n, p, o in zip(ab.names, ab.params, ab.offsets)}
            # d f_s: the stage's residual add, + the decoder input's third slice for the last feature
            dyA = ops.zeros((B, N + 1, D), dev)
            ops.copy_channels(G[Ra:].view(B, N * D), dyA.view(B, (N + 1) * D)[:, D:])
            if s == nl - 1:
                ops.add_f32(dyA[:, 1:], dvit.view(B, N, D), out=dyA[:, 1:])
            Gn = torch.empty((Ra + Rb, D), device=dev, dtype=torch.float32)
            _, part = ops.layernorm_bwd(dyA.view(Ra, D), fin[s][:Ra], nw, m.norm.eps, res=G[:Ra], out=Gn[:Ra])
            ops.reduce_rows(part.view(part.shape[0], 2 * D), inv, nslab[s])
            dx = G[Ra:]
            dc_in = None
            if dc_next is not None:
                dc_in, dx_extra = cn.backward16(s_cn, dc_next, inv, gs, "cross_cnn")
                dx = ops.add_f32(dx.view(B, N, D), dx_extra.view(B, N, D)).view(Rb, D)
            dx_in, dc_cv = cv.backward16(s_cv, dx, inv, gs, "cross_vit")
            if dc_in is not None:
                Lc = dc_cv.shape[0] // B
                ops.add_f32(dc_cv.view(B, Lc, D), dc_in.view(B, Lc, D), out=dc_cv.view(B, Lc, D))
            dc_next = dc_cv
            ops.copy_channels(dx_in, Gn[Ra:])
            if s == 0:   # every use of model.norm / CAViT / CACNN has contributed: finish their gradients, start the exchanges
                red = ops.reduce_rows(nslab, 1.0)
                vg["norm.weight"].copy_(red[:D]); vg["norm.bias"].copy_(red[D:])
                self._block_done(nb)
                ops.reduce_rows(slabs, 1.0, ab.grad)
                self.adapter_reducer.stage_done()
                if self.train_encoder:
                    self._encoder_backward(dc_next, dcat, inv)
                    self.encoder_reducer.stage_done()
            bi = nb - nl + s
            G = m.blocks[bi].backward(bsaves[bi], Gn, inv, vg, f"blocks.{bi}")
            bsaves[bi] = None
            self._block_done(bi)
        for i in range(nb - nl - 1, -1, -1):
            G = m.blocks[i].backward(bsaves[i], G, inv, vg, f"blocks.{i}")
            bsaves[i] = None
            self._block_done(i)
        m.embed_backward(a16, G[:Ra].view(B, N + 1, D), G[Ra:], H, W, inv, vg)
        self._block_done(-1)

    def _adapter_backward(self, asaves, dcat: torch.Tensor, inv: float) -> None:
        """(``dcat`` [B, h, w, 3D] = gradient of the FeatureDecoder input concat, or [B, h, w, D] = gradient of the
        adapter-stream map alone for the UNet head.)
        Backward of the four adapter stages (`train.py:304-387` under autograd, minus its no_grad):
        d cat[..., :D] is the gradient of the adapter stream; the c4 and pass-A slices of the decoder input come from
        the frozen encoder / backbone.  Per stage, in reverse: x = x2 + feat (identity), CACNN (only when its output
        fed a later stage: the last one is dead code in the reference flow), CAViT, and the input gradient of the
        frozen ViT block in front of the stage.  CAViT / CACNN are the same modules in all four stages, so every
        stage writes its own gradient slab and the slabs are summed in a fixed order."""
        m, cv, cn = self.model, self.cross_vit, self.cross_cnn
        B, h, w, D3 = dcat.shape
        D = m.embed_dim
        N = h * w
        nl = self.n_last_blocks
        nb = len(m.blocks)
        ab = self.adapter_bucket
        slabs = ops.zeros((nl, ab.numel), dcat.device)
        dx = torch.empty((B * N, D), device=dcat.device, dtype=torch.float32)
        ops.copy_channels(dcat.view(B * N, D3)[:, :D], dx)
        dc_next = None
        for s in range(nl - 1, -1, -1):
            bsaved, s_cv, s_cn = asaves[s]
            gs = {n: slabs[s, o:o + p.numel()].view(p.shape) for n, p, o in zip(ab.names, ab.params, ab.offsets)}
            dc_in = None
            if dc_next is not None:
                dc_in, dx_extra = cn.backward16(s_cn, dc_next, inv, gs, "cross_cnn")
                ops.add_f32(dx.view(B, N, D), dx_extra.view(B, N, D), out=dx.view(B, N, D))
            dx_in, dc_cv = cv.backward16(s_cv, dx, inv, gs, "cross_vit")
            if dc_in is not None:
                Lc = dc_cv.shape[0] // B
                ops.add_f32(dc_cv.view(B, Lc, D), dc_in.view(B, Lc, D), out=dc_cv.view(B, Lc, D))
            dc_next = dc_cv
            if s > 0:
                dx = m.blocks[nb - (nl - 1) + s - 1].backward(bsaved, dx_in, inv, None)
        ops.reduce_rows(slabs, 1.0, ab.grad)
        return dc_next  # gradient of the encoder's pyramid tokens c (stage-0 input), fp32 [B*Lc, D]

    def _mla_adapter_backward(self, asaves, dmaps, inv: float) -> torch.Tensor:
        """Backward of the four adapter uses of the `train_mla.py:300-383` flow: x0 = CAViT(x, c0); for j = 1..3:
        x = block_j(x_{j-1}), c_j = CACNN(c_{j-1}, x), x_j = CAViT(x, c_j); MLA inputs (x3 + f, x2, x1, x0).
        ``dmaps``: gradients of the four MLA inputs in decoder argument order, fp32 [B, h, w, D], times the loss scale.
        Writes the adapter bucket; -> gradient of the encoder's pyramid tokens c0, fp32 [B * Lc, D]."""
        m, cv, cn = self.model, self.cross_vit, self.cross_cnn
        nb = len(m.blocks)
        B, h, w, D = dmaps[0].shape
        N = h * w
        ab = self.adapter_bucket
        slabs = ops.zeros((4, ab.numel), dmaps[0].device)
        d_out = [dmaps[3], dmaps[2], dmaps[1], dmaps[0]]          # gradients of x0, x1, x2, x3
        dx = d_out[3].reshape(B * N, D).contiguous()
        dc_from_next = None                                        # gradient of c_j through CACNN_{j+1}'s query input
        for j in (3, 2, 1):
            bsaved, s_cv, s_cn = asaves[j]
            gs = {n: slabs[j, o:o + p.numel()].view(p.shape) for n, p, o in zip(ab.names, ab.params, ab.offsets)}
            dx_q, dc_j = cv.backward16(s_cv, dx, inv, gs, "cross_vit")          # query (block output) and feat (c_j)
            if dc_from_next is not None:
                Lc = dc_j.shape[0] // B
                ops.add_f32(dc_j.view(B, Lc, D), dc_from_next.view(B, Lc, D), out=dc_j.view(B, Lc, D))
            dc_from_next, dx_f = cn.backward16(s_cn, dc_j, inv, gs, "cross_cnn")  # c_{j-1} and the block output as CACNN's feat
            dxb = ops.add_f32(dx_q.view(B, N, D), dx_f.view(B, N, D)).view(B * N, D)
            bi = (nb - 3, nb - 2, nb - 2)[j - 1]
            dx_prev = m.blocks[bi].backward(bsaved, dxb, inv, None)
            dx = ops.add_f32(dx_prev.view(B, N, D), d_out[j - 1].reshape(B, N, D)).view(B * N, D)
        _, s_cv, _ = asaves[0]
        gs = {n: slabs[0, o:o + p.numel()].view(p.shape) for n, p, o in zip(ab.names, ab.params, ab.offsets)}
        _, dc0 = cv.backward16(s_cv, dx, inv, gs, "cross_vit")
        Lc = dc0.shape[0] // B
        ops.add_f32(dc0.view(B, Lc, D), dc_from_next.view(B, Lc, D), out=dc0.view(B, Lc, D))
        ops.reduce_rows(slabs, 1.0, ab.grad)
        return dc0

    def _encoder_backward(self, dc0: torch.Tensor, dcat: torch.Tensor, inv: float) -> None:
        """d c = what the adapter stages send back + the c4 slice of the decoder input (`train.py:395-400`: c4 is
        zero-padded, centred, into channels [D, 2D) of the concat) -> FeatureEncoder.backward_tokens."""
        esaved, shapes = self._esaved
        self._esaved = None
        D = self.model.embed_dim
        if dcat is None:            # train_mla.py flow: c reaches the head through the adapters only
            dc = dc0.view(-1, sum(a * b for a, b in shapes), D)
        else:
            B, h, w, D3 = dcat.shape
            h4, w4 = shapes[2]
            n4 = h4 * w4
            Lc = dc0.shape[0] // B
            top, left = (h - h4) // 2, (w - w4) // 2
            d4 = dcat[:, top:top + h4, left:left + w4, D:2 * D].reshape(B, n4, D).contiguous()
            dc = dc0.view(B, Lc, D)
            ops.add_f32(dc[:, Lc - n4:], d4, out=dc[:, Lc - n4:])
        gviews = {n[len("backbone_encoder."):]: v for n, v in self.encoder_bucket.views.items()}
        self.backbone_encoder.backward_tokens(esaved, dc, inv, gviews)

    @torch.no_grad()
    def eval_logits(self, inp: torch.Tensor) -> torch.Tensor:
        """Decoder logits, NHWC fp32, with the decoder's BatchNorm in its current train/eval mode."""
        if self.is_mla:
            logits, _ = self.seg_decoder._forward_core(self.features_mla(inp), save=False)
        else:
            logits, _ = self.seg_decoder._forward_core(*self.features(inp), save=False)
        return logits

    @torch.no_grad()
    def validate_step(self, inp: torch.Tensor, target: torch.Tensor, ce_weight: Optional[torch.Tensor] = None,
                      with_counts: bool = False):
        """`train.py:465-642` for one batch: -> device tensor [4] = (weighted CE, dice, pixel accuracy, n_images).
        The decoder runs in eval mode (running BatchNorm statistics, `train.py:451`); the encoder's SyncBatchNorm
        stays in train mode exactly like the reference."""
        was = self.seg_decoder.training
        self.seg_decoder.eval()
        logits = self.eval_logits(inp)
        self.seg_decoder.train(was)
        target = target.long().contiguous()
        if with_counts:  # + per-class pixel counts [C,3] for ch_iou / isi_iou (train_multi_class.py:582-589)
            m, counts = ops.ce_acc(logits, target, ce_weight, counts=True)
        else:
            m = ops.ce_acc(logits, target, ce_weight)
        loss1, _, _ = ops.dice_fwd(logits, target, 1, 10e-20, 1.0)
        return (m, loss1, counts) if with_counts else (m, loss1)


def make_vit_bucket(model, blocks_per_bucket: int, process_group, momentum: bool = False, min_first_blocks: int = 0,
                    compress: Optional[str] = None):
    """Flat gradient bucket of the whole backbone in gradient-ready order (final norm, blocks last..first, then the token
    embedding parameters) + a reducer over ``blocks_per_bucket``-block chunks (the last chunk takes the embeddings along).
    -> (bucket, reducer, fire_at) with fire_at = block indices after whose backward the next chunk is complete (-1 = after
    the embedding backward).  ``min_first_blocks``: the first chunk (which holds ``norm.*``) does not fire before that many
    blocks are done — the adapter flow writes the final-norm gradients only after its ``n_last_blocks`` stages, so a smaller
    first chunk would be all-reduced before they exist."""
    vnamed = dict(model.named_parameters())
    depth = len(model.blocks)
    groups = [[n for n in vnamed if n.startswith("norm.")]]
    groups += [[n for n in vnamed if n.startswith(f"blocks.{i}.")] for i in range(depth - 1, -1, -1)]
    groups.append([n for n in vnamed if not n.startswith("norm.") and not n.startswith("blocks.")])
    assert sum(len(g) for g in groups) == len(vnamed)
    for p in model.parameters():
        p.requires_grad_(True)
    bucket = FlatBucket([(n, vnamed[n]) for g in groups for n in g], momentum=momentum)
    fire_at, ranges, start = [], [], 0
    for j in range(1, depth + 1):                       # j = number of blocks finished
        if j % blocks_per_bucket == 0 and j != depth and j >= min_first_blocks:
            names = [n for g in groups[start:j + 1] for n in g]
            ranges.append(bucket.range_of(names)); fire_at.append(depth - j); start = j + 1
    ranges.append(bucket.range_of([n for g in groups[start:] for n in g])); fire_at.append(-1)
    return bucket, StageReducer(bucket.grad, ranges, process_group, compress=compress), fire_at


class EndToEndEngine(nn.Module):
    """BASELINE config 4 — the unfrozen end-to-end variant (`eval/eval_dinov2_setr_cross_ete.py:145-148,307-361`):

        x_norm_patchtokens = model(inp, is_training=True)          whole ViT under autograd / DDP
        logits = seg_decoder(tokens as (B, D, h, w)) ; resize to the label size ; loss = CE + DC(2)
        backward through the decoder AND all ViT blocks ; DDP all-reduces every gradient
        optimizer.step() on the decoder only (`:224-229` — the backbone gradients are computed and exchanged, then
        discarded; reproduced as written, DESIGN.md "config 4").

    The ViT runs ``forward_train`` / ``backward`` (fused attention forward+backward, LayerNorm / GELU / LayerScale
    backward kernels, weight-gradient GEMMs); its gradients live in a gradient-only flat bucket that is all-reduced
    in ``blocks_per_bucket``-block chunks on the side stream while earlier blocks are still in their backward.
    """

    def __init__(self, model, seg_decoder, *, lr: float = 0.01, momentum: float = 0.9, weight_decay: float = 0.0,
                 loss: str = "ce_dc", process_group=None, blocks_per_bucket: int = 4):
        super().__init__()
        if loss not in SegEngine.LOSSES:
            raise ValueError(f"loss must be one of {sorted(SegEngine.LOSSES)}")
        self.model, self.seg_decoder, self.loss_kind, self.process_group = model, seg_decoder, loss, process_group
        self.patch = model.patch_size
        for p in model.parameters():
            p.requires_grad_(True)
        order = list(seg_decoder.GRAD_ORDER)
        named = dict(seg_decoder.named_parameters())
        ordered = [(n, named[n]) for pre in order for n in named if n.startswith(pre + ".")]
        assert len(ordered) == len(named)
        self.bucket = FlatBucket(ordered)
        self.stage_ranges = [self.bucket.range_of([n for n in named if n.startswith(pre + ".")]) for pre in order]
        self.optimizer = SGD([self.bucket], lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.reducer = StageReducer(self.bucket.grad, self.stage_ranges, process_group)
        # backbone: gradient-ready order = final norm, blocks last..first, then the token embedding parameters; reduced
        # after every `blocks_per_bucket` blocks
        self.vit_bucket, self.vit_reducer, self._fire_at = make_vit_bucket(model, blocks_per_bucket, process_group)

    def _block_done(self, i: int) -> None:
        if i in self._fire_at:
            self.vit_reducer.stage_done()

    @torch.no_grad()
    def train_step(self, inp: torch.Tensor, target: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        m, dec = self.model, self.seg_decoder
        S = config.loss_scale
        dt = config.operand_dtype
        B, _, H, W = inp.shape
        h, w = H // self.patch, W // self.patch
        N, D = h * w, m.embed_dim
        tok, vsaved = m.forward_train(inp)
        t2 = torch.empty((B * N, D), device=inp.device, dtype=torch.float32)
        ops.copy_channels(vsaved_tokens_view(tok, B, N, D), t2.view(B, N * D))
        hi = ops.cast_pad(t2, D, dt).view(B, h, w, D)
        lo = ops.cast_pad(t2, D, dt, part=1).view(B, h, w, D) if config.split_conv else None
        logits, saved = dec._forward_core(hi, lo, save=True, training=True)
        target = target.long().contiguous()
        n_region, lmode, eps, n_ce = SegEngine.LOSSES[self.loss_kind]
        loss, coef, _ = ops.seg_loss_fwd(logits, target, n_region, lmode, eps, n_ce, None, S)
        dz = ops.seg_loss_bwd(logits, target, coef, n_region, lmode, n_ce, None)
        _, hh, ww, C = logits.shape
        r = ops.resize_bilinear_bwd(dz, hh, ww, dt, config.split_conv)
        d16, d_lo, bpart = r if config.split_conv else (r[0], None, r[1])
        inv = 1.0 / (S * world_size(self.process_group))
        self.reducer.begin()
        self.vit_reducer.begin()
        dX = dec._backward_core(saved, d16, bpart, inv, self.bucket.views, stage_done=self.reducer.stage_done, d_lo=d_lo,
                                need_input_grad=True)
        m.backward(vsaved, dX.view(B, N, D), inv, self.vit_bucket.views, block_done=self._block_done)
        self.reducer.finish()
        self.vit_reducer.finish()
        self.optimizer.step(1.0)
        if taps is not None:
            taps.update(logits=logits, loss=loss, tokens=t2.view(B, N, D))
        return loss.view(())


def vsaved_tokens_view(tok: torch.Tensor, B: int, N: int, D: int) -> torch.Tensor:
    """x_norm_patchtokens is the [:, 1:] view of the (B, N+1, D) normalised tokens: as [B, N*D] rows with the batch
    stride of the full tensor (what asis_copy_channels compacts)."""
    return tok.as_strided((B, N * D), (tok.stride(0), 1))
