"""HIP-backed ``FeatureEncoder`` (spatial-prior CNN) — API / ``state_dict`` mirror of
`backbones/encoders.py:4-74`.

torch ``nn.Conv2d`` / ``nn.BatchNorm2d`` modules are kept only as parameter containers (same
``state_dict`` keys as the reference, incl. ``running_mean/var/num_batches_tracked``); the forward
is: stem conv (Cin=3) direct kernel, every other 3x3 conv an implicit-GEMM MFMA kernel on NHWC
16-bit activations with BatchNorm statistics taken from its epilogue, BN(+ReLU)(+MaxPool) fused
apply kernels, and the 1x1 ``fc*`` projections as GEMMs that write straight into the
``[B, 5329+1296+324, D]`` token buffer the adapters consume (`train.py:283` concat is free).

BatchNorm runs in TRAIN mode (batch statistics, running-stat update) exactly like the reference,
which never calls ``.eval()`` on the encoder; with an initialised process group the statistics are
all-reduced (SyncBatchNorm).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import config, ops, parallel
from ..dinov2.layers.blocks import _Packed, _pack
from . import _bn


# ASIS_ENC_MX=0: the stem convolutions on three 16-bit parts (implicit GEMM) instead of MX operands on the halo-tile kernel
_ENC_MX = __import__("os").environ.get("ASIS_ENC_MX", "1") not in ("0", "")

class FeatureEncoder(_Packed):
    def __init__(self, inplanes=64, embed_dim=1024, with_cp=False):
        super().__init__()
        if inplanes % 8:
            raise ValueError("inplanes must be a multiple of 8")
        self.with_cp = with_cp
        self.inplanes, self.embed_dim = inplanes, embed_dim
        self.compute_c1 = True  # the reference computes c1 although train.py never uses it
        BN = nn.BatchNorm2d  # same state_dict keys as nn.SyncBatchNorm
        self.stem = nn.Sequential(
            nn.Conv2d(3, inplanes, kernel_size=3, stride=2, padding=1, bias=False), BN(inplanes), nn.ReLU(inplace=True),
            nn.Conv2d(inplanes, inplanes, kernel_size=3, stride=1, padding=1, bias=False), BN(inplanes), nn.ReLU(inplace=True),
            nn.Conv2d(inplanes, inplanes, kernel_size=3, stride=1, padding=1, bias=False), BN(inplanes), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        self.conv2 = nn.Sequential(nn.Conv2d(inplanes, 2 * inplanes, kernel_size=3, stride=2, padding=0, bias=False),
                                   BN(2 * inplanes), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(nn.Conv2d(2 * inplanes, 4 * inplanes, kernel_size=3, stride=2, padding=0, bias=False),
                                   BN(4 * inplanes), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(nn.Conv2d(4 * inplanes, 8 * inplanes, kernel_size=3, stride=2, padding=1, bias=False),
                                   BN(8 * inplanes), nn.ReLU(inplace=True))
        self.fc1 = nn.Conv2d(inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)
        self.fc2 = nn.Conv2d(2 * inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)
        self.fc3 = nn.Conv2d(4 * inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)
        self.fc4 = nn.Conv2d(8 * inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)

    # -- helpers ---------------------------------------------------------------------------------
    def _wconv(self, key, conv, part=0):
        return _pack(self._cache, f"{key}.p{part}", conv.weight,
                     lambda p: ops.pack_conv_weight(p.float().contiguous(), 0, config.operand_dtype, part))

    def _conv_bn(self, a, key, conv, bn, sync):
        """a = (hi, lo|None) NHWC 16-bit -> (raw fp32 NHWC, scale, shift)."""
        x16, x_lo = a
        if key in config.unsplit_layers:
            x_lo = None                     # lab switch: this layer's conv on plain 16-bit operands
        B, H, W, _ = x16.shape
        s, p = conv.stride[0], conv.padding[0]
        OH, OW = (H + 2 * p - 3) // s + 1, (W + 2 * p - 3) // s + 1
        stats = torch.empty((ops.gemm_tiles_m(B * OH * OW), 2, conv.out_channels), device=x16.device, dtype=torch.float32)
        mx_in = ops.mx_amax_of(x_lo)
        if mx_in is not None:
            # MX operand planes (the 64 -> 64 stem convolutions at 294^2): the halo-tile kernel, BatchNorm partial sums included
            w_mx, w_amax = _pack(self._cache, f"{key}.mx", conv.weight, lambda q: ops.pack_conv_weight_mx(q.float().contiguous(), 0, config.operand_dtype))
            raw, stats = ops.conv3x3_halo_mx(x16, x_lo, self._wconv(key, conv), w_mx, (mx_in, w_amax), want_stats=True)
        elif x_lo is not None:
            raw = ops.conv_gemm_split(x16, x_lo, self._wconv(key, conv), self._wconv(key, conv, 1), 3, 3, s, p, stats=stats)
        else:
            raw = ops.conv_gemm(x16, self._wconv(key, conv), 3, 3, s, p, stats=stats)
        scale, shift, _, _, _ = _bn.finalize(stats, B * OH * OW, bn, sync)
        return raw, scale, shift

    @staticmethod
    def _pair(r, split):
        return r if split else (r, None)

    def _fc(self, i, fc, a, out):
        """1x1 conv as a (split-precision) GEMM writing into its slice of the token buffer."""
        hi, lo = a
        B, h, w, Cs = hi.shape
        w_hi = self._w16(f"fc{i}", fc.weight)
        bias = self._f32(f"fc{i}_b", fc.bias)
        if lo is not None:
            w_lo = _pack(self._cache, f"fc{i}.lo", fc.weight,
                         lambda p: ops.cast_pad(p.reshape(p.shape[0], -1).contiguous().float(),
                                                dtype=config.operand_dtype, part=1))
            ops.gemm_split(hi.view(B, h * w, Cs), lo.view(B, h * w, Cs), w_hi, w_lo, out=out, bias_n=bias)
        else:
            ops.gemm(hi.view(B, h * w, Cs), w_hi, out=out, bias_n=bias)

    def forward_tokens(self, x, need_c1=False, sync_bn=True):
        """-> (c1 NHWC fp32 or None, c tokens fp32 [B, n2+n3+n4, D], [(h2,w2),(h3,w3),(h4,w4)])."""
        dt = config.operand_dtype
        sp = config.split_conv
        x = x.float().contiguous()
        B = x.shape[0]
        D = self.embed_dim
        st = self.stem
        raw = ops.conv3x3_c3(x, self._f32("stem0", st[0].weight), 2, 1)
        scale, shift, _, _, _ = _bn.finalize(ops.colstats(raw), raw.numel() // raw.shape[-1], st[1], sync_bn)
        # the two 64 -> 64 stem convolutions (stride 1, pad 1, 294^2 for a 588^2 image: 77 GF each, the encoder's largest) take MX
        # lo operands on the halo-tile kernel where that path is on: their inputs come out of bn_act in the MX form
        def stem_act(raw, scale, shift, conv):
            mx = (sp and _ENC_MX and config.mx_conv_on() and ops.CONV_HALO and dt == torch.float16 and conv.stride[0] == 1 and conv.padding[0] == 1 and
                  conv.in_channels % 64 == 0 and conv.out_channels in (64, 128))
            if mx:
                return ops.bn_act(raw, scale, shift, True, dt, True, mx_amax=ops.bn_relu_absmax(raw, scale, shift))
            return self._pair(ops.bn_act(raw, scale, shift, True, dt, sp), sp)
        a = stem_act(raw, scale, shift, st[3])
        raw, scale, shift = self._conv_bn(a, "stem3", st[3], st[4], sync_bn)
        a = stem_act(raw, scale, shift, st[6])
        raw, scale, shift = self._conv_bn(a, "stem6", st[6], st[7], sync_bn)
        s1 = self._pair(ops.bn_relu_maxpool(raw, scale, shift, dt, sp), sp)  # [B,147,147,C]
        raw, scale, shift = self._conv_bn(s1, "conv2", self.conv2[0], self.conv2[1], sync_bn)
        s2 = self._pair(ops.bn_act(raw, scale, shift, True, dt, sp), sp)
        raw, scale, shift = self._conv_bn(s2, "conv3", self.conv3[0], self.conv3[1], sync_bn)
        s3 = self._pair(ops.bn_act(raw, scale, shift, True, dt, sp), sp)
        raw, scale, shift = self._conv_bn(s3, "conv4", self.conv4[0], self.conv4[1], sync_bn)
        s4 = self._pair(ops.bn_act(raw, scale, shift, True, dt, sp), sp)
        shapes = [tuple(t[0].shape[1:3]) for t in (s2, s3, s4)]
        sizes = [h * w for h, w in shapes]
        ntok = sum(sizes)
        c = torch.empty((B, ntok, D), device=x.device, dtype=torch.float32)
        off = 0
        for i, (s, fc) in enumerate(((s2, self.fc2), (s3, self.fc3), (s4, self.fc4))):
            self._fc(i + 2, fc, s, c[:, off:off + sizes[i]])
            off += sizes[i]
        c1 = None
        if need_c1:
            h1, w1 = s1[0].shape[1:3]
            c1 = torch.empty((B, h1 * w1, D), device=x.device, dtype=torch.float32)
            self._fc(1, self.fc1, s1, c1)
            c1 = c1.view(B, h1, w1, D)
        return c1, c, shapes

    # ---- training path (`train_adapters` mode with the encoder in the trainable set) ----------------------------------
    def forward_tokens_train(self, x, sync_bn=True):
        """``forward_tokens`` (c1 is never used by the step) keeping what ``backward_tokens`` needs.
        -> (c tokens fp32 [B, n2+n3+n4, D], shapes, saved)."""
        from .decoders import conv_bn_relu_up_forward as stage
        dt = config.operand_dtype
        sp = config.split_conv
        x = x.float().contiguous()
        B, _, H, W = x.shape
        D = self.embed_dim
        st = self.stem
        raw0 = ops.conv3x3_c3(x, self._f32("stem0", st[0].weight), 2, 1)
        scale, shift, mean, invstd, count = _bn.finalize(ops.colstats(raw0), raw0.numel() // raw0.shape[-1], st[1], sync_bn)
        a = self._pair(ops.bn_act(raw0, scale, shift, True, dt, sp), sp)
        # the image as an 8-channel 16-bit NHWC operand of the stem's weight-gradient GEMM (channels 3..7 zero)
        x8 = ops.cast_pad(x.permute(0, 2, 3, 1).contiguous().view(B * H * W, 3), 8, dt).view(B, H, W, 8)
        saved = {"stem0": (x8, raw0, scale, shift, mean, invstd, count)}
        a, saved["stem3"] = stage(self, "stem3", a[0], a[1], st[3], st[4], 1, sync_bn, True)
        s1, saved["stem6"] = stage(self, "stem6", a[0], a[1], st[6], st[7], 1, sync_bn, True, pool=True)
        s2, saved["conv2"] = stage(self, "conv2", s1[0], s1[1], self.conv2[0], self.conv2[1], 1, sync_bn, True, stride=2, pad=0)
        s3, saved["conv3"] = stage(self, "conv3", s2[0], s2[1], self.conv3[0], self.conv3[1], 1, sync_bn, True, stride=2, pad=0)
        s4, saved["conv4"] = stage(self, "conv4", s3[0], s3[1], self.conv4[0], self.conv4[1], 1, sync_bn, True, stride=2, pad=1)
        shapes = [tuple(t[0].shape[1:3]) for t in (s2, s3, s4)]
        sizes = [h * w for h, w in shapes]
        c = torch.empty((B, sum(sizes), D), device=x.device, dtype=torch.float32)
        off = 0
        for i, (s, fc) in enumerate(((s2, self.fc2), (s3, self.fc3), (s4, self.fc4))):
            self._fc(i + 2, fc, s, c[:, off:off + sizes[i]])
            off += sizes[i]
        saved["maps"] = (s2[0], s3[0], s4[0])
        saved["sizes"] = sizes
        return c, shapes, saved

    def backward_tokens(self, saved, dc: torch.Tensor, inv_scale: float, grads: dict, prefix: str = "", sync_bn: bool = True):
        """dc fp32 [B, n2+n3+n4, D] = loss_scale * dL/dc -> every parameter gradient of the encoder into ``grads``
        (``fc1`` feeds only the unused c1: zero).  Stride-2 stages: weight gradient with the strided window geometry,
        input gradient as a stride-1 conv of the zero-inserted gradient; the stem's MaxPool through its argmax."""
        from .decoders import conv_bn_relu_up_backward as stage_bwd
        dt = config.operand_dtype
        pre = prefix + "." if prefix else ""
        B = dc.shape[0]
        D = self.embed_dim
        sizes = saved["sizes"]
        # ---- fc2..fc4: tokens = map . W^T + b
        dmaps = []
        off = 0
        for i, (m16, fc) in enumerate(zip(saved["maps"], (self.fc2, self.fc3, self.fc4))):
            _, h, w, Cs = m16.shape
            d2 = torch.empty((B * sizes[i], D), device=dc.device, dtype=torch.float32)
            ops.copy_channels(dc.as_strided((B, sizes[i] * D), (dc.stride(0), 1), dc.storage_offset() + off * D),
                              d2.view(B, sizes[i] * D))
            d16, cs = ops.cast_colsum(d2, dt)
            name = f"{pre}fc{i + 2}"
            gw = ops.wgrad(d16.view(1, B * sizes[i], 1, D), m16.view(1, B * sizes[i], 1, Cs), D, 1, 1, 1, 0, inv_scale)
            grads[name + ".weight"].view(D, Cs).copy_(gw.view(D, Cs))
            ops.reduce_rows(cs, inv_scale, grads[name + ".bias"])
            wT = self._wT16(f"fc{i + 2}T", fc.weight.view(D, Cs))
            dmaps.append(ops.gemm(d16, wT, out_f32=True).view(B, h, w, Cs))
            off += sizes[i]
        grads[pre + "fc1.weight"].zero_(); grads[pre + "fc1.bias"].zero_()
        # ---- conv4 <- conv3 <- conv2 <- stem (each map also feeds its fc projection)
        d3 = stage_bwd(self, "conv4", saved["conv4"], dmaps[2], self.conv4[0], self.conv4[1], inv_scale, grads, pre + "conv4",
                       True, sync_bn)
        ops.add_f32(d3.view(B, -1, d3.shape[-1]), dmaps[1].view(B, -1, d3.shape[-1]), out=d3.view(B, -1, d3.shape[-1]))
        d2_ = stage_bwd(self, "conv3", saved["conv3"], d3, self.conv3[0], self.conv3[1], inv_scale, grads, pre + "conv3", True,
                        sync_bn)
        ops.add_f32(d2_.view(B, -1, d2_.shape[-1]), dmaps[0].view(B, -1, d2_.shape[-1]), out=d2_.view(B, -1, d2_.shape[-1]))
        d1 = stage_bwd(self, "conv2", saved["conv2"], d2_, self.conv2[0], self.conv2[1], inv_scale, grads, pre + "conv2", True,
                       sync_bn)
        da = stage_bwd(self, "stem6", saved["stem6"], d1, self.stem[6], self.stem[7], inv_scale, grads, pre + "stem", True,
                       sync_bn, conv_name=pre + "stem.6", bn_name=pre + "stem.7")
        d0 = stage_bwd(self, "stem3", saved["stem3"], da, self.stem[3], self.stem[4], inv_scale, grads, pre + "stem", True,
                       sync_bn, conv_name=pre + "stem.3", bn_name=pre + "stem.4")
        # ---- stem conv 0 (3 -> inplanes, stride 2): BatchNorm / ReLU backward, weight gradient on the 8-channel image
        import torch.distributed as dist
        x8, raw0, scale, shift, mean, invstd, count = saved["stem0"]
        C0 = raw0.shape[-1]
        g, partial = ops.upsample_bn_relu_bwd(d0, raw0, scale, shift, mean, invstd, 1)
        red = ops.reduce_rows(partial.view(partial.shape[0], 2 * C0))
        local = red
        if sync_bn and parallel.bn_collectives_on():
            local = red.clone()  # gamma / beta gradients are local sums (averaged with the bucket), dx uses the global ones
            dist.all_reduce(red)
        dx16, _ = ops.bn_bwd_apply(g, raw0, mean, invstd, self._f32("stem1.g", self.stem[1].weight), red[C0:], red[:C0], count, dt)
        ops.reduce_rows(local[:C0].view(1, C0), inv_scale, grads[pre + "stem.1.bias"])
        ops.reduce_rows(local[C0:].view(1, C0), inv_scale, grads[pre + "stem.1.weight"])
        gw = ops.wgrad(dx16, x8, C0, 3, 3, 2, 1, inv_scale)                 # [C0, 8, 3, 3]
        grads[pre + "stem.0.weight"].copy_(gw[:, :3])

    def forward(self, x):
        """`encoders.py:49-74`: returns (c1 map (B,D,H/4,W/4), c2, c3, c4 token tensors)."""
        c1, c, shapes = self.forward_tokens(x, need_c1=self.compute_c1)
        n2, n3 = shapes[0][0] * shapes[0][1], shapes[1][0] * shapes[1][1]
        c2, c3, c4 = c[:, :n2], c[:, n2:n2 + n3], c[:, n2 + n3:]
        self.last_shapes = shapes
        self.last_tokens = c  # the concatenated buffer (train.py:283) for callers that want to skip torch.cat
        return (c1.permute(0, 3, 1, 2) if c1 is not None else None), c2, c3, c4
