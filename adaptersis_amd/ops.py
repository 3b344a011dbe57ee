"""Tensor-level wrappers over the C ABI (include/asis_hip.h).

torch is used here only as the owner of device memory and of the current HIP stream: every
function takes CUDA(=HIP) tensors, passes raw pointers/sizes to libasis_hip.so and returns
tensors allocated with ``torch.empty``.  A CPU tensor, a missing library or a bad shape raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_GELU_GRAD, ACT_NONE, ACT_RELU, ACT_SILU_MUL, GemmDesc, check, lib  # noqa: F401

T16_DEFAULT = torch.float16

# bench.py sets this to a list to time every GEMM launch with HIP events recorded on the launch stream:
# entries are (kind, algorithmic_flops, start_event, end_event, algorithmic_bytes).  None = no instrumentation.
# Algorithmic = 2*M*N*K and one read of each operand + one write of the output, also for split-precision launches
# (their two extra products are this build's precision choice, not work the reference asks for).
PROFILE = None


def _launch_timed(kind: str, flops: float, fn, nbytes: float = 0.0):
    if PROFILE is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn()
    e.record()
    PROFILE.append((kind, flops, s, e, nbytes))
    return rc


def _dt(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return _lib.ASIS_F16
    if dtype == torch.bfloat16:
        return _lib.ASIS_BF16
    raise ValueError(f"operand dtype must be float16 or bfloat16, got {dtype}")


def _dev(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AsisError("adaptersis_amd ops run on an MI355X (HIP) device only; got a CPU tensor "
                                 "(there is no CPU fallback)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("expected a contiguous float32 tensor")
    return t


def _gemm_desc(a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None, out_f32: bool = False,
               bias_n: Optional[torch.Tensor] = None, bias_m: Optional[torch.Tensor] = None,
               scale_n: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, act: int = ACT_NONE,
               stats: Optional[torch.Tensor] = None, a_lo: Optional[torch.Tensor] = None,
               b_lo: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None,
               out_lo: Optional[torch.Tensor] = None, rowstats: Optional[torch.Tensor] = None, res16=None, ln=None, mx=None):
    """-> (filled GemmDesc, out tensor, algorithmic flops, algorithmic bytes) of one ``gemm`` call (see ``gemm``).
    LayerNorm-fold chain (include/asis_hip.h): ``out_lo`` = second 16-bit plane of the output, ``rowstats`` = fp32
    [M, ceil(N / 64), 2] per-row partial sums, ``res16`` = (hi, lo) planes of the residual, ``ln`` = (mr [rows, 2], cs, cols)."""
    _dev(a, b, out, bias_n, bias_m, scale_n, res, stats, out_lo, rowstats)
    if a.dtype != b.dtype:
        raise ValueError("gemm operands must have the same 16-bit dtype")
    batch = 1
    if a.dim() == 3 or b.dim() == 3:
        batch = a.shape[0] if a.dim() == 3 else b.shape[0]
    M, K = a.shape[-2], a.shape[-1]
    N = b.shape[-2]
    if b.shape[-1] != K:
        raise ValueError(f"gemm: inner dims differ: a[...,{K}] vs b[...,{b.shape[-1]}]")
    if a.stride(-1) != 1 or b.stride(-1) != 1:
        raise ValueError("gemm: K must be contiguous in both operands")
    n_out = N // 2 if act == ACT_SILU_MUL else N      # SwiGLU epilogue: b = interleaved w12 (swiglu_rows), out = [M, Hd]
    if act == ACT_SILU_MUL and (out_f32 or batch != 1 or N % 32):
        raise ValueError("gemm: ACT_SILU_MUL writes a 16-bit [M, N / 2] matrix of one unbatched launch (N % 32 == 0)")
    if out is None:
        shape = (batch, M, n_out) if (a.dim() == 3 or b.dim() == 3) else (M, n_out)
        out = torch.empty(shape, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    else:
        out_f32 = out.dtype == torch.float32
        if out.stride(-1) != 1 or out.shape[-1] != n_out:
            raise ValueError("gemm: out must be contiguous in its last dim, one column per output feature")
    d = GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.lda, d.ldb, d.ldc = a.stride(-2), b.stride(-2), out.stride(-2)
    d.strideA = a.stride(0) if a.dim() == 3 else 0
    d.strideB = b.stride(0) if b.dim() == 3 else 0
    d.strideC = out.stride(0) if out.dim() == 3 else 0
    d.batch, d.M, d.N, d.K = batch, M, N, K
    d.bias_n, d.bias_m, d.scale_n = _p(_f32c(bias_n)), _p(_f32c(bias_m)), _p(_f32c(scale_n))
    if res is not None:
        if res.dtype != torch.float32 or res.stride(-1) != 1:
            raise ValueError("gemm: res must be float32, contiguous in its last dim")
        d.res, d.ldr = res.data_ptr(), res.stride(-2)
        d.strideR = res.stride(0) if res.dim() == 3 else 0
    d.act, d.out_f32, d.dtype = act, int(out_f32), _dt(a.dtype)
    d.stats = _p(stats)
    if aux is not None:
        if aux.dtype != a.dtype or aux.shape[-2:] != (M, N) or aux.stride(-1) != 1:
            raise ValueError("gemm: aux must be a 16-bit [M, N] matrix of the operand dtype")
        d.aux, d.ld_aux = aux.data_ptr(), aux.stride(-2)
    flops = 2.0 * batch * M * N * K
    if (a_lo is None) != (b_lo is None) and not (K % 64 == 0 and M >= 256 and N >= 32 and N % 4 == 0 and out.stride(-2) % 4 == 0):
        a_lo = b_lo = None   # one-sided (weight) split is an optional refinement: shapes off the large-tile path run plain
    if a_lo is not None:
        if a_lo.stride() != a.stride():
            raise ValueError("gemm: split halves must share the layout of their hi parts")
        d.A_lo = a_lo.data_ptr()
    if b_lo is not None:
        if b_lo.stride() != b.stride():
            raise ValueError("gemm: split halves must share the layout of their hi parts")
        d.B_lo = b_lo.data_ptr()
    if mx is not None:
        if a_lo is None or b_lo is None:
            raise ValueError("gemm: mx needs both MX planes (a_lo and b_lo)")
        _dev(mx[0], mx[1])
        d.mx_amax_a, d.mx_amax_b = mx[0].data_ptr(), mx[1].data_ptr()
    if out_lo is not None:
        if out_lo.dtype != out.dtype or out_lo.stride() != out.stride() or out.dtype == torch.float32:
            raise ValueError("gemm: out_lo must be a 16-bit tensor with the layout of out")
        d.C_lo = out_lo.data_ptr()
    if rowstats is not None:
        if rowstats.dtype != torch.float32 or rowstats.numel() != M * ((N + 63) // 64) * 2 or not rowstats.is_contiguous():
            raise ValueError("gemm: rowstats must be a contiguous float32 [M, ceil(N / 64), 2] tensor")
        d.rowstats = rowstats.data_ptr()
    if res16 is not None:
        rh, rl = res16
        _dev(rh, rl)
        if rh.dtype != a.dtype or rl.dtype != a.dtype or rh.stride() != rl.stride() or rh.stride(-1) != 1 or rh.shape[-2:] != (M, N):
            raise ValueError("gemm: res16 = (hi, lo) 16-bit [M, N] planes of the operand dtype with equal strides")
        d.res16, d.res16_lo, d.ldr16 = rh.data_ptr(), rl.data_ptr(), rh.stride(-2)
    if ln is not None:
        mr, cs, cols = ln
        _dev(mr, cs)
        d.ln_mr, d.ln_cs, d.ln_cols = _f32c(mr).data_ptr(), _f32c(cs).data_ptr(), int(bool(cols))
    nbytes = batch * (2.0 * (M * K + N * K) + M * N * ((4 if (out_f32 or out_lo is not None) else 2) +
                                                    (4 if (res is not None or res16 is not None) else 0)))
    return d, out, flops, nbytes


def gemm(a: torch.Tensor, b: torch.Tensor, **kw) -> torch.Tensor:
    """``out = epilogue(a @ b.T)``; a [M,K] or [batch,M,K], b [N,K] or [batch,N,K] (16-bit, K contiguous).
    Keywords: out, out_f32, bias_n, bias_m, scale_n, res, act, stats, a_lo, b_lo, aux (see ``_gemm_desc``).
    ``act=ACT_GELU_GRAD`` with ``aux`` (16-bit [M, N] pre-activation): out = (a @ b.T) * gelu'(aux).

    A 2-D operand next to a 3-D one is shared by every batch element.  Row strides may exceed K.
    """
    d, out, flops, nbytes = _gemm_desc(a, b, **kw)
    check(_launch_timed("gemm", flops, lambda: lib().asis_gemm(_stream(), C.byref(d)), nbytes), "asis_gemm")
    return out


def gemm_tiles_m(M: int) -> int:
    return lib().asis_gemm_tiles_m(int(M))


def gemm_set_option(name: str, value: int) -> None:
    """Run-time dispatch switch of ``asis_gemm`` (``"p8"``: the persistent 8-phase form, include/asis_hip.h)."""
    check(lib().asis_gemm_set_option(name.encode(), int(value)), "asis_gemm_set_option")


def conv_gemm(x_nhwc: torch.Tensor, w_packed: torch.Tensor, KH: int, KW: int, stride: int, pad: int, *,
              out: Optional[torch.Tensor] = None, out_f32: bool = True, bias_n: Optional[torch.Tensor] = None,
              act: int = ACT_NONE, stats: Optional[torch.Tensor] = None, accumulate: bool = False,
              x_lo: Optional[torch.Tensor] = None, w_lo: Optional[torch.Tensor] = None, ksplit: int = 1, mx=None) -> torch.Tensor:
    """Implicit-GEMM convolution: x [B,H,W,Cin] (16-bit NHWC), w_packed [Cout, KH*KW*Cin] ->
    out [B,OH,OW,Cout] (fp32 by default: BatchNorm statistics are taken on it).
    ``ksplit`` > 1: the reduction runs as that many side-by-side parts (fp32 partial maps summed in a fixed order, then the
    caller takes the
    BatchNorm statistics from the sum with ``colstats``) — for layers whose tile count fills only part of the machine."""
    _dev(x_nhwc, w_packed, out, bias_n, stats)
    if not x_nhwc.is_contiguous():
        raise ValueError("conv_gemm: x must be contiguous NHWC")
    Bn, H, W, Cin = x_nhwc.shape
    Cout, K = w_packed.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    if ksplit > 1:
        if out is not None or accumulate or not out_f32 or act != ACT_NONE or stats is not None:
            raise ValueError("conv_gemm: ksplit needs a fresh fp32 output without activation; take the BatchNorm "
                             "statistics from the sum with colstats()")
        parts = torch.empty((ksplit, Bn, OH, OW, Cout), device=x_nhwc.device, dtype=torch.float32)
        out = parts
    if out is None:
        out = torch.empty((Bn, OH, OW, Cout), device=x_nhwc.device, dtype=torch.float32 if out_f32 else x_nhwc.dtype)
    d = GemmDesc()
    d.A, d.B, d.C = x_nhwc.data_ptr(), w_packed.data_ptr(), out.data_ptr()
    if ksplit > 1:
        d.ksplit, d.strideC = ksplit, Bn * OH * OW * Cout
    d.lda, d.ldb, d.ldc = K, w_packed.stride(0), Cout
    d.batch, d.M, d.N, d.K = 1, Bn * OH * OW, Cout, K
    d.bias_n = _p(_f32c(bias_n))
    if accumulate:  # out += conv(x, w): the epilogue adds the previous fp32 contents (same element, same thread)
        if out.dtype != torch.float32:
            raise ValueError("conv_gemm: accumulate needs an fp32 output")
        d.res, d.ldr = out.data_ptr(), Cout
    d.act, d.out_f32, d.dtype = act, int(out.dtype == torch.float32), _dt(x_nhwc.dtype)
    d.conv, d.B_, d.H, d.W, d.Cin, d.OH, d.OW = 1, Bn, H, W, Cin, OH, OW
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.stats = _p(stats)
    flops = 2.0 * d.M * Cout * K
    if x_lo is not None:
        d.A_lo, d.B_lo = x_lo.data_ptr(), w_lo.data_ptr()
    if mx is not None:     # x_lo / w_lo are in the MX form; mx = (amax of x, amax of w) device floats
        d.mx_amax_a, d.mx_amax_b = _f32c(mx[0]).data_ptr(), _f32c(mx[1]).data_ptr()
    check(_launch_timed("conv", flops, lambda: lib().asis_gemm(_stream(), C.byref(d))), "asis_gemm(conv)")
    if ksplit > 1:
        out = torch.empty((Bn, OH, OW, Cout), device=x_nhwc.device, dtype=torch.float32)
        reduce_rows(parts.view(ksplit, -1), 1.0, out.view(-1))
    return out


_FUSED_SPLIT = os.environ.get("ASIS_GEMM_BIG", "1") != "0" and os.environ.get("ASIS_SPLIT_FUSED", "1") != "0"
# narrow MX convolutions (64 / 128 output channels) on the halo-tile kernel (csrc/convhalo.hip): 256 -> 128 at 168^2 544 vs 577 us,
# 128 -> 64 at 336^2 540 vs 692 us against the implicit-GEMM form (scripts/bench_conv_halo.py, profiles/r04_conv_halo_ab.txt);
# ASIS_CONV_HALO=0: implicit GEMM
CONV_HALO = os.environ.get("ASIS_CONV_HALO", "1") not in ("0", "")


def conv_gemm_split(x_hi, x_lo, w_hi, w_lo, KH: int, KW: int, stride: int, pad: int, *, bias_n=None, stats=None,
                    ksplit: int = 1, mx=None):
    """Split-precision convolution: x ~= x_hi + x_lo, w ~= w_hi + w_lo (16-bit halves), fp32 out =
    x_hi*w_hi + x_lo*w_hi + x_hi*w_lo (+ bias), accumulated in fp32 (the dropped x_lo*w_lo term is ~2^-22
    relative): one launch over a virtual 3K reduction on the large-tile kernel, else three accumulate passes."""
    if mx is None:
        _not_mx("conv_gemm_split", x_lo, w_lo)
    Bn, H, W, Cin = x_hi.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    Cout = w_hi.shape[0]
    if Cin % 64 == 0 and Bn * OH * OW >= 256 and Cout >= 32 and Cout % 4 == 0 and _FUSED_SPLIT:
        return conv_gemm(x_hi, w_hi, KH, KW, stride, pad, bias_n=bias_n, stats=stats, x_lo=x_lo, w_lo=w_lo, ksplit=ksplit, mx=mx)
    if mx is not None:
        raise ValueError("conv_gemm_split: MX operands need the fused large-tile path (mx_conv_ok)")
    out = conv_gemm(x_hi, w_hi, KH, KW, stride, pad, bias_n=bias_n)
    conv_gemm(x_lo, w_hi, KH, KW, stride, pad, out=out, accumulate=True)
    conv_gemm(x_hi, w_lo, KH, KW, stride, pad, out=out, accumulate=True, stats=stats)
    return out


def conv_halo_ok(x_hi: torch.Tensor, Cout: int, stride: int, pad: int, force: bool = False) -> bool:
    """shapes ``conv3x3_halo_mx`` covers: float16, stride 1, pad 1, Cin % 64 == 0, Cout 64 / 128 (csrc/convhalo.hip); ``force``:
    whatever the ASIS_CONV_HALO switch says"""
    return ((CONV_HALO or force) and x_hi.dtype == torch.float16 and stride == 1 and pad == 1 and x_hi.shape[3] % 64 == 0 and Cout in (64, 128)
            and x_hi.is_contiguous())


def conv3x3_halo_mx(x_hi, x_mx, w_hi, w_mx, mx, *, bias_n=None, want_stats: bool = False):
    """3x3 / stride 1 / pad 1 convolution of the MX operand planes on the halo-tile kernel (include/asis_hip.h:
    asis_conv3x3_halo_mx) -> fp32 NHWC out, or (out, stats [tiles, 2, Cout]) with ``want_stats``."""
    _dev(x_hi, x_mx, w_hi, w_mx, bias_n, mx[0], mx[1])
    Bn, H, W, Cin = x_hi.shape
    Cout = w_hi.shape[0]
    if not (x_hi.is_contiguous() and x_mx.is_contiguous() and w_hi.is_contiguous() and w_mx.is_contiguous() and w_hi.shape[1] == 9 * Cin):
        raise ValueError("conv3x3_halo_mx: contiguous NHWC planes and [Cout, 9 Cin] packed weights expected")
    out = torch.empty((Bn, H, W, Cout), device=x_hi.device, dtype=torch.float32)
    stats = None
    if want_stats:
        stats = torch.empty((lib().asis_conv3x3_halo_mx_tiles(Bn, H, W), 2, Cout), device=x_hi.device, dtype=torch.float32)
    check(lib().asis_conv3x3_halo_mx(_stream(), _dt(x_hi.dtype), x_hi.data_ptr(), x_mx.data_ptr(), w_hi.data_ptr(), w_mx.data_ptr(),
                                     _p(_f32c(bias_n)), mx[0].data_ptr(), mx[1].data_ptr(), out.data_ptr(), _p(stats), Bn, H, W, Cin, Cout),
          "asis_conv3x3_halo_mx")
    return (out, stats) if want_stats else out


def gemm_split(a_hi, a_lo, b_hi, b_lo, *, out: torch.Tensor, bias_n=None):
    """Split-precision ``out = (a_hi + a_lo) @ (b_hi + b_lo).T + bias`` into an fp32 ``out``."""
    M, K, N = a_hi.shape[-2], a_hi.shape[-1], b_hi.shape[-2]
    if K % 64 == 0 and M >= 256 and N >= 32 and N % 4 == 0 and _FUSED_SPLIT:
        return gemm(a_hi, b_hi, out=out, bias_n=bias_n, a_lo=a_lo, b_lo=b_lo)
    gemm(a_hi, b_hi, out=out, bias_n=bias_n)
    gemm(a_lo, b_hi, out=out, res=out)
    gemm(a_hi, b_lo, out=out, res=out)
    return out


def ln_fold_supported(M: int, N: int, K: int, batch: int = 1, producer: bool = False) -> bool:
    """True when ``asis_gemm`` dispatches this dense shape to a kernel that implements the LayerNorm-fold epilogue fields:
    consumers (``ln=``) the persistent 8-phase kernel or the one-tile-per-workgroup 8-phase form, producers (``out_lo`` /
    ``rowstats`` / ``res16``) the latter only.  Mirrors csrc/gemm.hip:launch."""
    if M < 256 or N < 256 or K % 64 or K < 128 or N % 8:
        return False
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if not producer and batch == 1 and tiles >= 256 and K <= 2048:
        return True
    return K >= 2048 or (K >= 1024 and tiles * batch >= 128)


def ln_stats_finalize(rowstats: torch.Tensor, D: int, eps: float = 1e-6) -> torch.Tensor:
    """per-row partial sums of a GEMM epilogue (``gemm(..., rowstats=)``) -> mr fp32 [rows, 2] = (mean, rstd) of nn.LayerNorm
    (8 spare rows behind it: a V^T GEMM with a rounded-up token count reads a few statistics past the last row)"""
    _dev(rowstats)
    rows, groups = rowstats.shape[0], rowstats.shape[1]
    mr = torch.empty((rows + 8, 2), device=rowstats.device, dtype=torch.float32)[:rows]
    check(lib().asis_ln_stats_finalize(_stream(), _f32c(rowstats).data_ptr(), rows, groups, D, float(eps), mr.data_ptr()),
          "asis_ln_stats_finalize")
    return mr


def split_stats(x: torch.Tensor, dtype: torch.dtype, eps: float = 1e-6):
    """fp32 [rows, D] -> (hi, lo 16-bit planes, mr fp32 [rows, 2]): entry into the folded-LayerNorm chain"""
    _dev(x)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("split_stats: x must be float32 [rows, D] contiguous in D")
    rows, D = x.shape
    hi = torch.empty((rows + 8, D), device=x.device, dtype=dtype)[:rows]     # spare rows: Attention.attend_rows reads past the end
    lo = torch.empty((rows, D), device=x.device, dtype=dtype)
    mr = torch.empty((rows + 8, 2), device=x.device, dtype=torch.float32)[:rows]
    check(lib().asis_split_stats(_stream(), _dt(dtype), x.data_ptr(), x.stride(0), hi.data_ptr(), lo.data_ptr(), D, mr.data_ptr(),
                                 rows, D, float(eps)), "asis_split_stats")
    return hi, lo, mr


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6,
              out_dtype: torch.dtype = T16_DEFAULT, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row LayerNorm of a float32 [..., D] tensor -> ``out_dtype`` (16-bit GEMM operand or float32)."""
    _dev(x, w, b, out)
    if x.dtype != torch.float32 or x.stride(-1) != 1:
        raise ValueError("layernorm: x must be float32 with a contiguous last dim")
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    o2 = out.view(-1, D)
    f32 = out.dtype == torch.float32
    check(lib().asis_layernorm(_stream(), _dt(out.dtype) if not f32 else 0, x2.data_ptr(), x2.stride(0),
                               _f32c(w).data_ptr(), _f32c(b).data_ptr(), float(eps), o2.data_ptr(), o2.stride(0),
                               int(f32), x2.shape[0], D), "asis_layernorm")
    return out


def layernorm_mx(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, out_dtype: torch.dtype, amax: torch.Tensor):
    """Row LayerNorm of a float32 [rows, D] tensor -> (16-bit output, its MX plane) in one pass; ``amax``: device float, an upper
    bound of |output| (include/asis_hip.h: asis_layernorm_mx)."""
    _dev(x, w, b, amax)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("layernorm_mx: x must be float32 [rows, D] with contiguous columns")
    R, D = x.shape
    hi = torch.empty((R, D), device=x.device, dtype=out_dtype)
    mx = torch.empty_like(hi)
    check(lib().asis_layernorm_mx(_stream(), _dt(out_dtype), x.data_ptr(), x.stride(0), _f32c(w).data_ptr(), _f32c(b).data_ptr(), float(eps),
                                  hi.data_ptr(), mx.data_ptr(), D, amax.data_ptr(), R, D), "asis_layernorm_mx")
    return hi, mx


def _check_out_lo(out: torch.Tensor, out_lo: Optional[torch.Tensor]) -> None:
    if out_lo is not None and (out_lo.shape != out.shape or out_lo.stride() != out.stride() or out_lo.dtype != out.dtype):
        raise ValueError("attention_fwd: out_lo must share the shape, layout and dtype of out")


def _attention_launch(q, k, vt, out, out_lo, B1, N1, B2, N2, H, scale, lse):
    """``scale`` None: q already carries scale * log2(e) (folded into the projection weights) -> the folded kernel"""
    if scale is None:
        check(lib().asis_attention_fwd_prescaled(_stream(), _dt(q.dtype), q.data_ptr(), k.data_ptr(), q.stride(0), vt.data_ptr(),
                                                 vt.stride(1), out.data_ptr(), _p(out_lo), out.stride(0), B1, N1, B2, N2, H,
                                                 _p(lse)), "asis_attention_fwd_prescaled")
    else:
        check(lib().asis_attention_fwd_split(_stream(), _dt(q.dtype), q.data_ptr(), k.data_ptr(), q.stride(0), vt.data_ptr(),
                                             vt.stride(1), out.data_ptr(), _p(out_lo), out.stride(0), B1, N1, B2, N2, H,
                                             float(scale), _p(lse)), "asis_attention_fwd")


def attention_fwd_qkv(qkv: torch.Tensor, segs, H: int, scale: Optional[float], out: torch.Tensor,
                      out_lo: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
                      mx_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv: 16-bit [rows, 3*H*64] = q | k | v of ONE projection GEMM (V row-major: no transposed copy); ``segs`` one or two
    (B, N) token batches stacked along the rows; ``scale`` None = q carries scale * log2(e) (folded projection).
    ``mx_amax`` (device float >= max |v|, e.g. ``absmax16`` of the v columns): ``out_lo`` receives the MX form of the output's lo
    half instead of the 16-bit residual (include/asis_hip.h: asis_attention_fwd_qkv_mx)."""
    _dev(qkv, out, out_lo, lse, mx_amax)
    D = H * 64
    if qkv.dim() != 2 or qkv.shape[1] < 3 * D or qkv.stride(1) != 1 or len(segs) not in (1, 2):
        raise ValueError("attention_fwd_qkv: qkv must be [rows, >= 3*H*64] with contiguous rows, one or two segments")
    (B1, N1), (B2, N2) = (segs[0], segs[1]) if len(segs) == 2 else (segs[0], (0, 0))
    if qkv.shape[0] != B1 * N1 + B2 * N2:
        raise ValueError("attention_fwd_qkv: segments do not cover the rows")
    _check_out_lo(out, out_lo)
    es = qkv.element_size()
    if mx_amax is not None:
        if out_lo is None:
            raise ValueError("attention_fwd_qkv: mx_amax needs the second output plane")
        check(lib().asis_attention_fwd_qkv_mx(_stream(), _dt(qkv.dtype), qkv.data_ptr(), qkv.data_ptr() + D * es,
                                              qkv.data_ptr() + 2 * D * es, qkv.stride(0), out.data_ptr(), out_lo.data_ptr(), out.stride(0),
                                              B1, N1, B2, N2, H, 0.0 if scale is None else float(scale), int(scale is None), _p(lse),
                                              mx_amax.data_ptr()), "asis_attention_fwd_qkv_mx")
        return out
    check(lib().asis_attention_fwd_qkv(_stream(), _dt(qkv.dtype), qkv.data_ptr(), qkv.data_ptr() + D * es,
                                       qkv.data_ptr() + 2 * D * es, qkv.stride(0), out.data_ptr(), _p(out_lo), out.stride(0),
                                       B1, N1, B2, N2, H, 0.0 if scale is None else float(scale), int(scale is None), _p(lse)),
          "asis_attention_fwd_qkv")
    return out


def attention_fwd(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, B: int, H: int, N: int, scale: Optional[float],
                  out: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
                  out_lo: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q, k: [B*N, >=H*64] views (same row stride); vt: [B, H*64, ldvt]; returns o [B*N, H*64].
    ``scale`` None = q was produced with scale * log2(e) folded into its projection (``Attention._qkv_folded``).
    ``lse``: optional fp32 [B,H,N] receiving the per-query log2-sum-exp (training forward).
    ``out_lo``: optional tensor laid out like ``out`` receiving the rounding residual of the 16-bit output (o ~= out + out_lo:
    the projection GEMM's split A operand, config.split_attn_out)."""
    _dev(q, k, vt, out, lse, out_lo)
    if q.stride(0) != k.stride(0) or q.stride(1) != 1 or k.stride(1) != 1:
        raise ValueError("attention_fwd: q and k must share a row stride and be contiguous in the last dim")
    if out is None:
        out = torch.empty((B * N, H * 64), device=q.device, dtype=q.dtype)
    if lse is not None and (lse.dtype != torch.float32 or lse.numel() != B * H * N or not lse.is_contiguous()):
        raise ValueError("attention_fwd: lse must be contiguous float32 [B,H,N]")
    _check_out_lo(out, out_lo)
    _attention_launch(q, k, vt, out, out_lo, B, N, 0, 0, H, scale, lse)
    return out


def attention_fwd_seg(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, B1: int, N1: int, B2: int, N2: int, H: int,
                      scale: Optional[float], out: torch.Tensor, out_lo: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Two stacked token batches in one launch: q, k [B1*N1 + B2*N2, >=H*64] row views, vt [B1+B2, H*64, ldvt]."""
    _dev(q, k, vt, out, out_lo)
    _check_out_lo(out, out_lo)
    if q.stride(0) != k.stride(0) or q.stride(1) != 1 or k.stride(1) != 1 or q.shape[0] != B1 * N1 + B2 * N2:
        raise ValueError("attention_fwd_seg: q and k must be row views over both batches with one row stride")
    _attention_launch(q, k, vt, out, out_lo, B1, N1, B2, N2, H, scale, None)
    return out


def token_ld(N: int) -> int:
    """row stride of the token-contiguous (transposed) attention operands: N rounded up to 64"""
    return (N + 63) // 64 * 64


def transpose_tokens(src: torch.Tensor, B: int, N: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """16-bit [B*N, C] view (row stride free) -> [B, C, token_ld(N)], tokens contiguous, zero padded."""
    _dev(src, out)
    Cc = src.shape[1]
    if src.stride(1) != 1 or src.shape[0] != B * N:
        raise ValueError("transpose_tokens: expected a [B*N, C] row view")
    ldt = token_ld(N)
    if out is None:
        out = torch.empty((B, Cc, ldt), device=src.device, dtype=src.dtype)
    check(lib().asis_transpose_tokens(_stream(), _dt(src.dtype), src.data_ptr(), src.stride(0), out.data_ptr(), ldt, B, N,
                                      Cc), "asis_transpose_tokens")
    return out


def attention_bwd_rows(q, k, v, o, dO, lse, segs, H: int, scale: float, dqkv: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> dqkv 16-bit [R, 3*H*64] = [dQ | dK | dV] for one or two stacked token batches ``segs`` = [(B1, N1)] or
    [(B1, N1), (B2, N2)] (rows of batch 2 behind batch 1 in every operand).  q, k, v: [R, H*64] views with one row stride
    (the three column blocks of the qkv projection); o, dO: [R, H*64]; lse: fp32, the forward's log-sum-exp of each batch
    ([B1, H, N1] then [B2, H, N2]) in one contiguous buffer.  No transposed operands (csrc/attn_bwd_pipe.hip)."""
    _dev(q, k, v, o, dO, lse, dqkv)
    Wd = H * 64
    if len(segs) not in (1, 2):
        raise ValueError("attention_bwd_rows: one or two stacked token batches")
    (B1, N1), (B2, N2) = segs[0], (segs[1] if len(segs) == 2 else (0, 0))
    R = B1 * N1 + B2 * N2
    if not (q.stride(0) == k.stride(0) == v.stride(0)) or q.stride(1) != 1 or q.shape[0] != R:
        raise ValueError("attention_bwd_rows: q, k, v must be [R, H*64] views sharing a row stride")
    if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != R * H:
        raise ValueError("attention_bwd_rows: lse must be one contiguous float32 buffer of R * H values")
    if dqkv is None:
        dqkv = torch.empty((R, 3 * Wd), device=q.device, dtype=q.dtype)
    D = torch.empty(R * H, device=q.device, dtype=torch.float32)
    es = dqkv.element_size()
    check(lib().asis_attention_bwd_rows(_stream(), _dt(q.dtype), q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0),
                                        o.data_ptr(), o.stride(0), dO.data_ptr(), dO.stride(0), lse.data_ptr(), D.data_ptr(),
                                        dqkv.data_ptr(), dqkv.data_ptr() + Wd * es, dqkv.data_ptr() + 2 * Wd * es,
                                        dqkv.stride(0), B1, N1, B2, N2, H, float(scale)), "asis_attention_bwd_rows")
    return dqkv


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6,
                  res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """fp32 [R, D] row views -> (dx = res + LN^T(dy), partial fp32 [nblk, 2, D] = (d weight, d bias) sums)."""
    _dev(dy, x, w, res, out)
    R, D = x.shape
    for t in (dy, x, res, out):
        if t is not None and (t.dtype != torch.float32 or t.stride(-1) != 1 or t.shape != x.shape):
            raise ValueError("layernorm_bwd: float32 [R, D] row views of one shape expected")
    if out is None:
        out = torch.empty((R, D), device=x.device, dtype=torch.float32)
    nblk = lib().asis_rowblock_nblk(R)
    partial = torch.empty((nblk, 2, D), device=x.device, dtype=torch.float32)
    check(lib().asis_layernorm_bwd(_stream(), dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), _f32c(w).data_ptr(),
                                   float(eps), _p(res), res.stride(0) if res is not None else 0, out.data_ptr(),
                                   out.stride(0), partial.data_ptr(), R, D), "asis_layernorm_bwd")
    return out, partial


def gelu16(pre: torch.Tensor, dpost: Optional[torch.Tensor] = None) -> torch.Tensor:
    """16-bit contiguous: gelu(pre), or with dpost the backward dpost * gelu'(pre)."""
    _dev(pre, dpost)
    if not pre.is_contiguous() or (dpost is not None and (not dpost.is_contiguous() or dpost.shape != pre.shape)):
        raise ValueError("gelu16: contiguous operands of one shape expected")
    out = torch.empty_like(pre)
    check(lib().asis_gelu16(_stream(), _dt(pre.dtype), pre.data_ptr(), _p(dpost), out.data_ptr(), pre.numel()), "asis_gelu16")
    return out


def gelu_split(x: torch.Tensor, dtype: torch.dtype, split: bool = True):
    """fp32 contiguous -> (16-bit gelu(x), its rounding residual | None): a split-precision operand pair."""
    _dev(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("gelu_split: contiguous float32 expected")
    hi = torch.empty(x.shape, device=x.device, dtype=dtype)
    lo = torch.empty_like(hi) if split else None
    check(lib().asis_gelu_split(_stream(), _dt(dtype), x.data_ptr(), hi.data_ptr(), _p(lo), x.numel()), "asis_gelu_split")
    return hi, lo


def swiglu_bwd(x12: torch.Tensor, dh: torch.Tensor) -> torch.Tensor:
    """x12 fp32 [R, 2*Hd], dh 16-bit [R, Hd] -> 16-bit [R, 2*Hd] = [d x1 | d x2]."""
    _dev(x12, dh)
    R, H2 = x12.shape
    if dh.shape != (R, H2 // 2) or not dh.is_contiguous():
        raise ValueError("swiglu_bwd: shape mismatch")
    out = torch.empty((R, H2), device=x12.device, dtype=dh.dtype)
    check(lib().asis_swiglu_bwd(_stream(), _dt(dh.dtype), _f32c(x12).data_ptr(), dh.data_ptr(), out.data_ptr(), R, H2 // 2),
          "asis_swiglu_bwd")
    return out


def colsum(x: torch.Tensor) -> torch.Tensor:
    """[R, C] row view (16-bit or float32) -> partial fp32 [nblk, C] column sums (finish with reduce_rows)."""
    _dev(x)
    R, Cc = x.shape
    if x.stride(1) != 1:
        raise ValueError("colsum: rows must be contiguous")
    dt = _lib.ASIS_F32 if x.dtype == torch.float32 else _dt(x.dtype)
    partial = torch.empty((lib().asis_rowblock_nblk(R), Cc), device=x.device, dtype=torch.float32)
    check(lib().asis_colsum(_stream(), dt, x.data_ptr(), x.stride(0), partial.data_ptr(), R, Cc), "asis_colsum")
    return partial


def cast_colsum(x: torch.Tensor, dtype: torch.dtype, scale: float = 1.0):
    """fp32 [R, D] row view -> (16-bit [R, D], partial fp32 [nblk, D] column sums of x) in one pass."""
    _dev(x)
    R, D = x.shape
    if x.dtype != torch.float32 or x.stride(1) != 1:
        raise ValueError("cast_colsum: float32 rows expected")
    out = torch.empty((R, D), device=x.device, dtype=dtype)
    partial = torch.empty((lib().asis_rowblock_nblk(R), D), device=x.device, dtype=torch.float32)
    check(lib().asis_cast_colsum(_stream(), _dt(dtype), x.data_ptr(), x.stride(0), out.data_ptr(), D, float(scale),
                                 partial.data_ptr(), R, D), "asis_cast_colsum")
    return out, partial


def ls_linear_finish(G: torch.Tensor, W: Optional[torch.Tensor], bias, gamma, cs, grad_scale: float, dW: torch.Tensor,
                     db: Optional[torch.Tensor], dgamma: Optional[torch.Tensor]) -> None:
    """See include/asis_hip.h: (G, cs) -> dW, db, dgamma of ``x + gamma * (A W^T + b)`` (gamma None: plain Linear)."""
    _dev(G, W, bias, gamma, cs, dW, db, dgamma)
    N, K = G.shape
    check(lib().asis_ls_linear_finish(_stream(), _f32c(G).data_ptr(), _p(_f32c(W)), _p(_f32c(bias)), _p(_f32c(gamma)),
                                      _p(_f32c(cs)), float(grad_scale), _f32c(dW).data_ptr(), _p(db), _p(dgamma), N, K),
          "asis_ls_linear_finish")


def im2col_patch(img: torch.Tensor, P: int, ldk: int, dtype: torch.dtype = T16_DEFAULT, split: bool = False):
    """fp32 [B,3,H,W] -> 16-bit [B*(H/P)*(W/P), ldk] (k = c*P*P + i*P + j, pad columns zero); ``split``: -> (hi, lo)."""
    _dev(img)
    if img.dtype != torch.float32 or not img.is_contiguous() or img.dim() != 4 or img.shape[1] != 3:
        raise ValueError("im2col_patch: img must be contiguous float32 [B,3,H,W]")
    B, _, Hi, Wi = img.shape
    if Hi % P or Wi % P:
        # error text mirrors dinov2/layers/patch_embed.py:72-73
        raise AssertionError(f"Input image height {Hi} / width {Wi} is not a multiple of patch size {P}")
    out = torch.empty((B * (Hi // P) * (Wi // P), ldk), device=img.device, dtype=dtype)
    lo = torch.empty_like(out) if split else None
    check(lib().asis_im2col_patch_split(_stream(), _dt(dtype), img.data_ptr(), B, Hi, Wi, P, out.data_ptr(), _p(lo), ldk),
          "asis_im2col_patch")
    return (out, lo) if split else out


def cast_pad(src: torch.Tensor, ld_dst: Optional[int] = None, dtype: torch.dtype = T16_DEFAULT,
             scale: float = 1.0, part: int = 0) -> torch.Tensor:
    """float32 [rows, cols] -> 16-bit [rows, ld_dst] with zero-filled pad columns (weight packing)."""
    _dev(src)
    if src.dtype != torch.float32 or src.dim() != 2 or src.stride(1) != 1:
        raise ValueError("cast_pad: src must be float32 [rows, cols] contiguous in cols")
    rows, cols = src.shape
    ld = ld_dst if ld_dst is not None else (cols + 7) // 8 * 8
    out = torch.empty((rows, ld), device=src.device, dtype=dtype)
    check(lib().asis_cast_pad(_stream(), _dt(dtype), src.data_ptr(), src.stride(0), out.data_ptr(), ld, rows, cols,
                              float(scale), int(part)),
          "asis_cast_pad")
    return out


def add_cls_pos(x: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B,N,D] f32, cls [D], pos [N+1,D] -> [B,N+1,D] (vision_transformer.py:196-197)."""
    _dev(x, cls, pos, out)
    B, N, D = x.shape
    if out is None:
        out = torch.empty((B, N + 1, D), device=x.device, dtype=torch.float32)
    elif out.dtype != torch.float32 or tuple(out.shape) != (B, N + 1, D) or not out.is_contiguous():
        raise ValueError("add_cls_pos: out must be a contiguous float32 [B, N+1, D] tensor")
    check(lib().asis_add_cls_pos(_stream(), _f32c(x).data_ptr(), _f32c(cls).data_ptr(), _f32c(pos).data_ptr(),
                                 out.data_ptr(), B, N, D), "asis_add_cls_pos")
    return out


# --------------------------------------------------------------------------------------------
# adapters
# --------------------------------------------------------------------------------------------
def msda_fwd(value: torch.Tensor, offaw: torch.Tensor, ref: torch.Tensor, shapes_i32: torch.Tensor,
             starts_i32: torch.Tensor, B: int, Lq: int, M: int, L: int, P: int, split: bool = False):
    """value 16-bit [B, Lin, D]; offaw fp32 [B*Lq, >= M*L*P*3]; ref fp32 [Lq, 2] -> out 16-bit [B*Lq, D]; ``split``: -> (out,
    out_lo) with out_lo the rounding residuals (a split-precision operand pair for output_proj)."""
    _dev(value, offaw, ref, shapes_i32, starts_i32)
    Lin, D = value.shape[1], value.shape[2]
    if not value.is_contiguous() or offaw.stride(1) != 1:
        raise ValueError("msda_fwd: value must be contiguous and offaw contiguous in its last dim")
    out = torch.empty((B * Lq, D), device=value.device, dtype=value.dtype)
    lo = torch.empty_like(out) if split else None
    check(lib().asis_msda_fwd_split(_stream(), _dt(value.dtype), value.data_ptr(), offaw.data_ptr(), offaw.stride(0),
                                    _f32c(ref).data_ptr(), shapes_i32.data_ptr(), starts_i32.data_ptr(), out.data_ptr(), _p(lo), B,
                                    Lq, Lin, M, L, P, D // M), "asis_msda_fwd")
    return (out, lo) if split else out


def dwconv_gelu(x: torch.Tensor, w9: torch.Tensor, bias: torch.Tensor, shapes_i32: torch.Tensor,
                starts_i32: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """x fp32 [B, Ntok, C] -> gelu(dwconv3x3(x)) 16-bit [B, Ntok, C] over the L token grids."""
    _dev(x, w9, bias, shapes_i32, starts_i32)
    B, Ntok, Cc = x.shape
    out = torch.empty((B, Ntok, Cc), device=x.device, dtype=dtype)
    check(lib().asis_dwconv_gelu(_stream(), _dt(dtype), _f32c(x).data_ptr(), _f32c(w9).data_ptr(), _f32c(bias).data_ptr(),
                                 shapes_i32.data_ptr(), starts_i32.data_ptr(), shapes_i32.shape[0], out.data_ptr(), B,
                                 Ntok, Cc), "asis_dwconv_gelu")
    return out


# --------------------------------------------------------------------------------------------
# CNN encoder / decoder companions
# --------------------------------------------------------------------------------------------
_ST_CACHE = {}
# d value of MSDeformAttn as a gather over taps bucketed by destination pixel (round 4); ASIS_MSDA_SORTED=0: the dense sampling
# matrix + batched GEMMs of round 2 (a 2.4 GB matrix at the headline batch)
MSDA_SORTED = os.environ.get("ASIS_MSDA_SORTED", "1") not in ("0", "")


def msda_bwd(value: torch.Tensor, offaw: torch.Tensor, ref: torch.Tensor, shapes_i32: torch.Tensor, starts_i32: torch.Tensor,
             dout: torch.Tensor, B: int, Lq: int, M: int, L: int, P: int, dense: bool = True):
    """value 16-bit [B, Lin, D]; offaw fp32 [B*Lq, M*L*P*3]; dout fp32 [B*Lq, D] ->
    (dvalue fp32 [B, Lin, D], doffaw fp32 like offaw).
    ``dense`` (default): d value through the dense sampling matrix and batched MFMA GEMMs (deterministic);
    otherwise the scatter kernels (LDS tile / global atomics)."""
    _dev(value, offaw, ref, shapes_i32, starts_i32, dout)
    _, Lin, D = value.shape
    Dh = D // M
    doffaw = torch.empty_like(offaw)
    use_dense = dense and Dh % 8 == 0 and D % 64 == 0   # token transpose works on 64-column tiles
    dvalue = torch.empty((B, Lin, D), device=value.device, dtype=torch.float32) if use_dense else \
        zeros((B, Lin, D), value.device)
    check(lib().asis_msda_bwd(_stream(), _dt(value.dtype), value.data_ptr(), _f32c(offaw).data_ptr(), offaw.stride(0),
                              _f32c(ref).data_ptr(), shapes_i32.data_ptr(), starts_i32.data_ptr(), _f32c(dout).data_ptr(),
                              None if use_dense else dvalue.data_ptr(), doffaw.data_ptr(), B, Lq, Lin, M, L, P, Dh),
          "asis_msda_bwd")
    if use_dense and MSDA_SORTED:
        # taps bucketed by destination pixel + one wave per (image, head, pixel) (csrc/adapter_bwd.hip): no dense matrix
        dt = value.dtype
        d16 = cast_pad(dout, D, dt)
        amax = absmax_f32(dout)
        cap = lib().asis_msda_vgrad_cap(Lq, L, P)
        key = ("sorted", B, M, Lin, cap, value.device)
        ws = _ST_CACHE.get(key)
        if ws is None:
            for k in [k for k in _ST_CACHE if k[0] != "sorted"]:
                del _ST_CACHE[k]
            ws = (torch.empty(B * M * Lin, device=value.device, dtype=torch.int32),
                  torch.empty(B * M * (Lin + 1), device=value.device, dtype=torch.int32),
                  torch.empty((B * M * cap, 2), device=value.device, dtype=torch.int32))
            _ST_CACHE[key] = ws
        check(lib().asis_msda_value_grad(_stream(), _dt(dt), offaw.data_ptr(), offaw.stride(0), ref.data_ptr(), shapes_i32.data_ptr(),
                                         starts_i32.data_ptr(), d16.data_ptr(), amax.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                                         ws[2].data_ptr(), dvalue.data_ptr(), B, Lq, Lin, M, L, P, Dh), "asis_msda_value_grad")
    elif use_dense:
        dt = value.dtype
        ldt = token_ld(Lq)
        key = (B, M, Lin, ldt, dt, value.device)
        ST = _ST_CACHE.get(key)
        if ST is None:
            _ST_CACHE.clear()          # one resident sampling matrix (2.4 GB at B = 12, 588^2): the call shapes alternate
            ST = torch.empty((B * M, Lin, ldt), device=value.device, dtype=dt)
            _ST_CACHE[key] = ST
        check(lib().asis_zero(_stream(), ST.data_ptr(), ST.numel() * ST.element_size()), "asis_zero")
        check(lib().asis_msda_sampling_matrix(_stream(), _dt(dt), offaw.data_ptr(), offaw.stride(0), ref.data_ptr(),
                                              shapes_i32.data_ptr(), starts_i32.data_ptr(), ST.data_ptr(), ldt, B, Lq, Lin, M, L,
                                              P), "asis_msda_sampling_matrix")
        doutT = transpose_tokens(cast_pad(dout, D, dt), B, Lq)                     # [B, D, ldt]
        st4 = ST.view(B, M, Lin, ldt)
        for m in range(M):   # out[b, pix, m*Dh + d] = sum_q ST[b, m, pix, q] * doutT[b, m*Dh + d, q]
            gemm(st4[:, m], doutT[:, m * Dh:(m + 1) * Dh], out=dvalue[:, :, m * Dh:(m + 1) * Dh], out_f32=True)
    return dvalue, doffaw


def dwconv_gelu_bwd(x: torch.Tensor, w9: torch.Tensor, bias: torch.Tensor, shapes_i32: torch.Tensor, starts_i32: torch.Tensor,
                    dy: torch.Tensor, dtype: torch.dtype):
    """x, dy fp32 [B, Ntok, C] -> (dx 16-bit [B, Ntok, C], partial fp32 [nblk, 10, C] = d w9 (9 rows) and d bias)."""
    _dev(x, w9, bias, shapes_i32, starts_i32, dy)
    B, Ntok, Cc = x.shape
    g = torch.empty_like(x)
    partial = torch.empty((lib().asis_dwconv_bwd_nblk(B * Ntok), 10, Cc), device=x.device, dtype=torch.float32)
    dx = torch.empty((B, Ntok, Cc), device=x.device, dtype=dtype)
    check(lib().asis_dwconv_gelu_bwd(_stream(), _dt(dtype), _f32c(x).data_ptr(), _f32c(w9).data_ptr(), _f32c(bias).data_ptr(),
                                     shapes_i32.data_ptr(), starts_i32.data_ptr(), shapes_i32.shape[0], _f32c(dy).data_ptr(),
                                     g.data_ptr(), partial.data_ptr(), dx.data_ptr(), B, Ntok, Cc), "asis_dwconv_gelu_bwd")
    return dx, partial


def conv3x3_c3(img: torch.Tensor, w: torch.Tensor, stride: int, pad: int) -> torch.Tensor:
    _dev(img, w)
    B, _, H, W = img.shape
    Cout = w.shape[0]
    OH, OW = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    out = torch.empty((B, OH, OW, Cout), device=img.device, dtype=torch.float32)
    check(lib().asis_conv3x3_c3(_stream(), _f32c(img).data_ptr(), _f32c(w).data_ptr(), out.data_ptr(), B, H, W, Cout,
                                stride, pad), "asis_conv3x3_c3")
    return out


def conv3x3_smallcout_fwd(x_hi: torch.Tensor, x_lo: Optional[torch.Tensor], w: torch.Tensor,
                          bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Direct fp32 3x3/s1/p1 conv for <= 16 output channels: x (hi [+lo]) 16-bit NHWC, w fp32 [Cout,Cin,3,3]."""
    _dev(x_hi, x_lo, w, bias)
    B, H, W, Cin = x_hi.shape
    Cout = w.shape[0]
    out = torch.empty((B, H, W, Cout), device=x_hi.device, dtype=torch.float32)
    check(lib().asis_conv3x3_smallcout_fwd(_stream(), _dt(x_hi.dtype), x_hi.data_ptr(), _p(x_lo), _f32c(w).data_ptr(),
                                           _p(_f32c(bias)), out.data_ptr(), B, H, W, Cin, Cout),
          "asis_conv3x3_smallcout_fwd")
    return out


# classifier conv with BatchNorm + ReLU + x2 upsampling evaluated on load (csrc/smallconv.hip): OFF by default — it removes 1.4 GB of
# writes and 2.1 GB of reads per step, but the on-load blend costs the two kernels as much as the removed pass took
# (profiles/r04_cls_upsample_on_load_ab.txt: 211.7 / 211.1 img/s without, 211.7 / 211.4 with)
FUSE_CLS_UP = os.environ.get("ASIS_FUSE_CLS_UP", "0") not in ("0", "")


def cls_up_ok(raw: torch.Tensor, Cout: int) -> bool:
    """shapes the upsample-on-load classifier kernels cover (csrc/smallconv.hip): fp32 NHWC raw map with 64 channels, <= 8 classes"""
    return bool(raw.dtype == torch.float32 and raw.dim() == 4 and raw.shape[3] == 64 and raw.is_contiguous()
                and Cout <= 8 and raw.shape[1] >= 4 and raw.shape[2] >= 8)


def conv3x3_smallcout_fwd_up(raw: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, w: torch.Tensor,
                             bias: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """conv3x3(upsample2(relu(raw * scale + shift))) with the upsampled map evaluated on load (include/asis_hip.h:
    asis_conv3x3_smallcout_fwd_up): raw fp32 NHWC [B, H, W, 64] -> fp32 [B, 2H, 2W, Cout]; ``dtype``: the 16-bit operand type."""
    _dev(raw, scale, shift, w, bias)
    B, H, W, Cin = raw.shape
    Cout = w.shape[0]
    out = torch.empty((B, 2 * H, 2 * W, Cout), device=raw.device, dtype=torch.float32)
    check(lib().asis_conv3x3_smallcout_fwd_up(_stream(), _dt(dtype), raw.data_ptr(), _f32c(scale).data_ptr(), _f32c(shift).data_ptr(),
                                              _f32c(w).data_ptr(), _p(_f32c(bias)), out.data_ptr(), B, H, W, Cin, Cout),
          "asis_conv3x3_smallcout_fwd_up")
    return out


def conv3x3_smallcout_wgrad_up(dy: torch.Tensor, raw: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, Cout: int,
                               inv_scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """weight gradient of the same fused classifier conv: dy 16-bit [B, 2H, 2W, 8], raw fp32 [B, H, W, 64] -> dW fp32 [Cout, 64, 3, 3]"""
    _dev(dy, raw, scale, shift, out)
    B, H, W, Cin = raw.shape
    if not dy.is_contiguous() or dy.shape[:3] != (B, 2 * H, 2 * W):
        raise ValueError("conv3x3_smallcout_wgrad_up: dy must be contiguous [B, 2H, 2W, CoP]")
    nblk = 1024
    slabs = torch.empty((nblk, Cout * 9 * Cin), device=dy.device, dtype=torch.float32)
    check(lib().asis_conv3x3_smallcout_wgrad_up(_stream(), _dt(dy.dtype), dy.data_ptr(), dy.shape[3], raw.data_ptr(), _f32c(scale).data_ptr(),
                                                _f32c(shift).data_ptr(), slabs.data_ptr(), nblk, B, H, W, Cin, Cout),
          "asis_conv3x3_smallcout_wgrad_up")
    if out is None:
        out = torch.empty((Cout, Cin, 3, 3), device=dy.device, dtype=torch.float32)
    reduce_rows(slabs, inv_scale, out.view(-1))
    return out


def conv3x3_smallcout_dgrad(dy_hi: torch.Tensor, dy_lo: Optional[torch.Tensor], w: torch.Tensor) -> torch.Tensor:
    """Input gradient of the small-Cout conv: dy 16-bit [B,H,W,CoP] (hi [+lo]) -> dx fp32 [B,H,W,Cin]."""
    _dev(dy_hi, dy_lo, w)
    B, H, W, CoP = dy_hi.shape
    Cout, Cin = w.shape[0], w.shape[1]
    dx = torch.empty((B, H, W, Cin), device=dy_hi.device, dtype=torch.float32)
    check(lib().asis_conv3x3_smallcout_dgrad(_stream(), _dt(dy_hi.dtype), dy_hi.data_ptr(), _p(dy_lo), CoP,
                                             _f32c(w).data_ptr(), dx.data_ptr(), B, H, W, Cin, Cout),
          "asis_conv3x3_smallcout_dgrad")
    return dx


def colstats(x: torch.Tensor) -> torch.Tensor:
    """fp32 [..., C] -> partial [nparts, 2, C] column sums / sums of squares."""
    _dev(x)
    Cc = x.shape[-1]
    R = x.numel() // Cc
    nparts = lib().asis_colstats_nparts(R)
    partial = torch.empty((nparts, 2, Cc), device=x.device, dtype=torch.float32)
    check(lib().asis_colstats(_stream(), _f32c(x).data_ptr(), R, Cc, partial.data_ptr()), "asis_colstats")
    return partial


def reduce_partials(partial: torch.Tensor) -> torch.Tensor:
    """partial fp32 [nparts, 2, C] -> sums float64 [2, C]."""
    _dev(partial)
    nparts, _, Cc = partial.shape
    sums = torch.empty((2, Cc), device=partial.device, dtype=torch.float64)
    check(lib().asis_reduce_partials(_stream(), _f32c(partial).data_ptr(), nparts, Cc, sums.data_ptr()),
          "asis_reduce_partials")
    return sums


def bn_finalize(sums: torch.Tensor, count: float, gamma, beta, eps: float, momentum: float, running_mean=None,
                running_var=None, num_batches_tracked=None):
    """-> (scale, shift, mean, invstd) fp32 [C]; updates the running buffers in place when given."""
    _dev(sums, gamma, beta, running_mean, running_var, num_batches_tracked)
    Cc = sums.shape[1]
    out = torch.empty((4, Cc), device=sums.device, dtype=torch.float32)
    check(lib().asis_bn_finalize(_stream(), sums.data_ptr(), float(count), Cc, _p(_f32c(gamma)), _p(_f32c(beta)),
                                 float(eps), float(momentum), _p(running_mean), _p(running_var), _p(num_batches_tracked),
                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr()),
          "asis_bn_finalize")
    return out[0], out[1], out[2], out[3]


class MxPlane(torch.Tensor):
    """The lo half of a split-precision operand in the MX form (two fp8 bytes per element, scaled by the tensor's absolute
    maximum) instead of a 16-bit rounding residual.  The producers (bn_act, bn_relu_upsample, decoder_input, bn_bwd_apply) tag it
    with that maximum (``_asis_mx_amax``); the TYPE survives views, slices and copies, the attribute does not, so a consumer
    that meets an MxPlane without its maximum (``mx_amax_of``), or as a 16-bit lo operand (``_not_mx``), raises instead of
    reading the fp8 pair as an fp16 residual."""

    @staticmethod
    def tag(t: torch.Tensor, amax: torch.Tensor) -> "MxPlane":
        p = t.as_subclass(MxPlane)
        p._asis_mx_amax = amax
        return p


def mx_amax_of(lo: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """the absolute maximum an MX-form lo plane was scaled by; None for a 16-bit residual (or no lo plane at all)"""
    if lo is None:
        return None
    amax = getattr(lo, "_asis_mx_amax", None)
    if amax is None and isinstance(lo, MxPlane):
        raise ValueError("an MX-form lo plane reached its consumer without its absolute maximum (a view / slice / copy of the "
                         "producer's tensor drops the tag): pass the producer's tensor itself")
    return amax


def _not_mx(what: str, *planes) -> None:
    for t in planes:
        if isinstance(t, MxPlane):
            raise ValueError(f"{what}: an MX-form lo plane was passed where a 16-bit rounding residual is expected")


def _lo(out: torch.Tensor, split: bool):
    return torch.empty_like(out) if split else None


def bn_eval_affine(bn) -> tuple:
    """(scale, shift) of an eval-mode nn.BatchNorm2d from its running statistics."""
    Cc = bn.running_mean.numel()
    _dev(bn.running_mean)
    out = torch.empty((2, Cc), device=bn.running_mean.device, dtype=torch.float32)
    check(lib().asis_bn_eval_affine(_stream(), _p(bn.weight.detach() if bn.weight is not None else None),
                                    _p(bn.bias.detach() if bn.bias is not None else None), bn.running_mean.data_ptr(),
                                    bn.running_var.data_ptr(), float(bn.eps), Cc, out[0].data_ptr(), out[1].data_ptr()),
          "asis_bn_eval_affine")
    return out[0], out[1]


def bn_act(x: torch.Tensor, scale, shift, relu: bool, dtype: torch.dtype, split: bool = False, mx_amax: Optional[torch.Tensor] = None):
    """-> out (and (out, out_lo) when split: the two halves of a split-precision operand).  ``mx_amax`` (device float, the output's
    absolute maximum: ``bn_relu_absmax``): the lo half in the MX form, tagged ``_asis_mx_amax`` for the consuming convolution."""
    _dev(x, scale, shift, mx_amax)
    Cc = x.shape[-1]
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    lo = _lo(out, split)
    if mx_amax is not None and split:
        check(lib().asis_bn_act_mx(_stream(), _dt(dtype), _f32c(x).data_ptr(), scale.data_ptr(), shift.data_ptr(), int(relu),
                                   out.data_ptr(), lo.data_ptr(), mx_amax.data_ptr(), x.numel() // Cc, Cc), "asis_bn_act_mx")
        return out, MxPlane.tag(lo, mx_amax)
    check(lib().asis_bn_act(_stream(), _dt(dtype), _f32c(x).data_ptr(), scale.data_ptr(), shift.data_ptr(), int(relu),
                            out.data_ptr(), _p(lo), x.numel() // Cc, Cc), "asis_bn_act")
    return (out, lo) if split else out


def bn_relu_maxpool(x: torch.Tensor, scale, shift, dtype: torch.dtype, split: bool = False):
    _dev(x, scale, shift)
    B, H, W, Cc = x.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), device=x.device, dtype=dtype)
    lo = _lo(out, split)
    check(lib().asis_bn_relu_maxpool(_stream(), _dt(dtype), _f32c(x).data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                     out.data_ptr(), _p(lo), B, H, W, Cc), "asis_bn_relu_maxpool")
    return (out, lo) if split else out


def mx_conv_ok(P: int, Cin: int, Cout: int) -> bool:
    """a split 3x3 convolution with P output pixels can take MX lo operands (csrc/gemm.hip: the fused large-tile conv forms)"""
    return Cin % 64 == 0 and P >= 256 and Cout >= 32 and Cout % 4 == 0 and _FUSED_SPLIT


def absmax_f32(x: torch.Tensor, amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """|x| maximum of an fp32 tensor whose rows are contiguous ([..., cols] with ONE outer stride) -> device float [1];
    ``amax`` given: accumulate into it (maximum over several tensors)."""
    _dev(x, amax)
    if x.dtype != torch.float32 or x.stride(-1) != 1:
        raise ValueError("absmax_f32: float32 tensor with a contiguous last dim")
    if x.is_contiguous():
        rows, cols, ld = 1, x.numel(), x.numel()
        if cols % 4:
            raise ValueError("absmax_f32: element count must be a multiple of 4")
        if cols >= 1 << 31:    # rows of 2^16 .. 2^20 elements (the kernel's column index is an int)
            cols = next((c for c in (1 << 20, 1 << 18, 1 << 16) if x.numel() % c == 0), x.shape[-1] if x.dim() > 1 else x.numel())
            rows, ld = x.numel() // cols, cols
    elif x.dim() == 3 and x[0].is_contiguous():
        rows, cols, ld = x.shape[0], x.shape[1] * x.shape[2], x.stride(0)
    elif x.dim() == 2:
        rows, cols, ld = x.shape[0], x.shape[1], x.stride(0)
    else:
        raise ValueError("absmax_f32: unsupported layout")
    reset = amax is None
    if amax is None:
        amax = torch.empty(1, device=x.device, dtype=torch.float32)
    check(lib().asis_absmax_f32(_stream(), x.data_ptr(), rows, cols, ld, amax.data_ptr(), int(reset)), "asis_absmax_f32")
    return amax


def bn_relu_absmax(x: torch.Tensor, scale, shift, relu: bool = True) -> torch.Tensor:
    """max of relu(x * scale + shift) over fp32 NHWC x -> device float [1] (the MX scale of the tensor bn_relu_upsample writes)"""
    _dev(x, scale, shift)
    C = x.shape[-1]
    amax = torch.empty(1, device=x.device, dtype=torch.float32)
    check(lib().asis_bn_relu_absmax(_stream(), _f32c(x).data_ptr(), _f32c(scale).data_ptr(), _f32c(shift).data_ptr(), x.numel() // C, C,
                                    int(relu), amax.data_ptr()), "asis_bn_relu_absmax")
    return amax


def bn_relu_upsample(x: torch.Tensor, scale, shift, factor: int, dtype: torch.dtype, split: bool = False, mx_amax=None):
    _dev(x, scale, shift)
    B, H, W, Cc = x.shape
    out = torch.empty((B, H * factor, W * factor, Cc), device=x.device, dtype=dtype)
    lo = _lo(out, split)
    if mx_amax is not None:    # lo in the MX form (two fp8 bytes per element), tagged with the tensor's absolute maximum
        if lo is None:
            raise ValueError("bn_relu_upsample: mx_amax needs split=True")
        check(lib().asis_bn_relu_upsample_mx(_stream(), _dt(dtype), _f32c(x).data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                             out.data_ptr(), lo.data_ptr(), _f32c(mx_amax).data_ptr(), B, H, W, Cc, factor),
              "asis_bn_relu_upsample_mx")
        return out, MxPlane.tag(lo, mx_amax)
    check(lib().asis_bn_relu_upsample(_stream(), _dt(dtype), _f32c(x).data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                      out.data_ptr(), _p(lo), B, H, W, Cc, factor), "asis_bn_relu_upsample")
    return (out, lo) if split else out


def pack_conv_weight(w: torch.Tensor, mode: int, dtype: torch.dtype, part: int = 0) -> torch.Tensor:
    """fp32 [Cout,Cin,KH,KW] -> mode 0: [Cout, KH*KW*Cin]; mode 1 (dgrad): [Cin, KH*KW*CoP].
    part=1 returns the rounding residual (second half of a split-precision operand)."""
    _dev(w)
    Cout, Cin, KH, KW = w.shape
    CoP = (Cout + 7) // 8 * 8
    rows, K = (Cout, KH * KW * Cin) if mode == 0 else (Cin, KH * KW * CoP)
    ld = (K + 7) // 8 * 8
    out = torch.empty((rows, ld), device=w.device, dtype=dtype)
    check(lib().asis_pack_conv_weight(_stream(), _dt(dtype), _f32c(w).data_ptr(), out.data_ptr(), Cout, Cin, KH, KW,
                                      mode, ld, int(part)), "asis_pack_conv_weight")
    return out[:, :K] if ld != K else out


def absmax16(x: torch.Tensor) -> torch.Tensor:
    """|x| maximum of a 16-bit [rows, cols] tensor (contiguous columns, cols % 8 == 0) -> device float [1]"""
    _dev(x)
    if x.dtype not in (torch.float16, torch.bfloat16) or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("absmax16: 16-bit [rows, cols] tensor with contiguous columns")
    amax = torch.empty(1, device=x.device, dtype=torch.float32)
    check(lib().asis_absmax_16(_stream(), _dt(x.dtype), x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), amax.data_ptr()), "asis_absmax_16")
    return amax


def mx_from_pair(hi: torch.Tensor, lo: torch.Tensor, wside: bool = False, amax: Optional[torch.Tensor] = None):
    """(hi, lo) 16-bit planes of a split-precision operand -> (its MX plane with the layout of ``hi``, amax [1]): the ``a_lo`` /
    ``b_lo`` (``wside``) + ``mx`` operands of a dense ``gemm`` (include/asis_hip.h: asis_mx_from_pair)."""
    _dev(hi, lo, amax)
    if hi.dtype != lo.dtype or hi.dim() != 2 or hi.stride() != lo.stride() or hi.stride(1) != 1 or hi.shape != lo.shape:
        raise ValueError("mx_from_pair: two 16-bit [rows, cols] planes with equal layout")
    if amax is None:
        amax = absmax16(hi)
    out = torch.empty_strided(hi.shape, hi.stride(), device=hi.device, dtype=hi.dtype) if hi.stride(0) != hi.shape[1] else torch.empty_like(hi)
    check(lib().asis_mx_from_pair(_stream(), _dt(hi.dtype), hi.data_ptr(), lo.data_ptr(), hi.stride(0), out.data_ptr(), out.stride(0),
                                  hi.shape[0], hi.shape[1], amax.data_ptr(), int(wside)), "asis_mx_from_pair")
    return out, amax


def pack_conv_weight_mx(w: torch.Tensor, mode: int, dtype: torch.dtype):
    """-> (MX form of the weight's lo operand (weight side: (lo8, hi8) per element), same shape as pack_conv_weight's, and the
    weight's absolute maximum as a device float [1])"""
    _dev(w)
    Cout, Cin, KH, KW = w.shape
    CoP = (Cout + 7) // 8 * 8
    rows, K = (Cout, KH * KW * Cin) if mode == 0 else (Cin, KH * KW * CoP)
    ld = (K + 7) // 8 * 8
    wf = _f32c(w)
    amax = absmax_f32(wf.view(Cout, -1) if (Cin * KH * KW) % 4 == 0 else wf.view(1, -1))
    out = torch.empty((rows, ld), device=w.device, dtype=dtype)
    check(lib().asis_pack_conv_weight_mx(_stream(), _dt(dtype), wf.data_ptr(), out.data_ptr(), Cout, Cin, KH, KW, mode, ld,
                                         amax.data_ptr()), "asis_pack_conv_weight_mx")
    return (out[:, :K] if ld != K else out), amax


def conv_weight_absmax(w: torch.Tensor) -> torch.Tensor:
    """absolute maximum of an OIHW fp32 convolution weight -> device float [1] (the MX planes' scale)"""
    wf = _f32c(w)
    return absmax_f32(wf.view(w.shape[0], -1) if (wf.numel() // w.shape[0]) % 4 == 0 else wf.view(1, -1))


def pack_conv_weight_pair(w: torch.Tensor, mode: int, dtype: torch.dtype, mx: bool = False, amax: Optional[torch.Tensor] = None):
    """-> (16-bit weight, its lo operand, amax | None) in ONE pass over the fp32 weight: lo = the rounding residual, or (``mx``)
    the MX form scaled by the weight's absolute maximum (``amax`` if the caller holds it, else one reduction pass).  Same
    layouts as ``pack_conv_weight``."""
    _dev(w, amax)
    Cout, Cin, KH, KW = w.shape
    CoP = (Cout + 7) // 8 * 8
    rows, K = (Cout, KH * KW * Cin) if mode == 0 else (Cin, KH * KW * CoP)
    ld = (K + 7) // 8 * 8
    wf = _f32c(w)
    if mx and amax is None:
        amax = conv_weight_absmax(wf)
    elif not mx:
        amax = None
    hi = torch.empty((rows, ld), device=w.device, dtype=dtype)
    lo = torch.empty((rows, ld), device=w.device, dtype=dtype)
    check(lib().asis_pack_conv_weight_pair(_stream(), _dt(dtype), wf.data_ptr(), hi.data_ptr(), lo.data_ptr(), Cout, Cin, KH, KW, mode, ld,
                                           _p(amax)), "asis_pack_conv_weight_pair")
    if ld != K:
        hi, lo = hi[:, :K], lo[:, :K]
    return hi, lo, amax


def _rows3(t: torch.Tensor, D: int, what: str):
    if t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != D:
        raise ValueError(f"{what}: expected float32 [B, n, {D}] with contiguous rows (batch stride free)")


def decoder_input(xs: torch.Tensor, c4: torch.Tensor, vit: torch.Tensor, hw, c4_hw, dtype: torch.dtype,
                  split: bool = False, mx: bool = False):
    """train.py:389-406: xs, vit fp32 [B, h*w, D]; c4 fp32 [B, h4*w4, D] (batch strides free) -> [B,h,w,3D]."""
    _dev(xs, c4, vit)
    B, _, D = xs.shape
    h, w = hw
    h4, w4 = c4_hw
    for t, n in ((xs, "xs"), (c4, "c4"), (vit, "vit")):
        _rows3(t, D, "decoder_input " + n)
    out = torch.empty((B, h, w, 3 * D), device=xs.device, dtype=dtype)
    lo = _lo(out, split)
    if mx and split:           # lo in the MX form: one absolute maximum over the three sources
        amax = absmax_f32(xs)
        absmax_f32(c4, amax)
        absmax_f32(vit, amax)
        check(lib().asis_decoder_input_mx(_stream(), _dt(dtype), xs.data_ptr(), xs.stride(0), c4.data_ptr(), c4.stride(0),
                                          vit.data_ptr(), vit.stride(0), out.data_ptr(), lo.data_ptr(), amax.data_ptr(), B, h, w,
                                          h4, w4, D), "asis_decoder_input_mx")
        return out, MxPlane.tag(lo, amax)
    check(lib().asis_decoder_input(_stream(), _dt(dtype), xs.data_ptr(), xs.stride(0), c4.data_ptr(), c4.stride(0),
                                   vit.data_ptr(), vit.stride(0), out.data_ptr(), _p(lo), B, h, w, h4, w4, D),
          "asis_decoder_input")
    return (out, lo) if split else out


def swiglu_rows(Hd: int, device) -> torch.Tensor:
    """row order of w12 (and its bias) for the ACT_SILU_MUL epilogue of ``gemm``: groups of 16 x1 rows followed by the 16 x2 rows
    of the same hidden columns (include/asis_hip.h) -> int64 [2 * Hd]"""
    if Hd % 16:
        raise ValueError("swiglu_rows: the hidden width must be a multiple of 16")
    g = torch.arange(Hd, device=device).view(Hd // 16, 1, 16)
    return torch.cat((g, g + Hd), 1).reshape(-1)


def swiglu_fused_ok(M: int, Hd: int, K: int, split: bool, mx: bool) -> bool:
    """shapes / operand forms the SwiGLU epilogue covers (include/asis_hip.h: ASIS_ACT_SILU_MUL); ASIS_SWIGLU_FUSED=0: never"""
    return (_SWIGLU_FUSED and Hd % 16 == 0 and 2 * Hd >= 256 and M >= 256 and K % 64 == 0 and (mx or not split)
            and os.environ.get("ASIS_GEMM_BIG", "1") != "0" and os.environ.get("ASIS_GEMM_8P_M16", "1") != "0")


_SWIGLU_FUSED = os.environ.get("ASIS_SWIGLU_FUSED", "1") not in ("0", "")


def swiglu(x12: torch.Tensor, dtype: torch.dtype, split: bool = False):
    """fp32 [R, 2*Hd] = [x1 | x2] -> 16-bit [R, Hd] = silu(x1) * x2 (dinov2/layers/swiglu_ffn.py:30-34); ``split``: -> (hi, lo)
    with lo the rounding residual (a split-precision operand pair)."""
    _dev(x12)
    R, two = x12.shape
    out = torch.empty((R, two // 2), device=x12.device, dtype=dtype)
    lo = torch.empty_like(out) if split else None
    check(lib().asis_swiglu_split(_stream(), _dt(dtype), _f32c(x12).data_ptr(), out.data_ptr(), _p(lo), R, two // 2), "asis_swiglu")
    return (out, lo) if split else out


def copy_channels(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[..., :] = src[..., :] for 2-D views [rows, C] whose rows are contiguous but strided (concat / split)."""
    _dev(src, dst)
    if src.dim() != 2 or dst.dim() != 2 or src.shape != dst.shape or src.stride(1) != 1 or dst.stride(1) != 1 \
            or src.dtype != dst.dtype:
        raise ValueError("copy_channels: expected matching 2-D row views")
    es = src.element_size()
    check(lib().asis_copy_channels(_stream(), src.data_ptr(), src.stride(0) * es, dst.data_ptr(), dst.stride(0) * es,
                                   src.shape[0], src.shape[1] * es), "asis_copy_channels")


def add_f32(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a + b for float32 [B, n, D] tensors with contiguous rows and free batch strides."""
    _dev(a, b, out)
    B, n, D = a.shape
    if out is None:
        out = torch.empty((B, n, D), device=a.device, dtype=torch.float32)
    for t, nm in ((a, "a"), (b, "b"), (out, "out")):
        _rows3(t, D, "add_f32 " + nm)
    check(lib().asis_add_f32(_stream(), a.data_ptr(), b.data_ptr(), out.data_ptr(), n * D, B, a.stride(0), b.stride(0),
                             out.stride(0)), "asis_add_f32")
    return out


# --------------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------------
def maxpool2_fwd(x: torch.Tensor, x_lo: Optional[torch.Tensor], save_idx: bool = True):
    """MaxPool2d(2) on a split-precision NHWC map: (hi, lo|None) [B,H,W,C] -> (hi, lo|None) [B,H/2,W/2,C], idx uint8."""
    _dev(x, x_lo)
    _not_mx("maxpool2_fwd", x_lo)
    B, H, W, Cc = x.shape
    if not x.is_contiguous() or (x_lo is not None and (not x_lo.is_contiguous() or x_lo.shape != x.shape)):
        raise ValueError("maxpool2_fwd: contiguous NHWC operands of one shape expected")
    out = torch.empty((B, H // 2, W // 2, Cc), device=x.device, dtype=x.dtype)
    out_lo = torch.empty_like(out) if x_lo is not None else None
    idx = torch.empty((B, H // 2, W // 2, Cc), device=x.device, dtype=torch.uint8) if save_idx else None
    check(lib().asis_maxpool2_fwd(_stream(), _dt(x.dtype), x.data_ptr(), _p(x_lo), out.data_ptr(), _p(out_lo), _p(idx), B, H,
                                  W, Cc), "asis_maxpool2_fwd")
    return out, out_lo, idx


def maxpool2_bwd(dy: torch.Tensor, idx: torch.Tensor, dx: torch.Tensor) -> torch.Tensor:
    """dx fp32 [B,H,W,C] += scatter of dy fp32 [B,H/2,W/2,C] at the saved argmax."""
    _dev(dy, idx, dx)
    B, H, W, Cc = dx.shape
    if dy.shape != (B, H // 2, W // 2, Cc) or idx.shape != dy.shape or not dx.is_contiguous():
        raise ValueError("maxpool2_bwd: shape mismatch")
    check(lib().asis_maxpool2_bwd(_stream(), _f32c(dy).data_ptr(), idx.data_ptr(), _f32c(dx).data_ptr(), B, H, W, Cc),
          "asis_maxpool2_bwd")
    return dx



def cls_l2norm(x: torch.Tensor, B: int, N: int, C: int):
    """class rows of the stacked fp32 [B*(N+C), D] matrix -> (chat fp32 [B,C,D] = x / ||x||, inv_c fp32 [B*C])."""
    _dev(x)
    D = x.shape[1]
    if x.dtype != torch.float32 or not x.is_contiguous() or x.shape[0] != B * (N + C):
        raise ValueError("cls_l2norm: contiguous float32 [B*(N+C), D] expected")
    chat = torch.empty((B, C, D), device=x.device, dtype=torch.float32)
    inv = torch.empty((B * C,), device=x.device, dtype=torch.float32)
    check(lib().asis_cls_l2norm(_stream(), x.data_ptr(), chat.data_ptr(), inv.data_ptr(), B, N, C, D), "asis_cls_l2norm")
    return chat, inv


def cls_l2norm_bwd(dchat: torch.Tensor, chat: torch.Tensor, inv_c: torch.Tensor, dx: torch.Tensor, N: int) -> None:
    """writes the class rows of dx (stacked fp32 [B*(N+C), D]); the patch rows are left as they are."""
    _dev(dchat, chat, inv_c, dx)
    B, C, D = chat.shape
    if dx.dtype != torch.float32 or not dx.is_contiguous() or tuple(dx.shape) != (B * (N + C), D) or dchat.shape != chat.shape:
        raise ValueError("cls_l2norm_bwd: shape mismatch")
    check(lib().asis_cls_l2norm_bwd(_stream(), _f32c(dchat).data_ptr(), chat.data_ptr(), inv_c.data_ptr(), dx.data_ptr(), B, N, C, D),
          "asis_cls_l2norm_bwd")


def mask_logits_fwd(P: torch.Tensor, chat: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, N: int):
    """stacked P fp32 [B*(N+C), D] (patch rows), chat [B,C,D] -> (logits fp32 [B*N, C] = LayerNorm_C(cosines), cosm, inv_p)."""
    _dev(P, chat, gamma, beta)
    B, C, D = chat.shape
    if P.dtype != torch.float32 or not P.is_contiguous() or tuple(P.shape) != (B * (N + C), D):
        raise ValueError("mask_logits_fwd: contiguous float32 [B*(N+C), D] expected")
    logits = torch.empty((B * N, C), device=P.device, dtype=torch.float32)
    cosm = torch.empty((B * N, C), device=P.device, dtype=torch.float32)
    inv_p = torch.empty((B * N,), device=P.device, dtype=torch.float32)
    check(lib().asis_mask_logits_fwd(_stream(), P.data_ptr(), chat.data_ptr(), _f32c(gamma).data_ptr(), _f32c(beta).data_ptr(),
                                     float(eps), logits.data_ptr(), cosm.data_ptr(), inv_p.data_ptr(), B, N, C, D),
          "asis_mask_logits_fwd")
    return logits, cosm, inv_p


def mask_logits_bwd(dlogits: torch.Tensor, cosm: torch.Tensor, inv_p: torch.Tensor, P: torch.Tensor, chat: torch.Tensor,
                    gamma: torch.Tensor, eps: float, dP: torch.Tensor, N: int):
    """-> (dcos fp32 [B*N, C], partial fp32 [nblk, 2, C]); writes the patch rows of dP (stacked, same shape as P)."""
    _dev(dlogits, cosm, inv_p, P, chat, gamma, dP)
    B, C, D = chat.shape
    if dP.shape != P.shape or dP.dtype != torch.float32 or not dP.is_contiguous() or tuple(dlogits.shape) != (B * N, C):
        raise ValueError("mask_logits_bwd: shape mismatch")
    dcos = torch.empty((B * N, C), device=P.device, dtype=torch.float32)
    part = torch.empty((lib().asis_mask_logits_nblk(B, N), 2, C), device=P.device, dtype=torch.float32)
    check(lib().asis_mask_logits_bwd(_stream(), _f32c(dlogits).data_ptr(), cosm.data_ptr(), inv_p.data_ptr(), P.data_ptr(),
                                     chat.data_ptr(), _f32c(gamma).data_ptr(), float(eps), dP.data_ptr(), dcos.data_ptr(),
                                     part.data_ptr(), B, N, C, D), "asis_mask_logits_bwd")
    return dcos, part


def mask_dchat(dcos: torch.Tensor, P: torch.Tensor, inv_p: torch.Tensor, B: int, N: int, C: int) -> torch.Tensor:
    _dev(dcos, P, inv_p)
    D = P.shape[1]
    out = torch.empty((B, C, D), device=P.device, dtype=torch.float32)
    check(lib().asis_mask_dchat(_stream(), dcos.data_ptr(), P.data_ptr(), inv_p.data_ptr(), out.data_ptr(), B, N, C, D),
          "asis_mask_dchat")
    return out


def nearest_tables(src: int, dst: int, device) -> tuple:
    """F.interpolate(mode='nearest') index tables for one axis: (src index of every dst index int32 [dst], first dst index of
    every src index int32 [src + 1]).  ATen (UpSample.h nearest_idx): identity when equal, dst >> 1 for an exact doubling,
    else min(floor(dst_index * float32(src / dst)), src - 1) in float32 arithmetic."""
    import numpy as np
    d = np.arange(dst, dtype=np.int64)
    if dst == src:
        idx = d
    elif dst == 2 * src:
        idx = d >> 1
    else:
        scale = np.float32(src) / np.float32(dst)
        idx = np.minimum(np.floor(d.astype(np.float32) * scale).astype(np.int64), src - 1)
    first = np.searchsorted(idx, np.arange(src + 1), side="left")
    return (torch.from_numpy(idx.astype(np.int32)).to(device), torch.from_numpy(first.astype(np.int32)).to(device))


def nearest_add_relu(x: torch.Tensor, x_lo: Optional[torch.Tensor], r: torch.Tensor, r_lo: Optional[torch.Tensor],
                     ys: torch.Tensor, xs: torch.Tensor) -> None:
    """x (+x_lo) 16-bit NHWC [B,H,W,C] <- relu(x + nearest-resized r [B,h,w,C]) in place (FCUUp + FusionModel)."""
    _dev(x, x_lo, r, r_lo, ys, xs)
    B, H, W, C = x.shape
    _, h, w, Cr = r.shape
    if Cr != C or r.shape[0] != B or ys.numel() != H or xs.numel() != W or ys.dtype != torch.int32 or xs.dtype != torch.int32:
        raise ValueError("nearest_add_relu: shape / table mismatch")
    if not (x.is_contiguous() and r.is_contiguous()) or r.dtype != x.dtype:
        raise ValueError("nearest_add_relu: contiguous maps of one 16-bit dtype expected")
    check(lib().asis_nearest_add_relu(_stream(), _dt(x.dtype), x.data_ptr(), _p(x_lo), r.data_ptr(), _p(r_lo), ys.data_ptr(),
                                      xs.data_ptr(), B, H, W, h, w, C), "asis_nearest_add_relu")


def nearest_sum(g: torch.Tensor, h: int, w: int, y0: torch.Tensor, x0: torch.Tensor) -> torch.Tensor:
    """Transpose of the nearest resize: g fp32 [B,H,W,C] -> fp32 [B,h,w,C] (sums over each source pixel's destination block)."""
    _dev(g, y0, x0)
    B, H, W, C = g.shape
    if g.dtype != torch.float32 or not g.is_contiguous() or y0.numel() != h + 1 or x0.numel() != w + 1:
        raise ValueError("nearest_sum: fp32 contiguous gradient and [h+1] / [w+1] int32 tables expected")
    out = torch.empty((B, h, w, C), device=g.device, dtype=torch.float32)
    check(lib().asis_nearest_sum(_stream(), g.data_ptr(), out.data_ptr(), y0.data_ptr(), x0.data_ptr(), B, H, W, h, w, C),
          "asis_nearest_sum")
    return out

def convt2x2_scatter(G: torch.Tensor, dst: torch.Tensor, dst_lo: Optional[torch.Tensor], B: int, H: int, W: int, coff: int,
                     padT: int = 0, padL: int = 0) -> None:
    """G fp32 [B*H*W, 4*Cout] (ConvTranspose2d k=2 s=2 as a GEMM) -> channels [coff, coff+Cout) of the NHWC 16-bit
    buffer(s) dst [B,H2,W2,Ctot] at pixel offset (padT, padL)."""
    _dev(G, dst, dst_lo)
    Cout = G.shape[1] // 4
    _, H2, W2, Ctot = dst.shape
    if G.shape[0] != B * H * W or not dst.is_contiguous():
        raise ValueError("convt2x2_scatter: shape mismatch")
    check(lib().asis_convt2x2_scatter(_stream(), _dt(dst.dtype), _f32c(G).data_ptr(), dst.data_ptr(), _p(dst_lo), B, H, W,
                                      Cout, H2, W2, Ctot, coff, padT, padL), "asis_convt2x2_scatter")


def conv1x1_dgrad_small(d_hi: torch.Tensor, d_lo: Optional[torch.Tensor], w: torch.Tensor, C: int) -> torch.Tensor:
    """input gradient of a 1x1 convolution with C <= 8 output channels: (d_hi + d_lo)[M, :C] @ w[C, Cq] -> fp32 [M, Cq]"""
    _dev(d_hi, d_lo, w)
    _not_mx("conv1x1_dgrad_small", d_lo)
    M, ldd = d_hi.shape[0], d_hi.stride(0)
    Cq = w.shape[1]
    if d_hi.dim() != 2 or d_hi.stride(1) != 1 or (d_lo is not None and (d_lo.stride() != d_hi.stride() or d_lo.dtype != d_hi.dtype)) \
            or w.dtype != torch.float32 or w.shape[0] != C or not w.is_contiguous():
        raise ValueError("conv1x1_dgrad_small: d_hi / d_lo 16-bit [M, >= C] row views with one layout, w contiguous float32 [C, Cq]")
    out = torch.empty((M, Cq), device=d_hi.device, dtype=torch.float32)
    check(lib().asis_conv1x1_dgrad_small(_stream(), _dt(d_hi.dtype), d_hi.data_ptr(), _p(d_lo), ldd, w.data_ptr(), out.data_ptr(), M, Cq, C),
          "asis_conv1x1_dgrad_small")
    return out


def convt2x2_gather(dcat: torch.Tensor, B: int, H: int, W: int, Cout: int, coff: int, padT: int, padL: int,
                    dtype: torch.dtype, split: bool):
    """d cat fp32 [B,H2,W2,Ctot] -> (dG hi, dG lo|None 16-bit [B*H*W, 4*Cout], d bias partial sums fp32 [nblk, Cout])."""
    _dev(dcat)
    _, H2, W2, Ctot = dcat.shape
    dG = torch.empty((B * H * W, 4 * Cout), device=dcat.device, dtype=dtype)
    dG_lo = torch.empty_like(dG) if split else None
    check(lib().asis_convt2x2_gather(_stream(), _dt(dtype), _f32c(dcat).data_ptr(), dG.data_ptr(), _p(dG_lo), B, H, W, Cout,
                                     H2, W2, Ctot, coff, padT, padL), "asis_convt2x2_gather")
    nblk = lib().asis_convt2x2_bias_nblk(B * 4 * H * W)
    partial = torch.empty((nblk, Cout), device=dcat.device, dtype=torch.float32)
    check(lib().asis_convt2x2_bias_grad(_stream(), dcat.data_ptr(), partial.data_ptr(), B, H, W, Cout, H2, W2, Ctot, coff,
                                        padT, padL), "asis_convt2x2_bias_grad")
    return dG, dG_lo, partial


LOSS_DICE, LOSS_IOU, LOSS_SOFTDICE, LOSS_TVERSKY, LOSS_NONE = 0, 1, 2, 3, 4


def seg_loss_fwd(logits: torch.Tensor, target: torch.Tensor, n_region: int, mode: int = LOSS_DICE, eps: float = 1e-19,
                 n_ce: int = 0, ce_weight: Optional[torch.Tensor] = None, grad_scale: float = 1.0):
    """logits fp32 NHWC [B,h,w,C]; target int64 [B,H,W] -> (loss [1], coef [B*C*2+1], sums [B,C,3]).
    Region term `mode` on softmax^n_region(resize(logits)) plus (n_ce > 0) the cross entropy of
    log softmax(softmax^(n_ce-1)(resize(logits))) — see include/asis_hip.h for the reference losses each case is."""
    _dev(logits, target, ce_weight)
    B, h, w, Cc = logits.shape
    H, W = target.shape[-2:]
    if target.dtype != torch.int64 or not target.is_contiguous():
        raise ValueError("seg_loss_fwd: target must be contiguous int64 [B,H,W]")
    if ce_weight is not None and (ce_weight.numel() != Cc or ce_weight.dtype != torch.float32):
        raise ValueError("seg_loss_fwd: ce_weight must be float32 [C]")
    nblk = lib().asis_dice_nblk(H, W)
    partial = torch.empty((B, nblk, Cc * 3 + 2), device=logits.device, dtype=torch.float32)
    sums = torch.empty((B, Cc, 3), device=logits.device, dtype=torch.float32)
    loss = torch.empty((1,), device=logits.device, dtype=torch.float32)
    coef = torch.empty((B * Cc * 2 + 1,), device=logits.device, dtype=torch.float32)
    check(lib().asis_seg_loss_fwd(_stream(), _f32c(logits).data_ptr(), target.data_ptr(), _p(_f32c(ce_weight)), B, h, w, H, W,
                                  Cc, int(n_region), int(mode), float(eps), int(n_ce), float(grad_scale), partial.data_ptr(),
                                  sums.data_ptr(), loss.data_ptr(), coef.data_ptr()), "asis_seg_loss_fwd")
    return loss, coef, sums


def seg_loss_bwd(logits: torch.Tensor, target: torch.Tensor, coef: torch.Tensor, n_region: int, mode: int = LOSS_DICE,
                 n_ce: int = 0, ce_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> dz fp32 [B,H,W,C] = d loss / d resized logits (times the grad_scale given to the forward)."""
    _dev(logits, target, coef, ce_weight)
    B, h, w, Cc = logits.shape
    H, W = target.shape[-2:]
    if coef.numel() != B * Cc * 2 + 1:
        raise ValueError("seg_loss_bwd: coef must be the B*C*2+1 floats of seg_loss_fwd")
    dz = torch.empty((B, H, W, Cc), device=logits.device, dtype=torch.float32)
    check(lib().asis_seg_loss_bwd(_stream(), _f32c(logits).data_ptr(), target.data_ptr(), coef.data_ptr(),
                                  _p(_f32c(ce_weight)), B, h, w, H, W, Cc, int(n_region), int(mode), int(n_ce),
                                  dz.data_ptr()), "asis_seg_loss_bwd")
    return dz


def dice_fwd(logits: torch.Tensor, target: torch.Tensor, n_softmax: int, eps: float = 1e-19, grad_scale: float = 1.0,
             mode: int = 0):
    """Region-only shorthand: mode 0 = Dice (segloss/dice.py), mode 1 = soft IoU (segloss/iou_multi.py, eps = smooth)."""
    return seg_loss_fwd(logits, target, n_softmax, mode, eps, 0, None, grad_scale)


def ce_acc(logits: torch.Tensor, target: torch.Tensor, weight: Optional[torch.Tensor] = None, counts: bool = False):
    """logits fp32 NHWC [B,h,w,C], target int64 [B,H,W] -> fp32 [3] = (sum w*nll, sum w, #correct); with ``counts``
    also int32 [C,3] = #(target == c), #(argmax == c), #(both) (the inputs of ch_iou / isi_iou)."""
    _dev(logits, target, weight)
    B, h, w, Cc = logits.shape
    H, W = target.shape[-2:]
    nblk = lib().asis_ce_acc_nblk(B * H * W)
    partial = torch.empty((nblk, 3), device=logits.device, dtype=torch.float32)
    cnt = torch.empty((Cc, 3), device=logits.device, dtype=torch.int32) if counts else None
    check(lib().asis_ce_acc_counts(_stream(), _f32c(logits).data_ptr(), target.data_ptr(), _p(_f32c(weight)), B, h, w, H, W,
                                   Cc, partial.data_ptr(), _p(cnt)), "asis_ce_acc")
    red = reduce_rows(partial)
    return (red, cnt) if counts else red


def dice_bwd(logits: torch.Tensor, target: torch.Tensor, coef: torch.Tensor, n_softmax: int) -> torch.Tensor:
    return seg_loss_bwd(logits, target, coef, n_softmax, LOSS_DICE, 0, None)


def resize_bilinear_fwd(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """fp32 NHWC [B,h,w,C] -> [B,H,W,C], F.interpolate(mode="bilinear", align_corners=False)."""
    _dev(x)
    B, h, w, Cc = x.shape
    out = torch.empty((B, H, W, Cc), device=x.device, dtype=torch.float32)
    check(lib().asis_resize_bilinear_fwd(_stream(), _f32c(x).data_ptr(), B, h, w, H, W, Cc, out.data_ptr()),
          "asis_resize_bilinear_fwd")
    return out


def resize_bilinear_bwd(dz: torch.Tensor, h: int, w: int, dtype: torch.dtype, split: bool = False):
    """dz fp32 [B,H,W,C] -> (d 16-bit [B,h,w,CP], partial [nblk, C]) (+ d_lo inserted second when split)."""
    _dev(dz)
    B, H, W, Cc = dz.shape
    f32 = dtype == torch.float32
    CP = Cc if f32 else (Cc + 7) // 8 * 8
    nblk = lib().asis_resize_bwd_nblk(B * h * w)
    out = torch.empty((B, h, w, CP), device=dz.device, dtype=dtype)
    partial = torch.empty((nblk, Cc), device=dz.device, dtype=torch.float32)
    lo = _lo(out, split and not f32)
    check(lib().asis_resize_bilinear_bwd(_stream(), 2 if f32 else _dt(dtype), _f32c(dz).data_ptr(), B, H, W, h, w, Cc, CP,
                                         out.data_ptr(), _p(lo), partial.data_ptr()), "asis_resize_bilinear_bwd")
    return (out, lo, partial) if split else (out, partial)


def reduce_rows(partial: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """partial fp32 [n, K] -> out fp32 [K] = scale * column sums (double accumulate, fixed order)."""
    _dev(partial, out)
    n, K = partial.shape[0], partial.numel() // partial.shape[0]
    if out is None:
        out = torch.empty((K,), device=partial.device, dtype=torch.float32)
    check(lib().asis_reduce_rows(_stream(), _f32c(partial).data_ptr(), n, K, float(scale), out.data_ptr()),
          "asis_reduce_rows")
    return out


# --------------------------------------------------------------------------------------------
# backward / optimizer
# --------------------------------------------------------------------------------------------
def upsample_bn_relu_bwd(dU: torch.Tensor, x: torch.Tensor, scale, shift, mean, invstd, factor: int):
    """-> (g fp32 [B,H,W,C], partial [nblk, 2, C])."""
    _dev(dU, x)
    B, H, W, Cc = x.shape
    nblk = lib().asis_bn_bwd_nblk(B * H * W, Cc)
    g = torch.empty_like(x)
    partial = torch.empty((nblk, 2, Cc), device=x.device, dtype=torch.float32)
    check(lib().asis_upsample_bn_relu_bwd(_stream(), _f32c(dU).data_ptr(), _f32c(x).data_ptr(), scale.data_ptr(),
                                          shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), g.data_ptr(),
                                          partial.data_ptr(), B, H, W, Cc, factor), "asis_upsample_bn_relu_bwd")
    return g, partial


def maxpool_bn_relu_bwd(dy: torch.Tensor, x: torch.Tensor, scale, shift, mean, invstd):
    """dy fp32 [B,OH,OW,C] (gradient of BN+ReLU+MaxPool(3,2,1) output), x raw [B,H,W,C] -> (g, partial [nblk,2,C])."""
    _dev(dy, x)
    B, H, W, Cc = x.shape
    g = torch.empty_like(x)
    partial = torch.empty((lib().asis_bn_bwd_nblk(B * H * W, Cc), 2, Cc), device=x.device, dtype=torch.float32)
    check(lib().asis_maxpool_bn_relu_bwd(_stream(), _f32c(dy).data_ptr(), _f32c(x).data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                         mean.data_ptr(), invstd.data_ptr(), g.data_ptr(), partial.data_ptr(), B, H, W, Cc),
          "asis_maxpool_bn_relu_bwd")
    return g, partial


def dilate2(x: torch.Tensor, x_lo: Optional[torch.Tensor], Hd: int, Wd: int):
    """16-bit NHWC [B,OH,OW,C] (+lo) -> [B,Hd,Wd,C] with the input at even positions, zeros elsewhere."""
    _dev(x, x_lo)
    _not_mx("dilate2", x_lo)
    B, OH, OW, Cc = x.shape
    out = torch.empty((B, Hd, Wd, Cc), device=x.device, dtype=x.dtype)
    out_lo = torch.empty_like(out) if x_lo is not None else None
    check(lib().asis_dilate2(_stream(), _dt(x.dtype), x.data_ptr(), _p(x_lo), out.data_ptr(), _p(out_lo), B, OH, OW, Hd, Wd, Cc),
          "asis_dilate2")
    return out, out_lo


def bn_bwd_apply(g: torch.Tensor, x: torch.Tensor, mean, invstd, gamma, dgamma, dbeta, count: float,
                 dtype: torch.dtype, split: bool = False, mx: bool = False):
    """-> (dx 16-bit same shape as x, [dx_lo when split,] partial [nblk, C] column sums of dx).  ``mx`` (with ``split``): dx_lo in
    the MX form, scaled by max |dx| from one extra pass over g and x (asis_bn_bwd_absmax), tagged ``_asis_mx_amax``."""
    _dev(g, x)
    Cc = x.shape[-1]
    R = x.numel() // Cc
    nblk = lib().asis_bn_bwd_nblk(R, Cc)
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    partial = torch.empty((nblk, Cc), device=x.device, dtype=torch.float32)
    lo = _lo(out, split)
    args = (_f32c(g).data_ptr(), _f32c(x).data_ptr(), mean.data_ptr(), invstd.data_ptr(), _f32c(gamma).data_ptr(), dgamma.data_ptr(),
            dbeta.data_ptr(), float(count))
    if mx and split:
        amax = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib().asis_bn_bwd_absmax(_stream(), *args, amax.data_ptr(), R, Cc), "asis_bn_bwd_absmax")
        check(lib().asis_bn_bwd_apply_mx(_stream(), _dt(dtype), *args, out.data_ptr(), lo.data_ptr(), amax.data_ptr(), partial.data_ptr(),
                                         R, Cc), "asis_bn_bwd_apply_mx")
        return out, MxPlane.tag(lo, amax), partial
    check(lib().asis_bn_bwd_apply(_stream(), _dt(dtype), *args, out.data_ptr(), _p(lo), partial.data_ptr(), R, Cc), "asis_bn_bwd_apply")
    return (out, lo, partial) if split else (out, partial)


# ASIS_WGRAD_HALO=0: the implicit-GEMM form of asis_wgrad for the narrow 3x3 decoder stages too (A/B: profiles/r05_wgrad_halo_ab.txt)
WGRAD_HALO = int(os.environ.get("ASIS_WGRAD_HALO", "1") or 0)


def wgrad(dy: torch.Tensor, x_nhwc: torch.Tensor, Cout: int, KH: int, KW: int, stride: int, pad: int,
          inv_scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy 16-bit [B,OH,OW,CoP]; x 16-bit [B,H,W,Cin] -> dW fp32 [Cout,Cin,KH,KW] (scaled by inv_scale)."""
    _dev(dy, x_nhwc, out)
    if not dy.is_contiguous() or not x_nhwc.is_contiguous():
        raise ValueError("wgrad: dy and x must be contiguous NHWC")
    Bn, H, W, Cin = x_nhwc.shape
    _, OH, OW, CoP = dy.shape
    P = Bn * OH * OW
    Ntot = KH * KW * Cin
    if (KH == 3 and KW == 3 and stride == 1 and pad == 1 and Cout <= 4 and Cin in (8, 16, 32, 64)
            and os.environ.get("ASIS_WGRAD_SMALLCOUT", "1") != "0"):
        # classifier convs: a handful of output channels -> direct fp32 kernel instead of an MFMA tile that is mostly padding
        nblk = 1024
        slabs = torch.empty((nblk, Cout * Ntot), device=dy.device, dtype=torch.float32)
        check(lib().asis_conv3x3_smallcout_wgrad(_stream(), _dt(dy.dtype), dy.data_ptr(), CoP, x_nhwc.data_ptr(),
                                                 slabs.data_ptr(), nblk, Bn, H, W, Cin, Cout), "asis_conv3x3_smallcout_wgrad")
        if out is None:
            out = torch.empty((Cout, Cin, KH, KW), device=dy.device, dtype=torch.float32)
        reduce_rows(slabs, inv_scale, out.view(-1))
        return out
    if WGRAD_HALO and KH == 3 and KW == 3 and stride == 1 and pad == 1 and Cout % 64 == 0 and Cin % 128 == 0 and H >= 8 and W >= 16:
        # every 3x3 / stride-1 decoder stage: halo-tile kernel, pixels as the MFMA K index through transposing LDS reads, x fetched
        # once per tile instead of once per tap (csrc/convwgrad.hip; at 12 images: 128 -> 64 at 336^2 718 -> 286 us, 256 -> 128 at
        # 168^2 494 -> 244, 512 -> 256 at 84^2 329 -> 240, 3072 -> 512 at 42^2 982 -> 781: profiles/r05_wgrad_halo_ab.txt)
        nblk = lib().asis_conv3x3_wgrad_halo_nblk(Bn, H, W, Cin, Cout)
        slabs = torch.empty((nblk, Cout * Ntot), device=dy.device, dtype=torch.float32)
        check(lib().asis_conv3x3_wgrad_halo(_stream(), _dt(dy.dtype), dy.data_ptr(), CoP, x_nhwc.data_ptr(), slabs.data_ptr(), nblk,
                                            Bn, H, W, Cin, Cout), "asis_conv3x3_wgrad_halo")
        if out is None:
            out = torch.empty((Cout, Cin, KH, KW), device=dy.device, dtype=torch.float32)
        reduce_rows(slabs, inv_scale, out.view(-1))
        return out
    splits = lib().asis_wgrad_splits(P, Cout, Ntot)
    if KH == 1 and KW == 1 and stride == 1 and pad == 0 and Cout % 256 == 0 and Cin % 128 == 0 and CoP == Cout and P >= 1024:
        # nn.Linear weight gradients on the 256 x 128 LDS-DMA form (csrc/wgrad.hip: wgrad_dense_big_kernel, two workgroups per
        # CU): one round of the 512 workgroup slots, e.g. qkv 96 tiles x 5 pixel splits (6 would spill into a second round)
        splits = max(1, min(512 // ((Cout // 256) * (Cin // 128)), P // 256))
    elif stride == 1 and Cout % 256 == 0 and Cin % 128 == 0 and CoP == Cout and P >= 1024:
        # stride-1 convolutions on the same form (wgrad_dense_big_kernel<CONV>): (Cout / 256) x (taps * Cin / 128) tiles
        splits = max(1, min(512 // ((Cout // 256) * (Ntot // 128)), P // 1024))
    slabs = torch.empty((splits, Cout * Ntot), device=dy.device, dtype=torch.float32)
    d = _lib.WgradDesc()
    d.dy, d.x, d.out = dy.data_ptr(), x_nhwc.data_ptr(), slabs.data_ptr()
    d.ld_dy, d.P, d.dtype = CoP, P, _dt(dy.dtype)
    d.Cout, d.CoP, d.Cin = Cout, CoP, Cin
    d.B_, d.H, d.W, d.OH, d.OW, d.KH, d.KW, d.stride, d.pad = Bn, H, W, OH, OW, KH, KW, stride, pad
    d.splits = splits
    check(lib().asis_wgrad(_stream(), C.byref(d)), "asis_wgrad")
    if out is None:
        out = torch.empty((Cout, Cin, KH, KW), device=dy.device, dtype=torch.float32)
    reduce_rows(slabs, inv_scale, out.view(-1))
    return out


def grad_guard(g: torch.Tensor, guard: torch.Tensor, reset: bool) -> None:
    """guard int32[2]: guard[0] |= any non-finite element of ``g`` (cleared first when ``reset``)."""
    _dev(g, guard)
    if guard.dtype != torch.int32 or guard.numel() < 2:
        raise ValueError("guard must be an int32 tensor of 2 elements")
    check(lib().asis_grad_guard(_stream(), _f32c(g).data_ptr(), g.numel(), guard.data_ptr(), int(reset)), "asis_grad_guard")


def grad_pack_bf16(g: torch.Tensor, out: torch.Tensor) -> None:
    """fp32 gradient range -> its bf16 transport form (parallel.StageReducer(compress="bf16"))"""
    _dev(g, out)
    if out.dtype != torch.bfloat16 or out.numel() != g.numel() or not out.is_contiguous():
        raise ValueError("grad_pack_bf16: out must be a contiguous bfloat16 tensor of the same length")
    check(lib().asis_grad_pack_bf16(_stream(), _f32c(g).data_ptr(), g.numel(), out.data_ptr()), "asis_grad_pack_bf16")


def grad_unpack_bf16(src: torch.Tensor, g: torch.Tensor) -> None:
    _dev(g, src)
    if src.dtype != torch.bfloat16 or src.numel() != g.numel() or not src.is_contiguous():
        raise ValueError("grad_unpack_bf16: src must be a contiguous bfloat16 tensor of the same length")
    check(lib().asis_grad_unpack_bf16(_stream(), src.data_ptr(), g.numel(), _f32c(g).data_ptr()), "asis_grad_unpack_bf16")


def zeros(shape, device, dtype=torch.float32) -> torch.Tensor:
    """zero-filled device tensor through hipMemsetAsync (asis_zero) instead of an ATen fill kernel (165 us per 173 MB in the
    config-4 step against ~30 us): the few buffers of the step that kernels ACCUMULATE into or only partly write"""
    t = torch.empty(shape, device=device, dtype=dtype)
    if t.is_cuda:
        check(lib().asis_zero(_stream(), t.data_ptr(), t.numel() * t.element_size()), "asis_zero")
    else:
        t.zero_()
    return t


def sgd_momentum(p: torch.Tensor, g: torch.Tensor, buf: torch.Tensor, lr: float, momentum: float, weight_decay: float,
                 inv_scale: float, first_step: bool, guard: Optional[torch.Tensor] = None, count_skip: bool = True) -> None:
    _dev(p, g, buf)
    if guard is not None:
        _dev(guard)
        check(lib().asis_sgd_momentum_guarded(_stream(), _f32c(p).data_ptr(), _f32c(g).data_ptr(), _f32c(buf).data_ptr(),
                                              p.numel(), float(lr), float(momentum), float(weight_decay), float(inv_scale),
                                              int(first_step), guard.data_ptr(), int(count_skip)), "asis_sgd_momentum_guarded")
        return
    check(lib().asis_sgd_momentum(_stream(), _f32c(p).data_ptr(), _f32c(g).data_ptr(), _f32c(buf).data_ptr(), p.numel(),
                                  float(lr), float(momentum), float(weight_decay), float(inv_scale), int(first_step)),
          "asis_sgd_momentum")


# --------------------------------------------------------------------------------------------
# input pipeline
# --------------------------------------------------------------------------------------------
def augment(img_u8: torch.Tensor, mask_u8: torch.Tensor, tables: dict):
    """uint8 [B,S,S,3] / uint8 [B,S,S] + the per-sample tables of ``tools.augment.TrainAugment.tables`` ->
    (fp32 [B,3,S,S] in [0,1], int64 [B,S,S])."""
    _dev(img_u8, mask_u8, *[t for t in tables.values() if torch.is_tensor(t)])
    B, S = img_u8.shape[0], img_u8.shape[1]
    if img_u8.dtype != torch.uint8 or mask_u8.dtype != torch.uint8 or img_u8.shape != (B, S, S, 3) or mask_u8.shape != (B, S, S) \
            or not img_u8.is_contiguous() or not mask_u8.is_contiguous():
        raise ValueError("augment: img uint8 [B,S,S,3] and mask uint8 [B,S,S], contiguous")
    want = {"geo": (torch.int32, (B, 4)), "xofs": (torch.int32, (B, S)), "yofs": (torch.int32, (B, S)),
            "xa": (torch.int16, (B, S, 2)), "ya": (torch.int16, (B, S, 2)), "mx": (torch.int32, (B, S)),
            "my": (torch.int32, (B, S)), "lut": (torch.uint8, (B, 256))}
    for k, (dt, shp) in want.items():
        t = tables[k]
        if t.dtype != dt or tuple(t.shape) != shp or not t.is_contiguous():
            raise ValueError(f"augment: table {k} must be contiguous {dt} {shp}")
    out = torch.empty((B, 3, S, S), device=img_u8.device, dtype=torch.float32)
    mout = torch.empty((B, S, S), device=img_u8.device, dtype=torch.int64)
    cl = tables.get("clahe")
    if cl is not None and tables.get("clahe_any", True):
        # CLAHE samples in the batch (train.py:161): geometric stage -> uint8, tile LUTs, blend + Lab -> RGB + tables + / 255
        from .tools import clahe as _cl
        if cl.dtype != torch.int32 or tuple(cl.shape) != (B, 2) or not cl.is_contiguous():
            raise ValueError("augment: table clahe must be contiguous int32 (B, 2)")
        _dev(cl)
        lt = _cl.lab_tables(img_u8.device)
        geo8 = torch.empty((B, S, S, 3), device=img_u8.device, dtype=torch.uint8)
        check(lib().asis_augment_geo_u8(_stream(), img_u8.data_ptr(), mask_u8.data_ptr(), tables["geo"].data_ptr(),
                                        tables["xofs"].data_ptr(), tables["yofs"].data_ptr(), tables["xa"].data_ptr(),
                                        tables["ya"].data_ptr(), tables["mx"].data_ptr(), tables["my"].data_ptr(),
                                        geo8.data_ptr(), mout.data_ptr(), B, S), "asis_augment_geo_u8")
        luts = torch.empty((B, _cl.TILES, _cl.TILES, 256), device=img_u8.device, dtype=torch.uint8)
        check(lib().asis_clahe(_stream(), geo8.data_ptr(), cl.data_ptr(), lt["gamma"].data_ptr(), lt["cbrt"].data_ptr(),
                               lt["l2yf"].data_ptr(), lt["ab2xz"].data_ptr(), lt["invgamma"].data_ptr(),
                               lt["fwd"].ctypes.data, lt["inv"].ctypes.data, luts.data_ptr(), tables["lut"].data_ptr(),
                               out.data_ptr(), B, S, _cl.TILES), "asis_clahe")
        return out, mout
    check(lib().asis_augment(_stream(), img_u8.data_ptr(), mask_u8.data_ptr(), tables["geo"].data_ptr(),
                             tables["xofs"].data_ptr(), tables["yofs"].data_ptr(), tables["xa"].data_ptr(),
                             tables["ya"].data_ptr(), tables["mx"].data_ptr(), tables["my"].data_ptr(), tables["lut"].data_ptr(),
                             out.data_ptr(), mout.data_ptr(), B, S), "asis_augment")
    return out, mout


# ---- dropout of the MaskTransformer head (csrc/dropout.hip: counter-based masks, include/asis_hip.h) -----------------------------
def dropout_f32(x: torch.Tensor, seed: int, site: int, p: float, res: Optional[torch.Tensor] = None, alpha: float = 1.0,
                bias_n: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 contiguous [..., C]: out = (res) + (alpha * x + bias_n) * keep / (1 - p)."""
    _dev(x, res, bias_n, out)
    if x.dtype != torch.float32 or not x.is_contiguous() or (res is not None and (res.shape != x.shape or not res.is_contiguous())):
        raise ValueError("dropout_f32: contiguous float32 operands of one shape expected")
    if out is None:
        out = torch.empty_like(x)
    check(lib().asis_dropout_f32(_stream(), x.data_ptr(), _p(res), out.data_ptr(), x.numel(), int(seed), int(site), float(p),
                                 float(alpha), _p(_f32c(bias_n)), int(x.shape[-1])), "asis_dropout_f32")
    return out


def dropout_t16(x: torch.Tensor, seed: int, site: int, p: float, x_lo: Optional[torch.Tensor] = None, rescale: bool = True) -> torch.Tensor:
    """16-bit contiguous, IN PLACE: x *= keep (/ (1 - p) with ``rescale``); ``x_lo`` (split-operand residual half) zeroed alike."""
    _dev(x, x_lo)
    if not x.is_contiguous() or (x_lo is not None and (x_lo.shape != x.shape or not x_lo.is_contiguous() or x_lo.dtype != x.dtype)):
        raise ValueError("dropout_t16: contiguous 16-bit operands of one shape expected")
    check(lib().asis_dropout_t16(_stream(), _dt(x.dtype), x.data_ptr(), _p(x_lo), x.numel(), int(seed), int(site), float(p),
                                 int(bool(rescale))), "asis_dropout_t16")
    return x


def dropout_mask(n: int, seed: int, site: int, p: float, device) -> torch.Tensor:
    """uint8 [n] keep flags of dropout layer ``site`` (test infrastructure replays them in the oracle)."""
    out = torch.empty((n,), device=device, dtype=torch.uint8)
    check(lib().asis_dropout_mask(_stream(), out.data_ptr(), n, int(seed), int(site), float(p)), "asis_dropout_mask")
    return out


def softmax_dropout_fwd(S: torch.Tensor, N: int, scale: float, seed: int, site: int, p: float, dtype: torch.dtype):
    """S fp32 [rows, ld] -> (P, P * keep / (1 - p)) 16-bit [rows, ld], P = softmax(scale * S[:, :N]), padding columns 0."""
    _dev(S)
    rows, ld = S.shape
    if S.dtype != torch.float32 or not S.is_contiguous():
        raise ValueError("softmax_dropout_fwd: contiguous float32 [rows, ld] expected")
    p16 = torch.empty((rows, ld), device=S.device, dtype=dtype)
    pd16 = torch.empty_like(p16)
    check(lib().asis_softmax_dropout_fwd(_stream(), _dt(dtype), S.data_ptr(), p16.data_ptr(), pd16.data_ptr(), rows, int(N), ld,
                                         float(scale), int(seed), int(site), float(p)), "asis_softmax_dropout_fwd")
    return p16, pd16


def softmax_dropout_bwd(p16: torch.Tensor, pd16: torch.Tensor, dPd: torch.Tensor, N: int, scale: float, seed: int, site: int,
                        p: float) -> torch.Tensor:
    """-> dS 16-bit [rows, ld] = scale * P * (keep / (1 - p) * dPd - rowsum(Pd * dPd))."""
    _dev(p16, pd16, dPd)
    rows, ld = p16.shape
    if dPd.dtype != torch.float32 or dPd.shape != p16.shape or not (p16.is_contiguous() and pd16.is_contiguous() and dPd.is_contiguous()):
        raise ValueError("softmax_dropout_bwd: contiguous [rows, ld] operands expected")
    ds = torch.empty_like(p16)
    check(lib().asis_softmax_dropout_bwd(_stream(), _dt(p16.dtype), p16.data_ptr(), pd16.data_ptr(), dPd.data_ptr(), ds.data_ptr(),
                                         rows, int(N), ld, float(scale), int(seed), int(site), float(p)), "asis_softmax_dropout_bwd")
    return ds
