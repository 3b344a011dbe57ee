"""Tensor-level wrappers over the C ABI (include/asis_hip.h).

torch is used here only as the owner of device memory and of the current HIP stream: every
function takes CUDA(=HIP) tensors, passes raw pointers/sizes to libasis_hip.so and returns
tensors allocated with ``torch.empty``.  A CPU tensor, a missing library or a bad shape raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, GemmDesc, check, lib  # noqa: F401

T16_DEFAULT = torch.float16


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


def gemm(a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None, out_f32: bool = False,
         bias_n: Optional[torch.Tensor] = None, bias_m: Optional[torch.Tensor] = None,
         scale_n: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out = epilogue(a @ b.T)``; a [M,K] or [batch,M,K], b [N,K] or [batch,N,K] (16-bit, K contiguous).

    A 2-D operand next to a 3-D one is shared by every batch element.  Row strides may exceed K.
    """
    _dev(a, b, out, bias_n, bias_m, scale_n, res, stats)
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
    if out is None:
        shape = (batch, M, N) if (a.dim() == 3 or b.dim() == 3) else (M, N)
        out = torch.empty(shape, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    else:
        out_f32 = out.dtype == torch.float32
        if out.stride(-1) != 1:
            raise ValueError("gemm: out must be contiguous in its last dim")
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
    check(lib().asis_gemm(_stream(), C.byref(d)), "asis_gemm")
    return out


def gemm_tiles_m(M: int) -> int:
    return lib().asis_gemm_tiles_m(int(M))


def conv_gemm(x_nhwc: torch.Tensor, w_packed: torch.Tensor, KH: int, KW: int, stride: int, pad: int, *,
              out: Optional[torch.Tensor] = None, out_f32: bool = True, bias_n: Optional[torch.Tensor] = None,
              act: int = ACT_NONE, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Implicit-GEMM convolution: x [B,H,W,Cin] (16-bit NHWC), w_packed [Cout, KH*KW*Cin] ->
    out [B,OH,OW,Cout] (fp32 by default: BatchNorm statistics are taken on it)."""
    _dev(x_nhwc, w_packed, out, bias_n, stats)
    if not x_nhwc.is_contiguous():
        raise ValueError("conv_gemm: x must be contiguous NHWC")
    Bn, H, W, Cin = x_nhwc.shape
    Cout, K = w_packed.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((Bn, OH, OW, Cout), device=x_nhwc.device, dtype=torch.float32 if out_f32 else x_nhwc.dtype)
    d = GemmDesc()
    d.A, d.B, d.C = x_nhwc.data_ptr(), w_packed.data_ptr(), out.data_ptr()
    d.lda, d.ldb, d.ldc = K, w_packed.stride(0), Cout
    d.batch, d.M, d.N, d.K = 1, Bn * OH * OW, Cout, K
    d.bias_n = _p(_f32c(bias_n))
    d.act, d.out_f32, d.dtype = act, int(out.dtype == torch.float32), _dt(x_nhwc.dtype)
    d.conv, d.B_, d.H, d.W, d.Cin, d.OH, d.OW = 1, Bn, H, W, Cin, OH, OW
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.stats = _p(stats)
    check(lib().asis_gemm(_stream(), C.byref(d)), "asis_gemm(conv)")
    return out


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


def attention_fwd(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, B: int, H: int, N: int, scale: float,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q, k: [B*N, >=H*64] views (same row stride); vt: [B, H*64, ldvt]; returns o [B*N, H*64]."""
    _dev(q, k, vt, out)
    if q.stride(0) != k.stride(0) or q.stride(1) != 1 or k.stride(1) != 1:
        raise ValueError("attention_fwd: q and k must share a row stride and be contiguous in the last dim")
    if out is None:
        out = torch.empty((B * N, H * 64), device=q.device, dtype=q.dtype)
    check(lib().asis_attention_fwd(_stream(), _dt(q.dtype), q.data_ptr(), k.data_ptr(), q.stride(0), vt.data_ptr(),
                                   vt.stride(1), out.data_ptr(), out.stride(0), B, H, N, float(scale)),
          "asis_attention_fwd")
    return out


def im2col_patch(img: torch.Tensor, P: int, ldk: int, dtype: torch.dtype = T16_DEFAULT) -> torch.Tensor:
    _dev(img)
    if img.dtype != torch.float32 or not img.is_contiguous() or img.dim() != 4 or img.shape[1] != 3:
        raise ValueError("im2col_patch: img must be contiguous float32 [B,3,H,W]")
    B, _, Hi, Wi = img.shape
    if Hi % P or Wi % P:
        # error text mirrors dinov2/layers/patch_embed.py:72-73
        raise AssertionError(f"Input image height {Hi} / width {Wi} is not a multiple of patch size {P}")
    out = torch.empty((B * (Hi // P) * (Wi // P), ldk), device=img.device, dtype=dtype)
    check(lib().asis_im2col_patch(_stream(), _dt(dtype), img.data_ptr(), B, Hi, Wi, P, out.data_ptr(), ldk),
          "asis_im2col_patch")
    return out


def cast_pad(src: torch.Tensor, ld_dst: Optional[int] = None, dtype: torch.dtype = T16_DEFAULT) -> torch.Tensor:
    """float32 [rows, cols] -> 16-bit [rows, ld_dst] with zero-filled pad columns (weight packing)."""
    _dev(src)
    if src.dtype != torch.float32 or src.dim() != 2 or src.stride(1) != 1:
        raise ValueError("cast_pad: src must be float32 [rows, cols] contiguous in cols")
    rows, cols = src.shape
    ld = ld_dst if ld_dst is not None else (cols + 7) // 8 * 8
    out = torch.empty((rows, ld), device=src.device, dtype=dtype)
    check(lib().asis_cast_pad(_stream(), _dt(dtype), src.data_ptr(), src.stride(0), out.data_ptr(), ld, rows, cols),
          "asis_cast_pad")
    return out


def add_cls_pos(x: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """x [B,N,D] f32, cls [D], pos [N+1,D] -> [B,N+1,D] (vision_transformer.py:196-197)."""
    _dev(x, cls, pos)
    B, N, D = x.shape
    out = torch.empty((B, N + 1, D), device=x.device, dtype=torch.float32)
    check(lib().asis_add_cls_pos(_stream(), _f32c(x).data_ptr(), _f32c(cls).data_ptr(), _f32c(pos).data_ptr(),
                                 out.data_ptr(), B, N, D), "asis_add_cls_pos")
    return out
