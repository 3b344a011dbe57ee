"""ctypes binding of libasis_hip.so (the C ABI declared in include/asis_hip.h).

The library is the product: if it is missing or fails to load, every op raises — there is no
eager-PyTorch or CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libasis_hip.so")

ASIS_F16, ASIS_BF16, ASIS_F32 = 0, 1, 2
ASIS_EINVAL, ASIS_ELAUNCH = -1, -2
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU_MUL, ACT_GELU_GRAD = 0, 1, 2, 3, 4


class AsisError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("strideA", C.c_int64), ("strideB", C.c_int64), ("strideC", C.c_int64),
        ("batch", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("bias_n", C.c_void_p), ("bias_m", C.c_void_p), ("scale_n", C.c_void_p), ("res", C.c_void_p),
        ("ldr", C.c_int64), ("strideR", C.c_int64),
        ("act", C.c_int32), ("out_f32", C.c_int32), ("dtype", C.c_int32),
        ("conv", C.c_int32),
        ("B_", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("OH", C.c_int32),
        ("OW", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("stats", C.c_void_p),
        ("A_lo", C.c_void_p), ("B_lo", C.c_void_p),
        ("aux", C.c_void_p), ("ld_aux", C.c_int64),
        ("ksplit", C.c_int32),
        ("mx_amax_a", C.c_void_p), ("mx_amax_b", C.c_void_p),
        ("C_lo", C.c_void_p), ("rowstats", C.c_void_p), ("res16", C.c_void_p), ("res16_lo", C.c_void_p), ("ldr16", C.c_int64),
        ("ln_mr", C.c_void_p), ("ln_cs", C.c_void_p), ("ln_cols", C.c_int32),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("x", C.c_void_p), ("out", C.c_void_p),
        ("ld_dy", C.c_int64), ("P", C.c_int64),
        ("dtype", C.c_int32), ("Cout", C.c_int32), ("CoP", C.c_int32), ("Cin", C.c_int32),
        ("B_", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("splits", C.c_int32), ("k_per_split", C.c_int64),
    ]


_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsisError(
            f"{LIB_PATH} is missing: build it with `python -m adaptersis_amd.build` "
            "(hipcc --offload-arch=gfx950). adaptersis_amd has no fallback path.")
    _lib = C.CDLL(LIB_PATH)
    _lib.asis_last_error.restype = C.c_char_p
    _declare(_lib)
    return _lib


# name -> argtypes ; every function returns int
_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
SIGNATURES = {
    "asis_version": [],
    "asis_device_count": [],
    "asis_gemm": [_vp, C.POINTER(GemmDesc)],
    "asis_gemm_tiles_m": [_i],
    "asis_absmax_f32": [_vp, _vp, _i64, _i, _i64, _vp, _i],
    "asis_bn_relu_absmax": [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "asis_absmax_16": [_vp, _i, _vp, _i64, _i, _i64, _vp],
    "asis_conv3x3_halo_mx_tiles": [_i, _i, _i],
    "asis_conv3x3_halo_mx": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asis_mx_from_pair": [_vp, _i, _vp, _vp, _i64, _vp, _i64, _i64, _i, _vp, _i],
    "asis_bn_relu_upsample_mx": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asis_pack_conv_weight_mx": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i64, _vp],
    "asis_pack_conv_weight_pair": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _vp],
    "asis_decoder_input_mx": [_vp, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_ln_stats_finalize": [_vp, _vp, _i64, _i, _i, _f, _vp],
    "asis_split_stats": [_vp, _i, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _f],
    "asis_gemm_set_option": [C.c_char_p, _i],
    "asis_layernorm": [_vp, _i, _vp, _i64, _vp, _vp, _f, _vp, _i64, _i, _i64, _i],
    "asis_layernorm_mx": [_vp, _i, _vp, _i64, _vp, _vp, _f, _vp, _vp, _i64, _vp, _i64, _i],
    "asis_attention_fwd_seg": [_vp, _i, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _f, _vp],
    "asis_attention_fwd_split": [_vp, _i, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _vp],
    "asis_attention_fwd_prescaled": [_vp, _i, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp],
    "asis_attention_fwd_qkv": [_vp, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _i, _vp],
    "asis_attention_fwd_qkv_mx": [_vp, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _i, _vp, _vp],
    "asis_attention_fwd_lse": [_vp, _i, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _f, _vp],
    "asis_transpose_tokens": [_vp, _i, _vp, _i64, _vp, _i64, _i, _i, _i],
    "asis_attention_bwd_rows": [_vp, _i, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i,
                                _i, _f],
    "asis_msda_bwd": [_vp, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "asis_msda_vgrad_cap": [_i, _i, _i],
    "asis_msda_value_grad": [_vp, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "asis_msda_sampling_matrix": [_vp, _i, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i],
    "asis_dwconv_bwd_nblk": [_i64],
    "asis_dwconv_gelu_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i],
    "asis_rowblock_nblk": [_i64],
    "asis_layernorm_bwd": [_vp, _vp, _i64, _vp, _i64, _vp, _f, _vp, _i64, _vp, _i64, _vp, _i64, _i],
    "asis_gelu16": [_vp, _i, _vp, _vp, _vp, _i64],
    "asis_gelu_split": [_vp, _i, _vp, _vp, _vp, _i64],
    "asis_cast_colsum": [_vp, _i, _vp, _i64, _vp, _i64, _f, _vp, _i64, _i],
    "asis_swiglu_bwd": [_vp, _i, _vp, _vp, _vp, _i64, _i],
    "asis_colsum": [_vp, _i, _vp, _i64, _vp, _i64, _i],
    "asis_ls_linear_finish": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i],
    "asis_attention_fwd": [_vp, _i, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _f],
    "asis_im2col_patch": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i64],
    "asis_im2col_patch_split": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i64],
    "asis_cast_pad": [_vp, _i, _vp, _i64, _vp, _i64, _i64, _i, _f, _i],
    "asis_add_cls_pos": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "asis_msda_fwd": [_vp, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "asis_msda_fwd_split": [_vp, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "asis_dwconv_gelu": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i],
    "asis_conv3x3_c3": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_conv3x3_smallcout_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asis_conv3x3_smallcout_fwd_up": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asis_conv3x3_smallcout_wgrad_up": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_conv3x3_smallcout_dgrad": [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i],
    "asis_conv3x3_smallcout_wgrad": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_colstats_nparts": [_i64],
    "asis_colstats": [_vp, _vp, _i64, _i, _vp],
    "asis_reduce_partials": [_vp, _vp, _i, _i, _vp],
    "asis_bn_finalize": [_vp, _vp, _d, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "asis_bn_eval_affine": [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp],
    "asis_bn_act": [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i],
    "asis_bn_act_mx": [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i64, _i],
    "asis_bn_relu_maxpool": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_bn_relu_upsample": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asis_pack_conv_weight": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i64, _i],
    "asis_decoder_input": [_vp, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_swiglu": [_vp, _i, _vp, _vp, _i64, _i],
    "asis_swiglu_split": [_vp, _i, _vp, _vp, _vp, _i64, _i],
    "asis_copy_channels": [_vp, _vp, _i64, _vp, _i64, _i64, _i64],
    "asis_add_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i64, _i64, _i64],
    "asis_maxpool2_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_maxpool2_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_cls_l2norm": [_vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_cls_l2norm_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_mask_logits_fwd": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_mask_logits_nblk": [_i, _i],
    "asis_mask_logits_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_mask_dchat": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_nearest_add_relu": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_nearest_sum": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_convt2x2_scatter": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "asis_convt2x2_gather": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "asis_convt2x2_bias_nblk": [_i64],
    "asis_conv1x1_dgrad_small": [_vp, _i, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i],
    "asis_convt2x2_bias_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "asis_dice_nblk": [_i, _i],
    "asis_seg_loss_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _vp, _vp, _vp, _vp],
    "asis_seg_loss_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "asis_dice_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _vp, _vp, _vp, _vp],
    "asis_dice_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "asis_ce_acc_nblk": [_i64],
    "asis_ce_acc_counts": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "asis_ce_acc": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "asis_resize_bilinear_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "asis_resize_bwd_nblk": [_i64],
    "asis_resize_bilinear_bwd": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "asis_reduce_rows": [_vp, _vp, _i, _i, _f, _vp],
    "asis_ew_blocks": [_i64],
    "asis_bn_bwd_nblk": [_i64, _i],
    "asis_maxpool_bn_relu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asis_dilate2": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_upsample_bn_relu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asis_bn_bwd_apply": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i64, _i],
    "asis_bn_bwd_apply_mx": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _i64, _i],
    "asis_bn_bwd_absmax": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _i64, _i],
    "asis_wgrad_splits": [_i64, _i, _i],
    "asis_wgrad": [_vp, C.POINTER(WgradDesc)],
    "asis_conv3x3_wgrad_halo_nblk": [_i, _i, _i, _i, _i],
    "asis_conv3x3_wgrad_halo": [_vp, _i, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asis_sgd_momentum": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i],
    "asis_augment": [_vp] * 13 + [_i, _i],
    "asis_augment_geo_u8": [_vp] * 12 + [_i, _i],
    "asis_clahe": [_vp] * 13 + [_i, _i, _i],
    "asis_dropout_f32": [_vp, _vp, _vp, _vp, _i64, C.c_uint64, _i, _f, _f, _vp, _i],
    "asis_dropout_t16": [_vp, _i, _vp, _vp, _i64, C.c_uint64, _i, _f, _i],
    "asis_dropout_mask": [_vp, _vp, _i64, C.c_uint64, _i, _f],
    "asis_softmax_dropout_fwd": [_vp, _i, _vp, _vp, _vp, _i64, _i, _i, _f, C.c_uint64, _i, _f],
    "asis_softmax_dropout_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, C.c_uint64, _i, _f],
    "asis_grad_guard": [_vp, _vp, _i64, _vp, _i],
    "asis_sgd_momentum_guarded": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _vp, _i],
    "asis_scale_f32": [_vp, _vp, _i64, _f],
    "asis_zero": [_vp, _vp, _i64],
    "asis_grad_pack_bf16": [_vp, _vp, _i64, _vp],
    "asis_grad_unpack_bf16": [_vp, _vp, _i64, _vp],
}


def _declare(l: C.CDLL) -> None:
    for name, argtypes in SIGNATURES.items():
        fn = getattr(l, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = C.c_int


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = lib().asis_last_error().decode(errors="replace")
    if rc == -1:
        raise ValueError(f"{what}: {msg}" if what else msg)
    raise AsisError(f"{what}: {msg}" if what else msg)
