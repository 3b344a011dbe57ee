"""Mirror of `dinov2/models/__init__.py:14-40` (``build_model`` / ``build_model_from_cfg``) for the
teacher/eval model the AdapterSIS path builds (`dinov2/eval/setup.py:62-75`)."""
from . import vision_transformer as vits


def build_model(args, only_teacher=False, img_size=224):
    arch = args.arch.removesuffix("_memeff")
    if "vit" not in arch:
        raise ValueError(f"unsupported arch {arch}")
    vit_kwargs = dict(img_size=img_size, patch_size=args.patch_size, init_values=args.layerscale,
                      ffn_layer=args.ffn_layer, block_chunks=args.block_chunks, qkv_bias=args.qkv_bias,
                      proj_bias=args.proj_bias, ffn_bias=args.ffn_bias)
    teacher = vits.__dict__[arch](**vit_kwargs)
    if only_teacher:
        return teacher, teacher.embed_dim
    # the student only differs by drop_path, which is a pre-training feature (out of scope): same model
    student = vits.__dict__[arch](**vit_kwargs)
    return student, teacher, student.embed_dim


def build_model_from_cfg(cfg, only_teacher=False):
    return build_model(cfg.student, only_teacher=only_teacher, img_size=cfg.crops.global_crops_size)
