"""CPU: structural rules of the build — the product never touches the oracle, never falls back to the CPU,
and keeps the reference's parameter names."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def product_sources():
    for base, _, files in os.walk(os.path.join(ROOT, "adaptersis_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                yield os.path.join(base, f)


def test_product_never_imports_the_oracle():
    bad = []
    for path in product_sources():
        src = open(path).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "ref_torch" in src:
            bad.append(path)
    assert not bad, bad


def test_ops_refuse_cpu_tensors():
    from adaptersis_amd import ops
    a = torch.zeros(8, 8, dtype=torch.float16)
    with pytest.raises(Exception, match="no CPU fallback"):
        ops.gemm(a, a)
    with pytest.raises(Exception, match="no CPU fallback"):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_modules_refuse_cpu_inputs():
    from adaptersis_amd.dinov2.models import vision_transformer as vits
    m = vits.vit_tiny_test(img_size=518, init_values=1e-5)
    with pytest.raises(Exception):
        m.patch_embed(torch.zeros(1, 3, 28, 28))


def test_state_dict_keys_match_the_reference_abi():
    """Key names are the persistence ABI."""
    from adaptersis_amd.backbones.adapter_blocks import CACNN, CAViT
    from adaptersis_amd.backbones.decoders import FeatureDecoder
    from adaptersis_amd.backbones.encoders import FeatureEncoder
    from adaptersis_amd.dinov2.models import vision_transformer as vits
    from adaptersis_amd.utils import weights as W
    for arch in ("vit_small", "vit_giant2"):
        D, depth, heads, ffn = W.VIT_CONFIGS[arch]
        with torch.device("meta"):  # key names only: no 1.1 G-parameter initialisation on the CPU
            m = vits.__dict__[arch](patch_size=14, img_size=518, init_values=1e-5, ffn_layer=ffn, block_chunks=0)
        sd = W.make_vit_state_dict(arch) if arch == "vit_small" else None
        keys = set(m.state_dict().keys())
        assert {"cls_token", "pos_embed", "mask_token", "patch_embed.proj.weight", "blocks.0.norm1.weight",
                "blocks.0.attn.qkv.weight", "blocks.0.attn.proj.bias", "blocks.0.ls1.gamma", "norm.weight"} <= keys
        assert ("blocks.0.mlp.fc1.weight" in keys) == (ffn == "mlp")
        assert ("blocks.0.mlp.w12.weight" in keys) == (ffn != "mlp")
        if sd is not None:
            assert set(sd.keys()) == keys
    assert set(FeatureEncoder(embed_dim=64).state_dict()) == set(W.make_encoder_state_dict(64))
    assert set(CAViT(dim=64, n_levels=3, num_heads=8).state_dict()) == set(W.make_cavit_state_dict(64))
    assert set(CACNN(dim=64, n_levels=1, num_heads=8).state_dict()) == set(W.make_cacnn_state_dict(64))
    assert set(FeatureDecoder(embed_dim=64, features=[64, 32, 16, 16, 8]).state_dict()) == \
        set(W.make_feature_decoder_state_dict(64, 2, (64, 32, 16, 16, 8)))


def test_reference_error_conventions():
    from adaptersis_amd.backbones.ops.modules import MSDeformAttn
    with pytest.raises(ValueError, match="divisible"):
        MSDeformAttn(d_model=100, n_heads=8)
    from adaptersis_amd.dinov2.layers import Attention
    with pytest.raises(ValueError):
        Attention(dim=96, num_heads=2)  # head dim 48: every DINOv2 arch has 64
