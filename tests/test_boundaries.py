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


def test_mx_plane_tag_is_checked_not_assumed():
    """an MX-form lo plane (two fp8 bytes per element) must never be read as a 16-bit residual: the type survives views and
    copies, the absolute maximum does not, and every consumer asks through ops.mx_amax_of / ops._not_mx"""
    from adaptersis_amd import ops
    lo = torch.zeros(2, 4, 4, 8, dtype=torch.float16)
    amax = torch.ones(1)
    assert ops.mx_amax_of(None) is None and ops.mx_amax_of(lo) is None          # a plain residual
    mx = ops.MxPlane.tag(lo, amax)
    assert ops.mx_amax_of(mx) is amax and mx.data_ptr() == lo.data_ptr()
    for dropped in (mx[:1], mx.clone(), mx.reshape(2, 16, 8), mx.contiguous()[0:1]):
        assert isinstance(dropped, ops.MxPlane)
        with pytest.raises(ValueError, match="without its absolute maximum"):
            ops.mx_amax_of(dropped)
    with pytest.raises(ValueError, match="16-bit rounding residual is expected"):
        ops._not_mx("dilate2", mx)
    ops._not_mx("dilate2", lo, None)


def test_layerscale_pow2_is_not_a_host_sync_per_step():
    """Block._ls_pow2 reads max|gamma| (a device-to-host sync) once for a frozen gamma and once per _LS_POW2_REFRESH changes
    for a trained one; packs scaled by a retired k are evicted"""
    from adaptersis_amd.dinov2.layers import blocks as B
    blk = B.Block(64, 1, qkv_bias=True, init_values=1e-5)
    g = blk.ls1.gamma
    assert blk._ls_pow2("ls1.k", g) == 17 and blk._ls_pow2("ls1.k", None) == 0      # 1e-5 * 2^17 = 1.31
    blk._cache["projT@17"] = ("x", None); blk._cache["n1w@-17"] = ("x", None); blk._cache["qkvT"] = ("x", None)
    for i in range(B._LS_POW2_REFRESH - 1):
        with torch.no_grad():
            g.mul_(1.0 + 1e-3)
        assert blk._ls_pow2("ls1.k", g) == 17, i
        assert blk._ls_pow2("ls1.k", g) == 17                                        # same generation: no counting
    with torch.no_grad():
        g.fill_(3e-4)                                                                # 3e-4 * 2^12 = 1.23
    assert blk._ls_pow2("ls1.k", g) == 12
    assert "projT@17" not in blk._cache and "n1w@-17" not in blk._cache and "qkvT" in blk._cache
    with torch.no_grad():
        g.zero_()
    for _ in range(B._LS_POW2_REFRESH):
        g._asis_gen = getattr(g, "_asis_gen", 0) + 1                                 # the fused optimizer's change marker
        k = blk._ls_pow2("ls1.k", g)
    assert k == 0


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


def test_dinov2_checkpoint_loader_prefixes_and_key(tmp_path):
    """`dinov2/utils/utils.py:20-33`: ``teacher``-keyed files, ``module.`` / ``backbone.`` prefixes, extra head keys
    (strict=False), plain state dicts; URLs are refused (no network on this path)."""
    import pytest
    import torch
    from adaptersis_amd.dinov2.models import vision_transformer as vits
    from adaptersis_amd.utils import misc, weights as W
    sd = W.make_vit_state_dict("vit_tiny_test", layerscale="kernel")
    model = vits.vit_tiny_test(img_size=518, init_values=1e-5, block_chunks=0)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    ck = {"teacher": {**{"module.backbone." + k: v for k, v in sd.items()}, "module.dino_head.mlp.0.weight": torch.zeros(3, 3)},
          "student": {"backbone." + k: torch.zeros_like(v) for k, v in sd.items()}, "epoch": 7}
    path = tmp_path / "dinov2_tiny.pth"
    torch.save(ck, path)
    msg = misc.load_pretrained_weights(model, str(path), "teacher")
    assert list(msg.missing_keys) == [] and list(msg.unexpected_keys) == ["dino_head.mlp.0.weight"]
    after = model.state_dict()
    assert all(torch.equal(after[k], sd[k]) for k in sd) and any(not torch.equal(after[k], before[k]) for k in sd)
    # a plain (un-keyed, un-prefixed) state dict loads as is; a wrong key falls back to the whole file
    model2 = vits.vit_tiny_test(img_size=518, init_values=1e-5, block_chunks=0)
    torch.save(sd, tmp_path / "plain.pth")
    msg = misc.load_pretrained_weights(model2, str(tmp_path / "plain.pth"), "teacher")
    assert not msg.missing_keys and not msg.unexpected_keys
    assert torch.equal(model2.state_dict()["blocks.3.mlp.fc2.weight"], sd["blocks.3.mlp.fc2.weight"])
    with pytest.raises(ValueError):
        misc.load_pretrained_weights(model2, "https://example.invalid/dinov2_vitl14_pretrain.pth", "teacher")


def test_restart_skips_reference_format_optimizer_entry(tmp_path, capsys):
    """a checkpoint written by the reference holds a per-parameter ``torch.optim.SGD`` state; loading it into the
    flat-bucket optimizer must be reported and skipped, not crash the run — and must leave the momentum untouched."""
    import torch
    from adaptersis_amd import optim
    from adaptersis_amd.utils import misc
    lin = torch.nn.Linear(4, 3)
    ref_opt = torch.optim.SGD(lin.parameters(), lr=0.1, momentum=0.9)
    lin(torch.ones(2, 4)).sum().backward()
    ref_opt.step()
    torch.save({"optimizer": ref_opt.state_dict(), "epoch": 3}, tmp_path / "checkpoint.pth.tar")
    lin2 = torch.nn.Linear(4, 3)
    bucket = optim.FlatBucket(list(lin2.named_parameters()))
    opt = optim.SGD([bucket], lr=0.1, momentum=0.9)
    bucket.momentum.fill_(0.25)
    rv = {"epoch": 0}
    misc.restart_from_checkpoint(str(tmp_path / "checkpoint.pth.tar"), run_variables=rv, optimizer=opt)
    assert rv["epoch"] == 3 and float(bucket.momentum.min()) == 0.25
    assert "failed to load 'optimizer'" in capsys.readouterr().out
    # its own format round-trips
    torch.save({"optimizer": opt.state_dict()}, tmp_path / "own.pth.tar")
    bucket.momentum.zero_()
    misc.restart_from_checkpoint(str(tmp_path / "own.pth.tar"), optimizer=opt)
    assert float(bucket.momentum.min()) == 0.25
