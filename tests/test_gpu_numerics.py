"""GPU: numerics modes and operand-range stress of the whole step.

* bf16 operand mode (north_star names "MFMA bf16"; DESIGN.md §3: fp16 is the default because bf16's 8-bit mantissa puts
  ~8x the rounding error on every GEMM operand): the whole ViT-L/14 588x588 step in ``ASIS_OPERAND=bf16`` against the
  reference goldens — holds 1e-3 on the reference configuration, not on the stress weights (bound stated below).
* f16 operand RANGE: DINOv2 checkpoints carry "massive activation" channels (residual outliers of 1e2-1e3 out of fc2 of
  a middle block, large fc1 pre-activations, saturated attention rows); there is no network to fetch real weights, so they are
  injected into the synthetic ones: every 16-bit operand (LayerNorm outputs, q/k/v/P, GELU outputs) must stay finite and
  the step within 1e-3 of the fp32 oracle."""
import pytest
import torch

from adaptersis_amd import config
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import golden_err, load_golden, rel_l2
from tests.test_gpu_step import build_engine

pytestmark = pytest.mark.gpu
BF16_L2_BOUND = 1e-3      # north_star's tolerance; round 4 measured 1.51e-3 here (the adapters' MSDA layers were single-bf16: 1.43e-3 by themselves)


@pytest.fixture
def bf16_mode():
    old_dt, old_ls = config.operand_dtype, config.loss_scale
    config.set_operand_dtype(torch.bfloat16)
    config.loss_scale = 1.0
    yield
    config.set_operand_dtype(old_dt)
    config.loss_scale = old_ls


@pytest.mark.parametrize("mode,tag,bound", [("init", "step_exact", 1e-3), ("kernel", "step_kernel", 9e-3)])   # measured 2.0e-5 / 7.5e-3
def test_bf16_operand_mode_whole_step_vs_reference_golden(dev, bf16_mode, mode, tag, bound):
    g = load_golden("step")
    eng, _ = build_engine("vit_large", mode, dev)
    img, tgt = W.synthetic_batch(1, 588)
    taps = {}
    loss = eng.train_step(img.to(dev), tgt.to(dev), taps)
    e_cat = golden_err(taps["cat"].float().permute(0, 3, 1, 2), g[f"{tag}.cat"])
    e_lg = golden_err(taps["logits"].permute(0, 3, 1, 2), g[f"{tag}.logits"])
    print(f"bf16 operands, {tag}: cat {e_cat:.2e} logits {e_lg:.2e} loss {float(loss):.6f} golden {float(g[f'{tag}.loss']):.6f}")
    assert torch.isfinite(loss).item()
    # reference configuration (LayerScale 1e-5, adapter gamma 0): 1e-3 holds in bf16 too — the split-precision convs and patch
    # embedding keep 16 significant bits where it matters; stress weights: bf16's 8-bit mantissa on all 48 block evaluations
    assert e_lg < bound, (tag, e_lg)


def test_bf16_stress_golden_on_precise_level_2(dev, bf16_mode):
    """north_star asks for bf16 MFMA operands AND 1e-3 on the logits.  On the stress golden single bf16 operands cannot hold it
    (the activation sites alone put 8.7e-3 on the logits); ``precise_level 2`` (every linear layer of the ViT
    blocks on hi + lo bf16 operands = 16 significant bits) plus the adapters' MSDA layers on split operands
    (config.precise_adapters_on: automatic for bf16 at level 2 — tests/precision_probe.py vit_large 588 bf16 --fdec puts 1.43e-3 of
    the 1.49e-3 that level 2 leaves into that group: weights 9.2e-4, sampled rows 6.2e-4) is the bf16 configuration that holds
    north_star's 1e-3.  What remains single-bf16: q, k, v and P inside the fused attention (2.6e-4, 2.0e-4), the ConvFFN (3.5e-4),
    the MSDA inputs and value tensor (2.8e-4, 2.7e-4)."""
    g = load_golden("step")
    old = config.precise_level_policy
    config.precise_level_policy = 2
    try:
        eng, _ = build_engine("vit_large", "kernel", dev)
        assert eng.precise_level == 2
        config.precise_level = 2
        assert config.precise_adapters_on()
        img, tgt = W.synthetic_batch(1, 588)
        taps = {}
        loss = eng.train_step(img.to(dev), tgt.to(dev), taps)
    finally:
        config.precise_level_policy = old
    e_x = golden_err(taps["x_final"], g["step_kernel.x_final"])
    e_lg = golden_err(taps["logits"].permute(0, 3, 1, 2), g["step_kernel.logits"])
    print(f"bf16 operands + precise_level 2, step_kernel: x_final {e_x:.2e} logits {e_lg:.2e} loss {float(loss):.6f}")
    assert torch.isfinite(loss).item()
    assert e_lg < BF16_L2_BOUND, e_lg


def _massive_weights(arch):
    """Synthetic weights with DINOv2-like outliers: two residual channels pushed to ~+-600 by one fc2 bias and the patch-embed
    bias, one block with 6x fc1 weights (pre-activations of +-30), one with 3x qkv weights (saturated attention rows)."""
    D, depth, heads, _ = W.VIT_CONFIGS[arch]
    sd = W.make_vit_state_dict(arch, layerscale="kernel")
    sd["patch_embed.proj.bias"][5] += 300.0
    sd["blocks.1.mlp.fc2.bias"][77] -= 2000.0            # x LayerScale gamma (0.05 .. 0.5) -> -100 .. -1000 in the residual
    sd["blocks.2.mlp.fc1.weight"] *= 6.0
    sd["blocks.1.attn.qkv.weight"] *= 3.0
    return sd


def test_f16_operand_range_with_massive_activations(dev):
    arch, size, B = "vit_tiny_test", 224, 2
    D, depth, heads, ffn = W.VIT_CONFIGS[arch]
    eng, sds = build_engine(arch, "kernel", dev, (128, 32, 16, 16, 8), lr=0.05)
    vsd = _massive_weights(arch)
    eng.model.load_state_dict(vsd)
    img, tgt = W.synthetic_batch(B, size, seed=4)
    taps = {}
    loss = eng.train_step(img.to(dev), tgt.to(dev), taps)
    otaps = {}
    with torch.no_grad():
        ocat = O.adapter_forward(img, vsd, {k: v.clone() for k, v in sds["enc"].items()}, sds["cv"], sds["cn"], heads, taps=otaps)
    params = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sds["dec"].items()}
    oloss = O.train_step_loss(ocat, tgt, params, 2, otaps)
    xo = otaps["x_stage3"]
    amax = float(xo.abs().max())
    assert amax > 100.0, amax                              # the outliers really are in the stream
    x = taps["x_final"].float().cpu()
    assert torch.isfinite(x).all() and torch.isfinite(taps["logits"]).all() and torch.isfinite(loss).item()
    keep = [c for c in range(D) if c not in (5, 77)]
    e_all, e_rest = rel_l2(x, xo), rel_l2(x[..., keep], xo[..., keep])
    e_lg = rel_l2(taps["logits"].permute(0, 3, 1, 2), otaps["logits"])
    print(f"massive activations (|x|max {amax:.0f}): x_final {e_all:.2e} (non-outlier channels {e_rest:.2e}) logits {e_lg:.2e}")
    assert e_rest < 1e-3 and e_lg < 1e-3
    assert abs(float(loss) - float(oloss)) < 1e-4


def test_ln_fold_chain_with_massive_activations(dev):
    """VERDICT r4 / ADVICE r4: the LayerNorm-fold chain (residual stream as f16 hi + lo planes, hi as the ONLY A operand of q|k,
    V^T and fc1, LayerNorm statistics from epilogue partial sums, rstd * (acc - mean * cs)) under DINOv2-like outlier channels.
    The massive-activation step test above runs vit_tiny at 224^2, where ``Block.fold_ok`` is false; this one runs a run of four
    ViT-L-width blocks on 17 645 stacked rows (5 + 5 images at 588^2: the smallest batch at which every GEMM of the chain takes a
    kernel that implements the fold fields) and asserts the chain is really taken.  Outliers: one channel at +300 from the
    input, one at -100 .. -1000 from a fc2 bias through LayerScale, 6x fc1 weights in one block (pre-activations of +-30), 3x qkv
    in another (saturated attention rows).  Oracle: the fp32 restatement block by block on the same rows."""
    from adaptersis_amd.dinov2.layers.blocks import run_blocks
    from adaptersis_amd.dinov2.models import vision_transformer as vits
    if config.operand_dtype != torch.float16 or not config.ln_fold:
        pytest.skip("the LayerNorm fold is an f16 path (ASIS_LN_FOLD=1)")
    arch = "vit_large_d4"
    D, depth, heads, ffn = W.VIT_CONFIGS[arch]
    sd = _massive_weights(arch)
    model = vits.__dict__[arch](patch_size=14, img_size=518, init_values=1e-5, ffn_layer=ffn, block_chunks=0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    Bn, N = 5, 1764
    segs = [(Bn, N + 1), (Bn, N)]
    R = sum(b * n for b, n in segs)
    x0 = W.tensor("lnfold.massive.x", (R, D), 1.0)
    x0[:, 5] += 300.0                                     # what the patch-embed bias outlier puts into the stream
    blocks = list(model.blocks)
    assert all(blk.fold_ok(R, segs) for blk in blocks), "the fold chain is not taken at this shape"
    y = run_blocks(blocks, x0.to(dev), segs).float().cpu()
    assert torch.isfinite(y).all()
    ref = []
    r0 = 0
    with torch.no_grad():
        for b, n in segs:
            t = x0[r0:r0 + b * n].view(b, n, D)
            for i in range(depth):
                t = O.block(t, sd, f"blocks.{i}", heads)
            ref.append(t.reshape(b * n, D))
            r0 += b * n
    ref = torch.cat(ref)
    amax = float(ref.abs().max())
    assert amax > 100.0, amax                              # the outliers really are in the stream
    keep = [c for c in range(D) if c not in (5, 77)]
    e_all, e_rest = rel_l2(y, ref), rel_l2(y[:, keep], ref[:, keep])
    e5, e77 = rel_l2(y[:, 5], ref[:, 5]), rel_l2(y[:, 77], ref[:, 77])
    print(f"LayerNorm-fold chain, massive activations (|x|max {amax:.0f}, {R} rows): all {e_all:.2e} non-outlier channels {e_rest:.2e} "
          f"channel 5 {e5:.2e} channel 77 {e77:.2e}")
    # the same bars as the step-level test: 1e-3 on the channels that carry the signal; the outlier channels themselves are
    # large values carried exactly by the hi + lo planes (22 significant bits)
    assert e_rest < 1e-3 and e5 < 1e-4 and e77 < 1e-3
    # and against the unfolded path on the same rows: the fold changes WHERE the LayerNorm is applied, not what is computed
    old = config.ln_fold
    try:
        config.ln_fold = False
        y0 = run_blocks(blocks, x0.to(dev), segs).float().cpu()
    finally:
        config.ln_fold = old
    print(f"  unfolded path on the same rows: non-outlier channels {rel_l2(y0[:, keep], ref[:, keep]):.2e}; fold vs unfolded {rel_l2(y[:, keep], y0[:, keep]):.2e}")
    assert rel_l2(y[:, keep], y0[:, keep]) < 1e-3
