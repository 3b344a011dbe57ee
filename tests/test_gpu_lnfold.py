"""GPU: the LayerNorm-fold chain (config.ln_fold; include/asis_hip.h: asis_gemm_desc.C_lo / rowstats / res16 / ln_mr) — the
residual stream of the frozen trunk as two 16-bit planes between GEMM epilogues, LayerNorm (`block.py:89-114`: norm1 / norm2,
`vision_transformer.py:89` eps 1e-6) folded into the weights of the layers that consume it and undone in their epilogues.
Each epilogue field against fp32 torch on the same 16-bit-rounded operands, on both kernels that implement them (the persistent
8-phase kernel and the one-tile-per-workgroup 8-phase form), then two chained ViT-L-width blocks against the fp32 oracle block
next to the unfolded path.  The headline-batch golden (tests/test_gpu_step.py::test_vitl_588_headline_batch_two_steps) runs
the chain end to end against the imported reference."""
import pytest
import torch
import torch.nn.functional as F

from adaptersis_amd import config, ops
from adaptersis_amd.dinov2.layers.blocks import Block, MemEffAttention, Mlp, run_blocks
from adaptersis_amd.utils import weights as W
from oracle import ref_torch as O
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu
DT = torch.float16


def test_split_stats_planes_and_statistics(dev):
    x = (W.tensor("lf.x", (3000, 1024), 1.0) * 3 + W.tensor("lf.m", (3000, 1), 1.0) * 2).to(dev)
    hi, lo, mr = ops.split_stats(x, DT, 1e-6)
    assert torch.equal(hi, x.to(DT)) and torch.equal(lo, (x - x.to(DT).float()).to(DT))
    assert rel_l2(hi.float() + lo.float(), x) < 3e-7                      # ~22 significant bits
    mean, var = x.double().mean(1), x.double().var(1, unbiased=False)
    assert rel_l2(mr[:, 0], mean.float()) < 1e-6 and rel_l2(mr[:, 1], (var + 1e-6).rsqrt().float()) < 1e-6


@pytest.mark.parametrize("p8,M,N,K", [(1, 17645, 1024, 1024), (0, 17645, 1024, 4096), (1, 9000, 512, 2048),
                                      # N % 256 in {64, 128, 192}: the last 256-column tile overhangs N, its wave column groups past
                                      # the last 64-column group must not store statistics (they zeroed the next row's)
                                      (0, 9001, 1088, 1024), (0, 9001, 1152, 1024), (0, 9001, 1216, 4096), (1, 17645, 1344, 1024)])
def test_planes_rowstats_res16_epilogue(dev, p8, M, N, K):
    """proj / fc2 shaped: out = res + scale * (A B^T + bias) with the residual read from two planes and the result written as
    two planes + per-row partial sums; finalize -> LayerNorm statistics of the fp32 result."""
    ops.gemm_set_option("p8", p8)
    try:
        a = W.tensor(f"lf.a{M}.{K}", (M, K), 1.0).to(dev).to(DT)
        b = W.tensor(f"lf.b{N}.{K}", (N, K), 0.05).to(dev).to(DT)
        bn, sc = W.tensor(f"lf.bn{N}", (N,), 1.0).to(dev), W.tensor(f"lf.sc{N}", (N,), 1.0).to(dev)
        res = (W.tensor(f"lf.r{M}.{N}", (M, N), 3.0) + 0.7).to(dev)
        rh, rl, _ = ops.split_stats(res, DT)
        ref = (rh.float() + rl.float()) + sc * (a.float() @ b.float().t() + bn)
        outs = []
        for _ in range(2):
            oh, ol = torch.full((M, N), 7.0, device=dev, dtype=DT), torch.full((M, N), 7.0, device=dev, dtype=DT)
            # one guard row behind the table: a store past the last row's groups would land there
            stg = torch.full((M + 1, (N + 63) // 64, 2), -123.0, device=dev, dtype=torch.float32)
            st = stg[:M]
            ops.gemm(a, b, out=oh, out_lo=ol, rowstats=st, bias_n=bn, scale_n=sc, res16=(rh, rl))
            assert bool((stg[M] == -123.0).all()), "row statistics written past the table"
            outs.append((oh, ol, st))
        oh, ol, st = outs[0]
        assert all(torch.equal(x, y) for x, y in zip(outs[0], outs[1])), "not reproducible"
        assert rel_l2(oh.float() + ol.float(), ref) < 2e-6
        mr = ops.ln_stats_finalize(st, N, 1e-6)
        mean, var = ref.double().mean(1), ref.double().var(1, unbiased=False)
        assert rel_l2(mr[:, 0], mean.float()) < 2e-5 and rel_l2(mr[:, 1], (var + 1e-6).rsqrt().float()) < 2e-5
        # the same launch with an fp32 output (the last block of a run): bit-consistent with the planes
        o32 = ops.gemm(a, b, out_f32=True, bias_n=bn, scale_n=sc, res16=(rh, rl))
        assert torch.equal(o32.to(DT), oh) and torch.equal((o32 - oh.float()).to(DT), ol) and rel_l2(o32, ref) < 2e-6
    finally:
        ops.gemm_set_option("p8", 1)


@pytest.mark.parametrize("rows,D", [(1000, 768), (4097, 1024), (333, 1536), (70, 1000), (5, 64), (129, 2048)])
def test_ln_stats_finalize_from_group_sums(dev, rows, D):
    """(sum, sum of squares) per 64-column group -> (mean, rstd): 16 / 32 lanes per row, ragged last group, rows % 16 != 0"""
    x = (W.tensor(f"lsf.x{rows}.{D}", (rows, D), 1.0) * 3 + W.tensor(f"lsf.m{rows}", (rows, 1), 5.0)).to(dev)
    groups = (D + 63) // 64
    xp = torch.nn.functional.pad(x, (0, groups * 64 - D)).view(rows, groups, 64)
    st = torch.stack((xp.sum(2), (xp * xp).sum(2)), 2).contiguous()
    mr = ops.ln_stats_finalize(st, D, 1e-6)
    xd = x.double()
    assert mr.shape == (rows, 2)
    assert rel_l2(mr[:, 0], xd.mean(1).float()) < 1e-6
    assert rel_l2(mr[:, 1], (xd.var(1, unbiased=False) + 1e-6).rsqrt().float()) < 2e-5
    assert torch.equal(mr, ops.ln_stats_finalize(st, D, 1e-6))


@pytest.mark.parametrize("p8,M,N,K,act", [(2, 17645, 2048, 1024, ops.ACT_NONE), (2, 17645, 4096, 1024, ops.ACT_GELU),
                                          (0, 17645, 2048, 1024, ops.ACT_NONE)])
def test_ln_fold_rows(dev, p8, M, N, K, act):
    """qk / fc1 shaped: LN(x) W^T + b from the hi plane of x, W' = W diag(w_ln), (mean, rstd) per row."""
    ops.gemm_set_option("p8", p8)
    try:
        x = (W.tensor(f"lr.x{M}", (M, K), 1.0) * 2 + W.tensor(f"lr.m{M}", (M, 1), 1.0)).to(dev)
        wln, bln = (1 + 0.3 * W.tensor("lr.w", (K,), 1.0)).to(dev), (0.2 * W.tensor("lr.b", (K,), 1.0)).to(dev)
        w, b = W.tensor(f"lr.W{N}", (N, K), 0.05).to(dev), W.tensor(f"lr.B{N}", (N,), 1.0).to(dev)
        hi, _, mr = ops.split_stats(x, DT)
        w16 = ops.cast_pad((w * wln[None]).contiguous(), dtype=DT)
        cs, b2 = w16.float().sum(1).contiguous(), (b + w @ bln).contiguous()
        out = ops.gemm(hi, w16, bias_n=b2, act=act, ln=(mr, cs, False))
        ref = F.layer_norm(x, (K,), wln, bln, 1e-6) @ w.t() + b
        if act == ops.ACT_GELU:
            ref = F.gelu(ref)
        # against the exact fp32 LayerNorm + Linear: one 16-bit rounding of x, of W' and of the output
        assert rel_l2(out, ref) < 1.2e-3
        # against the same arithmetic in fp32 on the rounded operands: only the output rounding remains
        ref2 = mr[:, 1:2] * (hi.float() @ w16.float().t() - mr[:, 0:1] * cs[None]) + b2
        if act == ops.ACT_GELU:
            ref2 = F.gelu(ref2)
        assert rel_l2(out, ref2) < 6e-4
    finally:
        ops.gemm_set_option("p8", 1)


def test_ln_fold_columns_batched_vt(dev):
    """the swapped V^T GEMM: out[b, d, t] = LN(x)[b, t, :] . W_v[d, :] + b_v[d], statistics per output column."""
    Bn, N, D = 6, 1765, 1024
    N8 = (N + 7) // 8 * 8
    ldv = (N + 63) // 64 * 64
    x = (W.tensor("lc.x", (Bn * N, D), 1.0) * 2 + W.tensor("lc.m", (Bn * N, 1), 1.0)).to(dev)
    wln, bln = (1 + 0.3 * W.tensor("lc.w", (D,), 1.0)).to(dev), (0.2 * W.tensor("lc.b", (D,), 1.0)).to(dev)
    wv, bv = W.tensor("lc.W", (D, D), 0.05).to(dev), W.tensor("lc.B", (D,), 1.0).to(dev)
    hi, _, mr = ops.split_stats(x, DT)
    w16 = ops.cast_pad((wv * wln[None]).contiguous(), dtype=DT)
    cs, b2 = w16.float().sum(1).contiguous(), (bv + wv @ bln).contiguous()
    vt = torch.full((Bn, D, ldv), 5.0, device=dev, dtype=DT)
    ops.gemm(w16, hi.as_strided((Bn, N8, D), (N * D, D, 1)), out=vt.as_strided((Bn, D, N8), (D * ldv, ldv, 1)), bias_m=b2,
             ln=(mr, cs, True))
    ref = torch.einsum("fd,bnd->bfn", wv, F.layer_norm(x, (D,), wln, bln, 1e-6).view(Bn, N, D)) + bv[None, :, None]
    assert rel_l2(vt[:, :, :N], ref) < 1.2e-3
    assert torch.all(vt[:, :, N8:] == 5.0)


def _vitl_block(dev, i):
    sd = W.make_vit_state_dict("vit_large_d4", layerscale="kernel")
    blk = Block(1024, 16, mlp_ratio=4.0, qkv_bias=True, proj_bias=True, ffn_bias=True, init_values=1e-5, attn_class=MemEffAttention, ffn_layer=Mlp)
    blk.load_state_dict({k[len(f"blocks.{i}."):]: v for k, v in sd.items() if k.startswith(f"blocks.{i}.")})
    return blk.to(dev).eval(), sd


def test_two_blocks_on_the_fold_chain_vs_oracle(dev):
    """two ViT-L-width blocks, two stacked token batches (5 images each: every GEMM on the 8-phase kernels): the fold chain
    (norm1 of block 0 as a kernel, planes from there on, fp32 out of block 1) against the fp32 oracle block, next to the
    unfolded path."""
    b0, sd = _vitl_block(dev, 0)
    b1, _ = _vitl_block(dev, 1)
    Bn, N = 5, 1764
    segs = [(Bn, N + 1), (Bn, N)]
    R = Bn * (2 * N + 1)
    x = W.tensor("lf.tok", (R, 1024), 1.0).to(dev)
    old = config.ln_fold
    saved = (config.split_attn_out, config.precise_level, config.operand_dtype)   # engines of earlier tests leave their policy here
    config.split_attn_out, config.precise_level = False, 0
    config.set_operand_dtype(DT)
    try:
        config.ln_fold = True
        assert b0.fold_ok(R, segs) and b1.fold_ok(R, segs)
        y_fold = run_blocks([b0, b1], x, segs)
        y_fold2 = run_blocks([b0, b1], x, segs)
        config.ln_fold = False
        assert not b0.fold_ok(R, segs)
        y_std = run_blocks([b0, b1], x, segs)
    finally:
        config.ln_fold = old
        config.split_attn_out, config.precise_level = saved[:2]
        config.set_operand_dtype(saved[2])
    torch.cuda.synchronize()
    assert torch.equal(y_fold, y_fold2), "fold chain is not reproducible"
    xc = x.cpu()
    refs, r0 = [], 0
    for Bs, Ns in segs:      # oracle per image of the first and the last image of each batch (CPU time)
        for img in (0, Bs - 1):
            t = xc[r0 + img * Ns: r0 + (img + 1) * Ns][None]
            with torch.no_grad():
                t = O.block(O.block(t, sd, "blocks.0", 16), sd, "blocks.1", 16)
            refs.append((r0 + img * Ns, t[0]))
        r0 += Bs * Ns
    e_fold = max(rel_l2(y_fold[a:a + t.shape[0]] - x[a:a + t.shape[0]], t.to(dev) - x[a:a + t.shape[0]]) for a, t in refs)
    e_std = max(rel_l2(y_std[a:a + t.shape[0]] - x[a:a + t.shape[0]], t.to(dev) - x[a:a + t.shape[0]]) for a, t in refs)
    print(f"two blocks, branch outputs vs fp32 oracle: fold chain {e_fold:.2e}, unfolded {e_std:.2e}")
    assert e_fold < 1.5e-3 and e_fold < 1.3 * e_std + 1e-4
