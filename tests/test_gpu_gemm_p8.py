"""GPU: the PERSISTENT 8-phase dense GEMM (`csrc/gemm_p8.h`, 31 % of the headline step) forced onto multi-tile-per-workgroup
shapes (`asis_gemm_set_option("p8", 2)`: from 16 tiles on, any K) — every epilogue the step dispatches to it (q|k 16-bit + bias,
fc1 bias + erf-GELU, proj / fc2 LayerScale + fp32 residual, GELU-grad input gradients, plain fp32), split-precision K parts
(A_lo alone = `config.split_attn_out`; A_lo + B_lo = `precise_level 2`), ragged last row / column tiles, a padded output whose pad
must stay untouched, three runs each (a racy tile hand-off shows as run-to-run differences) and bit-identity with the
one-tile-per-workgroup kernel (`csrc/gemm_big.h`, same arithmetic order).  fp32 torch on the same 16-bit-rounded operands is the
reference (tests/test_gpu_kernels.py conventions).  This was `scripts/gemm_p8_probe.py check`."""
import pytest
import torch
import torch.nn.functional as F

from adaptersis_amd import ops
from adaptersis_amd.utils import weights as W
from tests.conftest import rel_l2

pytestmark = pytest.mark.gpu

# (M, N, K): ragged M and N, K = 128 .. 4096, the stacked ViT-L row count of the headline batch (42 348 = 12 x (1765 + 1764))
SHAPES = [(8192 + 77, 2048, 1024), (4100, 1096, 512), (42348, 1024, 1024), (4096, 1024, 128), (9000, 512, 4096), (5000, 4096, 1024)]


@pytest.fixture()
def p8_forced():
    ops.gemm_set_option("p8", 2)
    yield
    ops.gemm_set_option("p8", 1)


def _operands(dev, dt, M, N, K):
    a = W.tensor(f"p8.a{M}", (M, K), 1.0).to(dev).to(dt)
    b = W.tensor(f"p8.b{N}.{K}", (N, K), 1.0).to(dev).to(dt)
    bn, sc = W.tensor(f"p8.bn{N}", (N,), 1.0).to(dev), W.tensor(f"p8.sc{N}", (N,), 1.0).to(dev)
    res = W.tensor(f"p8.r{M}.{N}", (M, N), 3.0).to(dev)
    aux = W.tensor(f"p8.aux{M}.{N}", (M, N), 1.5).to(dev).to(dt)
    return a, b, bn, sc, res, aux


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_p8_every_epilogue(dev, p8_forced, dt, M, N, K):
    if dt == torch.bfloat16 and (M, N, K) not in (SHAPES[1], SHAPES[2]):
        pytest.skip("bf16: two shapes cover the template instance")
    a, b, bn, sc, res, aux = _operands(dev, dt, M, N, K)
    acc = a.float() @ b.float().t()
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    cases = {
        "bias 16-bit": (dict(bias_n=bn), acc + bn),
        "bias gelu 16-bit": (dict(bias_n=bn, act=ops.ACT_GELU), F.gelu(acc + bn)),
        "plain f32": (dict(out_f32=True), acc),
        "bias scale res f32": (dict(out_f32=True, bias_n=bn, scale_n=sc, res=res), res + sc * (acc + bn)),
        "scale 16-bit": (dict(scale_n=sc), sc * acc),
        "relu f32": (dict(out_f32=True, bias_n=bn, act=ops.ACT_RELU), F.relu(acc + bn)),
        "gelu-grad 16-bit": (dict(act=ops.ACT_GELU_GRAD, aux=aux), acc * x.grad),
    }
    tol16 = 1e-3 if dt == torch.float16 else 6e-3     # one 16-bit rounding of the output
    for name, (kw, ref) in cases.items():
        outs = []
        for _ in range(3):
            o = torch.full((M, N + 8), 7.0, device=dev, dtype=torch.float32 if kw.get("out_f32") else dt)
            ops.gemm(a, b, out=o[:, :N], **kw)
            outs.append(o)
        torch.cuda.synchronize()
        c = outs[0]
        assert torch.all(c[:, N:] == 7.0), f"{name}: wrote outside its columns"
        assert all(torch.equal(c, o) for o in outs[1:]), f"{name}: not reproducible run to run"
        assert bool(torch.isfinite(c.float()).all()), name
        e = rel_l2(c[:, :N], ref)
        assert e < (3e-6 if c.dtype == torch.float32 else tol16), (name, e)


@pytest.mark.parametrize("M,N,K", [s for s in SHAPES if s[2] <= 1024])
def test_p8_split_precision_parts(dev, p8_forced, M, N, K):
    """hi + lo K parts on the persistent stream: (A, B), (A_lo, B) [, (A, B_lo)] against fp32 torch on the same halves."""
    dt = torch.float16
    _, _, bn, sc, res, _ = _operands(dev, dt, M, N, K)
    a32 = W.tensor(f"p8.a32.{M}", (M, K), 1.0).to(dev)
    b32 = W.tensor(f"p8.b32.{N}.{K}", (N, K), 1.0).to(dev)
    ah, bh = a32.to(dt), b32.to(dt)
    al, bl = (a32 - ah.float()).to(dt), (b32 - bh.float()).to(dt)
    if M * N >= 256 * 65536:
        ops.gemm_set_option("p8", 3)
    cases = (("A_lo f32 + res", dict(a_lo=al, out_f32=True, bias_n=bn, scale_n=sc, res=res),
              res + sc * ((ah.float() + al.float()) @ bh.float().t() + bn)),
             ("A_lo + B_lo f32", dict(a_lo=al, b_lo=bl, out_f32=True),
              (ah.float() + al.float()) @ bh.float().t() + ah.float() @ bl.float().t()))
    for name, kw, ref in cases:
        outs = [ops.gemm(ah, bh, **kw) for _ in range(3)]
        torch.cuda.synchronize()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), f"{name}: not reproducible run to run"
        e = rel_l2(outs[0], ref)
        assert e < 3e-6, (name, e)


@pytest.mark.parametrize("M,N,K", [SHAPES[0], SHAPES[2], SHAPES[4]])
def test_p8_bit_identical_to_one_tile_per_workgroup(dev, M, N, K):
    dt = torch.float16
    a, b, bn, sc, res, _ = _operands(dev, dt, M, N, K)
    try:
        ops.gemm_set_option("p8", 0)
        o0 = ops.gemm(a, b, out_f32=True, bias_n=bn, scale_n=sc, res=res)
        h0 = ops.gemm(a, b, bias_n=bn, act=ops.ACT_GELU)
        ops.gemm_set_option("p8", 2)
        o1 = ops.gemm(a, b, out_f32=True, bias_n=bn, scale_n=sc, res=res)
        h1 = ops.gemm(a, b, bias_n=bn, act=ops.ACT_GELU)
    finally:
        ops.gemm_set_option("p8", 1)
    assert torch.equal(o0, o1) and torch.equal(h0, h1)
