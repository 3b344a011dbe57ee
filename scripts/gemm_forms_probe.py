"""Every dense-GEMM form of the dispatcher on the same cases (one process per form: the switches are read once).
    ASIS_GEMM_8P=2 python scripts/gemm_forms_probe.py      8-phase 256x256x64 form wherever it fits (16x16x32 MFMAs)
    ASIS_GEMM_8P=2 ASIS_GEMM_8P_M16=0 ...                   the same on 32x32x16 MFMAs
    ASIS_GEMM_8P=0 ...                                      two-workgroup 256x128x32 form only
Prints `case ... err <rel-L2>` lines and `worst <max err>`; ragged M / N, strided output, every epilogue variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from adaptersis_amd import ops
from adaptersis_amd.utils import weights as W

dev = torch.device("cuda:0")
dt = torch.float16
worst = 0.0
for (M, N, K) in ((1000, 512, 1024), (2309, 1100, 256), (256, 256, 64), (4100, 768, 2048), (777, 1024, 1088)):
    a = W.tensor(f"gf.a{M}", (M, K), 1.0).to(dev).to(dt)
    b = W.tensor(f"gf.b{N}", (N, K), 1.0).to(dev).to(dt)
    bn, sc = W.tensor(f"gf.bn{N}", (N,), 1.0).to(dev), W.tensor(f"gf.sc{N}", (N,), 1.0).to(dev)
    bm = W.tensor(f"gf.bm{M}", (M,), 1.0).to(dev)
    res = W.tensor(f"gf.r{M}", (M, N), 3.0).to(dev)
    acc = a.float() @ b.float().t()
    cases = {
        "plain f32": (dict(out_f32=True), acc),
        "bias gelu 16-bit": (dict(bias_n=bn, act=ops.ACT_GELU), F.gelu(acc + bn)),
        "bias scale res f32": (dict(out_f32=True, bias_n=bn, scale_n=sc, res=res), res + sc * (acc + bn)),
        "bias_m relu f32": (dict(out_f32=True, bias_m=bm, act=ops.ACT_RELU), F.relu(acc + bm[:, None])),
    }
    for name, (kw, ref) in cases.items():
        c = ops.gemm(a, b, **kw)
        tol_scale = 1.0 if c.dtype == torch.float32 else 300.0
        e = float((c.float() - ref).norm() / ref.norm()) / tol_scale
        worst = max(worst, e)
        print(f"case {M}x{N}x{K} {name}: err {e * tol_scale:.2e}")
    big = torch.full((M, N + 8), 7.0, device=dev)
    ops.gemm(a, b, out=big[:, :N], bias_n=bn)
    e = float((big[:, :N] - (acc + bn)).norm() / acc.norm())
    worst = max(worst, e)
    assert torch.all(big[:, N:] == 7.0)
    print(f"case {M}x{N}x{K} strided out: err {e:.2e}")
    # GELU'(aux) epilogue (fc2's input gradient)
    if M >= 256 and N >= 128 and K % 32 == 0:
        aux = W.tensor(f"gf.aux{M}", (M, N), 1.5).to(dev).to(dt)
        g = ops.gemm(a, b, act=ops.ACT_GELU_GRAD, aux=aux)
        x = aux.float().requires_grad_(True)
        F.gelu(x).sum().backward()
        ref = acc * x.grad
        e = float((g.float() - ref).norm() / ref.norm()) / 300.0
        worst = max(worst, e)
        print(f"case {M}x{N}x{K} gelu-grad 16-bit: err {e * 300:.2e}")
print(f"worst {worst:.3e}")
