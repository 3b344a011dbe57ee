"""Persistent 8-phase dense GEMM (csrc/gemm_p8.h) against the one-tile-per-workgroup form (csrc/gemm_big.h) in ONE process
(`asis_gemm_set_option("p8", v)`): correctness on multi-tile-per-workgroup shapes with ragged M / N and every epilogue, run
three times each (a racy hand-off shows as run-to-run differences), then interleaved timing rounds on the stacked ViT-L shapes.
    python scripts/gemm_p8_probe.py check        -> `case ...` lines + `worst <err>` + `unstable <n>`
    python scripts/gemm_p8_probe.py time [R]     -> per shape: median / min us of R interleaved rounds, p8 = 0 and 1"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from adaptersis_amd import ops
from adaptersis_amd.utils import weights as W

dev = torch.device("cuda:0")
dt = torch.float16


def check():
    worst, unstable = 0.0, 0
    ops.gemm_set_option("p8", 2)
    for (M, N, K) in ((8192 + 77, 2048, 1024), (4100, 1096, 512), (42348, 1024, 1024), (4096, 1024, 128), (9000, 512, 4096),
                      (5000, 4096, 1024)):
        a = W.tensor(f"p8.a{M}", (M, K), 1.0).to(dev).to(dt)
        b = W.tensor(f"p8.b{N}.{K}", (N, K), 1.0).to(dev).to(dt)
        bn, sc = W.tensor(f"p8.bn{N}", (N,), 1.0).to(dev), W.tensor(f"p8.sc{N}", (N,), 1.0).to(dev)
        res = W.tensor(f"p8.r{M}.{N}", (M, N), 3.0).to(dev)
        aux = W.tensor(f"p8.aux{M}.{N}", (M, N), 1.5).to(dev).to(dt)
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
        for name, (kw, ref) in cases.items():
            outs = []
            for rep in range(3):
                o = torch.full((M, N + 8), 7.0, device=dev, dtype=torch.float32 if kw.get("out_f32") else dt)
                ops.gemm(a, b, out=o[:, :N], **kw)
                outs.append(o)
            torch.cuda.synchronize()
            c = outs[0]
            if not torch.all(c[:, N:] == 7.0):
                print("WROTE OUTSIDE ITS COLUMNS"); unstable += 100
            same = all(torch.equal(c, o) for o in outs[1:])
            unstable += 0 if same else 1
            tol_scale = 1.0 if c.dtype == torch.float32 else 300.0
            e = float((c[:, :N].float() - ref).norm() / ref.norm()) / tol_scale
            bad = int((~torch.isfinite(c.float())).sum())
            worst = max(worst, e if bad == 0 else 1.0)
            print(f"case {M}x{N}x{K} {name}: err {e * tol_scale:.2e} non-finite {bad} stable {same}")
        # split-precision operands as further K parts of the persistent stream: A_lo alone (the attention output of
        # config.split_attn_out), A_lo + B_lo
        if K <= 1024:
            a32 = W.tensor(f"p8.a32.{M}", (M, K), 1.0).to(dev)
            b32 = W.tensor(f"p8.b32.{N}.{K}", (N, K), 1.0).to(dev)
            ah, bh = a32.to(dt), b32.to(dt)
            al, bl = (a32 - ah.float()).to(dt), (b32 - bh.float()).to(dt)
            ops.gemm_set_option("p8", 3 if M * N >= 256 * 65536 else 2)
            for name, kw, ref in (("A_lo f32 + res", dict(a_lo=al, out_f32=True, bias_n=bn, scale_n=sc, res=res),
                                   res + sc * ((ah.float() + al.float()) @ bh.float().t() + bn)),
                                  ("A_lo + B_lo f32", dict(a_lo=al, b_lo=bl, out_f32=True),
                                   (ah.float() + al.float()) @ bh.float().t() + ah.float() @ bl.float().t())):
                outs = [ops.gemm(ah, bh, **kw) for _ in range(3)]
                torch.cuda.synchronize()
                same = all(torch.equal(outs[0], o) for o in outs[1:])
                e = float((outs[0] - ref).norm() / ref.norm())
                unstable += 0 if same else 1
                worst = max(worst, e)
                print(f"case {M}x{N}x{K} {name}: err {e:.2e} stable {same}")
            ops.gemm_set_option("p8", 2)
        # bit-identical to the one-tile-per-workgroup kernel (same arithmetic order)
        ops.gemm_set_option("p8", 0)
        o0 = ops.gemm(a, b, out_f32=True, bias_n=bn, scale_n=sc, res=res)
        h0 = ops.gemm(a, b, bias_n=bn, act=ops.ACT_GELU)
        ops.gemm_set_option("p8", 2)
        o1 = ops.gemm(a, b, out_f32=True, bias_n=bn, scale_n=sc, res=res)
        h1 = ops.gemm(a, b, bias_n=bn, act=ops.ACT_GELU)
        eq = torch.equal(o0, o1) and torch.equal(h0, h1)
        print(f"case {M}x{N}x{K} bit-identical to gemm_big: {eq}")
        unstable += 0 if eq else 1
    print(f"worst {worst:.3e}")
    print(f"unstable {unstable}")


SHAPES = [("qk", 42348, 2048, 1024, {}), ("proj+ls+res", 42348, 1024, 1024, {"res": 1}), ("fc1+gelu", 42348, 4096, 1024, {"act": 1}),
          ("fc2+ls+res", 42348, 1024, 4096, {"res": 1}), ("ViT-g w12", 42348, 8192, 1536, {}), ("ViT-B qk", 42348, 1536, 768, {})]


def timing(rounds):
    fs = {}
    for name, M, N, K, kw in SHAPES:
        x = (torch.rand(M, K, device=dev) * 2 - 1).half()
        w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).half()
        b = torch.rand(N, device=dev)
        extra = {}
        if kw.get("res"):
            extra = dict(out_f32=True, scale_n=torch.rand(N, device=dev), res=torch.rand(M, N, device=dev))
        if kw.get("act"):
            extra = dict(act=ops.ACT_GELU)
        o = torch.empty(M, N, device=dev, dtype=torch.float32 if kw.get("res") else torch.float16)
        fs[name] = (lambda x=x, w=w, o=o, b=b, extra=extra: ops.gemm(x, w, out=o, bias_n=b, **extra), 2.0 * M * N * K)
    res = {(n, v): [] for n in fs for v in (0, 1)}
    for v in (0, 1):   # warm both forms
        ops.gemm_set_option("p8", v)
        for n, (f, _) in fs.items():
            for _ in range(10):
                f()
    for r in range(rounds):
        for n, (f, _) in fs.items():
            for v in (0, 1):
                ops.gemm_set_option("p8", v)
                f()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    f()
                e.record()
                torch.cuda.synchronize()
                res[(n, v)].append(s.elapsed_time(e) / 10 * 1e3)
    for n, (f, fl) in fs.items():
        m0, m1 = statistics.median(res[(n, 0)]), statistics.median(res[(n, 1)])
        print(f"{n:12s} gemm_big {m0:7.1f} us (min {min(res[(n, 0)]):7.1f}, {fl / m0 / 1e6:5.0f} TF)   p8 {m1:7.1f} us "
              f"(min {min(res[(n, 1)]):7.1f}, {fl / m1 / 1e6:5.0f} TF)   p8/big {m1 / m0:.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "check":
        check()
    else:
        timing(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
