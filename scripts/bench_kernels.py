"""Micro-benchmarks of the hot kernels at ViT-L/14 588^2 shapes (B images). HIP-event timed."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--dtype", default="f16")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    dev = torch.device("cuda:0")
    B, N, D, H = a.batch, 1764, 1024, 16
    M = B * N
    g = torch.Generator(device="cpu").manual_seed(0)

    def rand(*s):
        return (torch.rand(*s, generator=g) * 2 - 1).to(dev)

    print(f"B={B} M={M} dtype={dt}")
    for name, n, k, kw in (("qk", 2 * D, D, {}), ("proj+ls+res", D, D, {"res": True}), ("fc1+gelu", 4 * D, D, {"act": ops.ACT_GELU}),
                           ("fc2+ls+res", D, 4 * D, {"res": True})):
        x = rand(M, k).to(dt)
        w = (rand(n, k) * 0.05).to(dt)
        bias = rand(n)
        extra = {}
        if kw.get("res"):
            extra = dict(out_f32=True, scale_n=rand(n), res=rand(M, n))
        if kw.get("act"):
            extra = dict(act=kw["act"])
        out = torch.empty(M, n, device=dev, dtype=torch.float32 if kw.get("res") else dt)
        ms = timeit(lambda: ops.gemm(x, w, out=out, bias_n=bias, **extra))
        print(f"gemm {name:14s} M={M} N={n} K={k}: {ms:.3f} ms  {2.0 * M * n * k / ms / 1e9:.1f} TFLOP/s")
    # V^T gemm
    x = rand(B, N, D).to(dt)
    wv = (rand(D, D) * 0.05).to(dt)
    ldvt = 1792
    vt = torch.zeros(B, D, ldvt, device=dev, dtype=dt)
    bias = rand(D)
    ms = timeit(lambda: ops.gemm(wv, x, out=vt.as_strided((B, D, N), (D * ldvt, ldvt, 1)), bias_m=bias))
    print(f"gemm v^T batched: {ms:.3f} ms  {2.0 * M * D * D / ms / 1e9:.1f} TFLOP/s")
    qk = rand(M, 2 * D).to(dt)
    o = torch.empty(M, D, device=dev, dtype=dt)
    ms = timeit(lambda: ops.attention_fwd(qk[:, :D], qk[:, D:], vt, B, H, N, 0.125, out=o))
    print(f"attention fwd: {ms:.3f} ms  {4.0 * B * H * N * N * 64 / ms / 1e9:.1f} TFLOP/s")
    xf = rand(M, D)
    w1, b1 = rand(D), rand(D)
    y = torch.empty(M, D, device=dev, dtype=dt)
    ms = timeit(lambda: ops.layernorm(xf, w1, b1, 1e-6, dt, out=y))
    print(f"layernorm: {ms:.3f} ms  {M * D * 6 / ms / 1e6:.1f} GB/s")


if __name__ == "__main__":
    main()
