"""Attention forward microbench (ViT-L shape): python scripts/bench_attn.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops


def main():
    dev = torch.device("cuda:0")
    B, H, N, D = 12, 16, 1764, 1024
    dt = torch.float16
    M = B * N
    qk = (torch.randn(M, 2 * D, device=dev)).to(dt)
    ldvt = 1792
    vt = torch.zeros(B, D, ldvt, device=dev, dtype=dt)
    vt[:, :, :N] = torch.randn(B, D, N, device=dev).to(dt)
    o = torch.empty(M, D, device=dev, dtype=dt)
    f = lambda: ops.attention_fwd(qk[:, :D], qk[:, D:], vt, B, H, N, 0.125, out=o)
    if os.environ.get("ASIS_ATTN_VAR") == "qkv":   # row-major V out of one [tokens, 3 D] matrix
        qkv = torch.cat([qk, vt[:, :, :N].transpose(1, 2).reshape(M, D)], dim=1).contiguous()
        f = lambda: ops.attention_fwd_qkv(qkv, [(B, N)], H, 0.125, o)
    f()
    q = qk[:, :D].float().view(B, N, H, 64).transpose(1, 2)
    k = qk[:, D:].float().view(B, N, H, 64).transpose(1, 2)
    v = vt[:, :, :N].float().view(B, H, 64, N).transpose(2, 3)
    ref = torch.softmax(q @ k.transpose(2, 3) * 0.125, -1) @ v
    ref = ref.transpose(1, 2).reshape(M, D)
    err = ((o.float() - ref).norm() / ref.norm()).item()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"ABLATE={os.environ.get('ASIS_ATTN_ABLATE', '0')} VAR={os.environ.get('ASIS_ATTN_VAR', '-')}: {ms:.3f} ms  "
          f"{4.0 * B * H * N * N * 64 / ms / 1e9:.0f} TFLOP/s  rel {err:.2e}")


if __name__ == "__main__":
    main()
