"""Attention backward microbench (ViT-L shape, config 4's two stacked passes): python scripts/bench_attn_bwd.py
asis_attention_bwd_rows, both batches in one launch and one launch per batch (the round-1 form with transposed operand images
it replaced measured 1.63 ms for both batches on the same box: profiles/r05_attn_bwd_pmc.txt)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    H, D = 16, 1024
    segs = [(12, 1765), (12, 1764)]
    dt = torch.float16
    R = sum(b * n for b, n in segs)
    torch.manual_seed(0)
    qkv = torch.randn(R, 3 * D, device=dev).to(dt)
    dO = torch.randn(R, D, device=dev).to(dt)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o = torch.empty(R, D, device=dev, dtype=dt)
    lse = torch.empty(R * H, device=dev, dtype=torch.float32)
    r0 = l0 = 0
    views = []
    for B, N in segs:
        r1, l1 = r0 + B * N, l0 + B * H * N
        vt = ops.transpose_tokens(v[r0:r1], B, N)
        ops.attention_fwd(q[r0:r1], k[r0:r1], vt, B, H, N, 0.125, out=o[r0:r1], lse=lse[l0:l1].view(B, H, N))
        views.append((r0, r1, lse[l0:l1].view(B, H, N)))
        r0, l0 = r1, l1
    out_new = torch.empty(R, 3 * D, device=dev, dtype=dt)

    def new():
        ops.attention_bwd_rows(q, k, v, o, dO, lse, segs, H, 0.125, dqkv=out_new)

    def new_sep():
        for (B, N), (a, b, l) in zip(segs, views):
            ops.attention_bwd_rows(q[a:b], k[a:b], v[a:b], o[a:b], dO[a:b], l.reshape(-1), [(B, N)], H, 0.125, dqkv=out_new[a:b])

    if os.environ.get("ASIS_PMC"):   # counter passes: three launches
        for _ in range(3):
            new()
        torch.cuda.synchronize()
        return
    new()
    torch.cuda.synchronize()
    print("finite", bool(torch.isfinite(out_new.float()).all()))
    fl = sum(14.0 * B * H * N * N * 64 for B, N in segs)   # 7 products
    for name, f in (("both batches in one launch", new), ("one launch per batch", new_sep)):
        ms = timeit(f)
        print(f"{name:45s} {ms:8.3f} ms   {fl / ms / 1e9:6.0f} TFLOP/s (7 products)")


if __name__ == "__main__":
    main()
