"""Attention backward microbench (ViT-L shape, config 4's two stacked passes): python scripts/bench_attn_bwd.py
old = asis_attention_bwd (transposed operand images, two launches sets) ; new = asis_attention_bwd_rows"""
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
    out_old = torch.empty(R, 3 * D, device=dev, dtype=dt)
    out_new = torch.empty(R, 3 * D, device=dev, dtype=dt)

    def old():
        for (B, N), (a, b, l) in zip(segs, views):
            ops.attention_bwd(q[a:b], k[a:b], v[a:b], ops.transpose_tokens(q[a:b], B, N), ops.transpose_tokens(k[a:b], B, N),
                              ops.transpose_tokens(dO[a:b], B, N), o[a:b], dO[a:b], l, B, H, N, 0.125, dqkv=out_old[a:b])

    def new():
        ops.attention_bwd_rows(q, k, v, o, dO, lse, segs, H, 0.125, dqkv=out_new)

    def new_sep():
        l0 = 0
        for (B, N), (a, b, l) in zip(segs, views):
            ops.attention_bwd_rows(q[a:b], k[a:b], v[a:b], o[a:b], dO[a:b], l.reshape(-1), [(B, N)], H, 0.125, dqkv=out_new[a:b])

    if os.environ.get("ASIS_PMC"):   # counter passes: three launches of the new form only
        for _ in range(3):
            new()
        torch.cuda.synchronize()
        return
    old(); new()
    torch.cuda.synchronize()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, b = out_old[:, sl].float(), out_new[:, sl].float()
        print(f"{name}: new vs old rel-L2 {float((a - b).norm() / a.norm()):.2e}  finite {bool(torch.isfinite(b).all())}")
    fl = sum(14.0 * B * H * N * N * 64 for B, N in segs)   # 7 products
    for name, f in (("old (incl. 3 transposes + rowdot per batch)", old), ("new stacked", new), ("new per batch", new_sep)):
        ms = timeit(f)
        print(f"{name:45s} {ms:8.3f} ms   {fl / ms / 1e9:6.0f} TFLOP/s (7 products)")


if __name__ == "__main__":
    main()
