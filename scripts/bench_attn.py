"""Attention forward microbench (ViT-L shape, the trunk's two stacked passes): python scripts/bench_attn.py
(profiles/r05_attn_fwd_half_ab.txt holds this table for the round-3 kernel and the half-tile rebuild that was measured and dropped)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops


def timeit(f, n=40):
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
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    c = 0.125 * 1.4426950408889634
    qs = (q.float() * c).to(dt)                      # q as the folded projection delivers it
    qks = torch.cat([qs, k], dim=1).contiguous()
    ldvt = 1792
    vt = torch.full((sum(b for b, _ in segs), D, ldvt), float("nan"), device=dev, dtype=dt)   # pad columns: anything
    r0 = b0 = 0
    refs = []
    for B, N in segs:
        vt[b0:b0 + B, :, :N] = v[r0:r0 + B * N].view(B, N, D).transpose(1, 2)
        qf = q[r0:r0 + B * N].float().view(B, N, H, 64).transpose(1, 2)
        kf = k[r0:r0 + B * N].float().view(B, N, H, 64).transpose(1, 2)
        vf = v[r0:r0 + B * N].float().view(B, N, H, 64).transpose(1, 2)
        refs.append((torch.softmax(qf @ kf.transpose(2, 3) * 0.125, -1) @ vf).transpose(1, 2).reshape(B * N, D))
        r0 += B * N
        b0 += B
    ref = torch.cat(refs)
    (B1, N1), (B2, N2) = segs
    o = torch.empty(R, D, device=dev, dtype=dt)
    forms = {
        "unfolded, V^T, stacked": lambda: ops.attention_fwd_seg(q, k, vt, B1, N1, B2, N2, H, 0.125, out=o),
        "pre-scaled q, V^T, stacked (the trunk)": lambda: ops.attention_fwd_seg(qks[:, :D], qks[:, D:], vt, B1, N1, B2, N2, H, None, out=o),
        "unfolded, V rows (qkv), stacked": lambda: ops.attention_fwd_qkv(qkv, segs, H, 0.125, o),
    }
    fl = sum(4.0 * B * H * N * N * 64 for B, N in segs)
    for name, f in forms.items():
        o.zero_()
        f()
        err = float((o.float() - ref).norm() / ref.norm())
        ms = timeit(f)
        print(f"{name:42s} {ms * 1e3:8.1f} us  "
              f"{fl / ms / 1e9:6.0f} TFLOP/s  rel {err:.2e}")


if __name__ == "__main__":
    main()
