"""Layernorm microbench: python scripts/bench_ln.py  (ASIS_LN_FAST=0|1|2 selects the kernel form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops


def main():
    dev = torch.device("cuda:0")
    for M, D in ((42348, 1024), (21180, 1024), (21168, 1536)):
        x = torch.randn(M, D, device=dev)
        w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
        y = torch.empty(M, D, device=dev, dtype=torch.float16)
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)
        ops.layernorm(x, w, b, 1e-6, torch.float16, out=y)
        err = ((y.float() - ref).norm() / ref.norm()).item()
        for _ in range(5):
            ops.layernorm(x, w, b, 1e-6, torch.float16, out=y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            ops.layernorm(x, w, b, 1e-6, torch.float16, out=y)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"LN_FAST={os.environ.get('ASIS_LN_FAST', '2')} M={M} D={D}: {ms * 1e3:.1f} us  {M * D * 6 / ms / 1e6:.0f} GB/s  rel {err:.2e}")


if __name__ == "__main__":
    main()
