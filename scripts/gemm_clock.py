"""Shader clock while the dense GEMM main loop runs back to back (lab build path: ASIS_GEMM_NOEPI=1)."""
import os, sys
os.environ["ASIS_GEMM_NOEPI"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops

dev = torch.device("cuda:0")
for (M, N, K) in ((42336, 1024, 4096), (42336, 4096, 1024)):
    x = (torch.rand(M, K, device=dev) * 2 - 1).half()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).half()
    out = torch.zeros(M, N, device=dev, dtype=torch.float16)
    for iters in (3, 300):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops.gemm(x, w, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        raw = out.view(-1)[:8].view(torch.int64).cpu()
        print(f"M={M} N={N} K={K} after {iters} launches: main loop {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s, one workgroup "
              f"{raw[0].item()} shader ticks in {raw[1].item() / 100:.1f} us -> {raw[0].item() / max(raw[1].item(), 1) * 100:.0f} MHz")
