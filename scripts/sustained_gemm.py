"""Does the dense GEMM hold its burst rate when it runs back to back for ~1 s (power / clock behaviour)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
dev = torch.device("cuda:0")
M, N, K = 42336, 1024, 4096
x = (torch.rand(M, K, device=dev) * 2 - 1).half(); w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).half()
res = torch.rand(M, N, device=dev); sc = torch.rand(N, device=dev); b = torch.rand(N, device=dev)
out = torch.empty(M, N, device=dev)
def run(iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): ops.gemm(x, w, out=out, bias_n=b, scale_n=sc, res=res, out_f32=True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
run(5)
for iters in (10, 100, 1000, 3000, 10):
    ms = run(iters)
    print(f"iters {iters:5d}: {ms:.3f} ms/launch {2.0*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
