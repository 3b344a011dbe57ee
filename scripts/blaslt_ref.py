"""Reference point only (not used by the product): what torch.matmul (hipBLASLt / rocBLAS) reaches on the ViT-L GEMM shapes."""
import torch, time
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for M in (21168, 42348):
    for name, n, k in (("qk", 2048, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
        a = torch.randn(M, k, device=dev, dtype=torch.float16)
        w = torch.randn(n, k, device=dev, dtype=torch.float16)
        ms = timeit(lambda: torch.matmul(a, w.t()))
        print(f"torch.matmul f16 {name:5s} M={M} N={n} K={k}: {ms:.3f} ms {2.0*M*n*k/ms/1e9:.0f} TFLOP/s")
