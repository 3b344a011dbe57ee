"""SGD + overflow guard on one flat bucket: the 16-byte kernels (aligned) against the scalar ones (a view one element in).
    python scripts/bench_sgd.py [n_params]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
for off, name in ((0, "16-byte"), (1, "scalar")):
    p, g, b = (torch.rand(n + 4, device=dev)[off:off + n] for _ in range(3))
    guard = torch.zeros(2, device=dev, dtype=torch.int32)
    def run():
        ops.grad_guard(g, guard, True)
        ops.sgd_momentum(p, g, b, 1e-3, 0.9, 1e-5, 1.0, False, guard)
    for f, what, nbytes in ((lambda: ops.grad_guard(g, guard, True), "guard", 4.0 * n),
                            (lambda: ops.sgd_momentum(p, g, b, 1e-3, 0.9, 1e-5, 1.0, False, guard), "sgd", 20.0 * n)):
        for _ in range(3):
            f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        print(f"{name:8s} {what:6s} n={n}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s")
