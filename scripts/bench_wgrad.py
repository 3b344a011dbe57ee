"""nn.Linear weight gradients of one stacked ViT-L block (M = 42348 rows): the wgrad launch alone and with its slab reduction.
    ASIS_WGRAD_BIG=0|1 python scripts/bench_wgrad.py     (the switch is read once per process)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
dev = torch.device("cuda:0")
M = 42348
for name, Cout, Cin in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    dy = (torch.rand(1, M, 1, Cout, device=dev) * 2 - 1).half()
    x = (torch.rand(1, M, 1, Cin, device=dev) * 2 - 1).half()
    out = torch.empty(Cout, Cin, 1, 1, device=dev)
    f = lambda: ops.wgrad(dy, x, Cout, 1, 1, 1, 0, 1.0, out=out)
    for _ in range(5):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    ref = dy.view(M, Cout)[:4096].float().t() @ x.view(M, Cin)[:4096].float()
    chk = ops.wgrad(dy[:, :4096].contiguous(), x[:, :4096].contiguous(), Cout, 1, 1, 1, 0, 1.0).view(Cout, Cin)
    err = float((chk - ref).norm() / ref.norm())
    print(f"{name:5s} dW[{Cout},{Cin}] over {M} rows: {us:7.1f} us incl. slab reduce = {2.0 * M * Cout * Cin / us / 1e6:6.0f} TFLOP/s   (4096-row check rel-L2 {err:.1e}, "
          f"splits {ops.lib().asis_wgrad_splits(M, Cout, Cin)})")
