"""Weight gradients of the narrow decoder convolutions at the headline geometry (12 images): the halo-tile kernel
(csrc/convwgrad.hip) against the implicit-GEMM form of asis_wgrad, launch + slab reduction.
    python scripts/bench_conv_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
dev = torch.device("cuda:0")
B = 12
for name, Cin, Cout, H in (("decoder_4 (128 -> 64 at 336^2)", 128, 64, 336), ("decoder_3 (256 -> 128 at 168^2)", 256, 128, 168),
                           ("decoder_2 (512 -> 256 at 84^2)", 512, 256, 84), ("decoder_1 (3072 -> 512 at 42^2)", 3072, 512, 42)):
    dy = (torch.rand(B, H, H, Cout, device=dev) * 2 - 1).half()
    x = (torch.rand(B, H, H, Cin, device=dev) * 2 - 1).half()
    res = {}
    for halo in (False, True):
        ops.WGRAD_HALO = 1 if halo else 0
        out = torch.empty(Cout, Cin, 3, 3, device=dev)
        f = lambda: ops.wgrad(dy, x, Cout, 3, 3, 1, 1, 1.0, out=out)
        for _ in range(3):
            f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record(); torch.cuda.synchronize()
        res[halo] = (s.elapsed_time(e) / 20 * 1e3, out.clone())
    ops.WGRAD_HALO = 1
    fl = 2.0 * B * H * H * Cout * Cin * 9
    err = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    print(f"{name:34s} implicit GEMM {res[False][0]:7.1f} us ({fl / res[False][0] / 1e6:5.0f} TFLOP/s)   halo tile {res[True][0]:7.1f} us "
          f"({fl / res[True][0] / 1e6:5.0f} TFLOP/s)   rel-L2 between them {err:.1e}")
