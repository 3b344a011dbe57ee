"""The classifier conv (64 -> 2 at 12 x 672 x 672) in isolation: forward / input gradient / weight gradient, us per launch.
ASIS_SMALLCOUT_ROW=0|1 selects the tap-gather or the row-walk form (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
dev = torch.device("cuda:0")
B, H, C = 12, 672, 64
x = torch.randn(B * H * H, C, device=dev)
xh, xl = ops.cast_pad(x, C, torch.float16).view(B, H, H, C), ops.cast_pad(x, C, torch.float16, part=1).view(B, H, H, C)
w = torch.randn(2, C, 3, 3, device=dev) * 0.1
bias = torch.zeros(2, device=dev)
d = torch.randn(B * H * H, 2, device=dev)
dh, dl = ops.cast_pad(d, 8, torch.float16).view(B, H, H, 8), ops.cast_pad(d, 8, torch.float16, part=1).view(B, H, H, 8)

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

print("ASIS_SMALLCOUT_ROW =", os.environ.get("ASIS_SMALLCOUT_ROW", "(default)"))
print(f"fwd   {t(lambda: ops.conv3x3_smallcout_fwd(xh, xl, w, bias)):8.1f} us")
print(f"dgrad {t(lambda: ops.conv3x3_smallcout_dgrad(dh, dl, w)):8.1f} us")
out = torch.empty_like(w)
print(f"wgrad {t(lambda: ops.wgrad(dh, xh, 2, 3, 3, 1, 1, 1.0, out=out)):8.1f} us")
