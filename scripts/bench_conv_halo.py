"""Narrow MX split convolutions of the FeatureDecoder at the headline batch (`decoders.py:109-135`: d3 256 -> 128 at 168^2, d4 128 -> 64
at 336^2): implicit-GEMM form (csrc/gemm_big.h MX instances) vs the halo-tile kernel (csrc/convhalo.hip), interleaved rounds.
    python scripts/bench_conv_halo.py [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
from adaptersis_amd.utils import weights as W

dev = torch.device("cuda:0")
dt = torch.float16
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, B, H, Cin, Cout in (("d3", 12, 168, 256, 128), ("d4", 12, 336, 128, 64)):
    x = torch.relu(torch.randn(B, H, H, Cin, device=dev) + 0.3)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.03)
    bias = torch.randn(Cout, device=dev)
    amax_x = ops.absmax_f32(x.view(-1, Cin))
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    x_hi, x_mx = ops.bn_relu_upsample(x, one, zero, 1, dt, True, mx_amax=amax_x)
    w_hi = ops.pack_conv_weight(w, 0, dt)
    w_mx, amax_w = ops.pack_conv_weight_mx(w, 0, dt)
    stats = torch.empty((ops.gemm_tiles_m(B * H * H), 2, Cout), device=dev, dtype=torch.float32)
    fns = {"implicit GEMM": lambda: ops.conv_gemm_split(x_hi, x_mx, w_hi, w_mx, 3, 3, 1, 1, bias_n=bias, stats=stats, mx=(amax_x, amax_w)),
           "halo tile": lambda: ops.conv3x3_halo_mx(x_hi, x_mx, w_hi, w_mx, (amax_x, amax_w), bias_n=bias, want_stats=True)}
    res = {k: [] for k in fns}
    for k, f in fns.items():
        for _ in range(5):
            f()
    for _ in range(R):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                f()
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10 * 1e3)
    gf = 2.0 * B * H * H * Cout * 9 * Cin / 1e9
    print(f"{name} ({Cin} -> {Cout} @ {H}^2, {gf:.0f} GF algorithmic): " + "   ".join(
        f"{k} {statistics.median(v):7.1f} us ({gf / statistics.median(v) * 1e3:5.0f} TF/s)" for k, v in res.items()))
