"""Lab: the FeatureDecoder's four split 3x3 convolutions at the headline batch (12 x 42^2 x 3072 -> 512, 84^2 x 512 -> 256,
168^2 x 256 -> 128, 336^2 x 128 -> 64): three 16-bit parts vs 16-bit + one MX (block-scaled fp8) correction pass, interleaved
rounds in one process.    python scripts/bench_conv_mx.py [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
from adaptersis_amd.backbones.decoders import _conv_ksplit

dev = torch.device("cuda:0")
dt = torch.float16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
LAYERS = [("d1", 42, 3072, 512), ("d2", 84, 512, 256), ("d3", 168, 256, 128), ("d4", 336, 128, 64)]
fs = {}
for name, H, Cin, Cout in LAYERS:
    x = torch.relu(torch.randn(12, H, H, Cin, device=dev))
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
    x2 = x.view(-1, Cin)
    xh, xl = ops.cast_pad(x2, Cin, dt).view(x.shape), ops.cast_pad(x2, Cin, dt, part=1).view(x.shape)
    wh, wl = ops.pack_conv_weight(w, 0, dt), ops.pack_conv_weight(w, 0, dt, 1)
    ax = ops.absmax_f32(x2)
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    _, xm = ops.bn_relu_upsample(x, one, zero, 1, dt, True, mx_amax=ax)
    wm, aw = ops.pack_conv_weight_mx(w, 0, dt)
    ks = _conv_ksplit(12 * H * H, Cout, Cin, True)
    fl = 2.0 * 12 * H * H * Cout * 9 * Cin
    fs[name] = (lambda xh=xh, xl=xl, wh=wh, wl=wl, ks=ks: ops.conv_gemm_split(xh, xl, wh, wl, 3, 3, 1, 1, ksplit=ks),
                lambda xh=xh, xm=xm, wh=wh, wm=wm, ks=ks, ax=ax, aw=aw: ops.conv_gemm_split(xh, xm, wh, wm, 3, 3, 1, 1, ksplit=ks, mx=(ax, aw)),
                fl, ks)
    del x, w
res = {(n, v): [] for n in fs for v in (0, 1)}
for n, f in fs.items():
    for v in (0, 1):
        for _ in range(3):
            f[v]()
for r in range(rounds):
    for n, f in fs.items():
        for v in (0, 1):
            f[v]()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                f[v]()
            e.record()
            torch.cuda.synchronize()
            res[(n, v)].append(s.elapsed_time(e) / 5 * 1e3)
for n, f in fs.items():
    m0, m1 = statistics.median(res[(n, 0)]), statistics.median(res[(n, 1)])
    print(f"{n} (ksplit {f[3]}): three 16-bit parts {m0:8.1f} us ({f[2] / m0 / 1e6:5.0f} TF/s algorithmic)   16-bit + MX {m1:8.1f} us "
          f"({f[2] / m1 / 1e6:5.0f} TF/s)   ratio {m1 / m0:.3f}")
