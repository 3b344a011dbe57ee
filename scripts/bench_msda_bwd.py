"""Lab: MSDeformAttn d value at the step's two geometries (ViT-L width, 12 images): taps bucketed by pixel (round 4) vs the dense
sampling matrix + batched GEMMs (round 2).   python scripts/bench_msda_bwd.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops
dev = torch.device("cuda:0")
dt = torch.float16
for Lq, shapes in ((1764, [(73, 73), (36, 36), (18, 18)]), (6949, [(42, 42)])):
    B, M, Dh, P = 12, 8, 128, 4
    L, D = len(shapes), M * Dh
    Lin = sum(a * b for a, b in shapes)
    value = torch.randn(B, Lin, D, device=dev).to(dt)
    offaw = torch.cat([torch.randn(B * Lq, M * L * P * 2, device=dev) * 2.5, torch.randn(B * Lq, M * L * P, device=dev)], 1).contiguous()
    g = torch.arange(Lq, dtype=torch.float32, device=dev)
    ref = torch.stack([(g % 42 + 0.5) / 42 % 1.0, ((g // 42) % 42 + 0.5) / 42], -1)
    dout = torch.randn(B * Lq, D, device=dev) * 1e-3
    starts, acc = [], 0
    for a, b in shapes:
        starts.append(acc); acc += a * b
    sh = torch.tensor(shapes, dtype=torch.int32, device=dev)
    st = torch.tensor(starts, dtype=torch.int32, device=dev)
    res = {}
    for form in (1, 0, 1, 0):
        ops.MSDA_SORTED = bool(form)
        for _ in range(2):
            ops.msda_bwd(value, offaw, ref, sh, st, dout, B, Lq, M, L, P)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.msda_bwd(value, offaw, ref, sh, st, dout, B, Lq, M, L, P)
        e.record(); torch.cuda.synchronize()
        res.setdefault(form, []).append(s.elapsed_time(e) / 5 * 1e3)
    print(f"Lq {Lq} Lin {Lin}: msda_bwd (offsets / weights kernel + d value) sorted {min(res[1]):.0f} us, dense matrix {min(res[0]):.0f} us")
