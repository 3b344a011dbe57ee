"""Lab: persistent-GEMM launches (one process: the env switch is read once) on multi-tile-per-workgroup shapes, checked
against the default kernel's math (fp32 matmul of the 16-bit operands).  usage: persist_probe.py {qk|proj|fc1|all} [M]
Prints one line per shape: `<name> M=<M>: rel-L2 <err>  non-finite <n>`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptersis_amd import ops

SHAPES = {"qk": (2048, 1024), "proj": (1024, 1024), "fc1": (4096, 1024)}


def run(which, M, dev):
    torch.manual_seed(0)
    N, K = SHAPES[which]
    x = (torch.rand(M, K, device=dev) * 2 - 1).half()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).half()
    b = torch.rand(N, device=dev)
    kw = {}
    if which == "proj":
        kw = dict(out_f32=True, scale_n=torch.rand(N, device=dev), res=torch.rand(M, N, device=dev))
    if which == "fc1":
        kw = dict(act=ops.ACT_GELU)
    o = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if which == "proj" else torch.float16)
    ops.gemm(x, w, out=o, bias_n=b, **kw)
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + b
    if which == "fc1":
        ref = torch.nn.functional.gelu(ref)
    if which == "proj":
        ref = kw["res"] + kw["scale_n"] * ref
    err = float((o.float() - ref).norm() / ref.norm())
    bad = int((~torch.isfinite(o.float())).sum())
    print(f"{which} M={M}: rel-L2 {err:.3e}  non-finite {bad}  PERSIST={os.environ.get('ASIS_GEMM_PERSIST')} LAB={os.environ.get('ASIS_PERSIST_LAB')}")


if __name__ == "__main__":
    which = sys.argv[1]
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    dev = torch.device("cuda:0")
    for name in (SHAPES if which == "all" else [which]):
        run(name, M, dev)
