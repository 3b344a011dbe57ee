"""Whole dense GEMM vs its main loop alone (ASIS_GEMM_NOEPI=1 in a second process) at the stacked ViT-L shapes."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [("qk", 42348, 2048, 1024, {}), ("proj+ls+res", 42348, 1024, 1024, {"res": 1}), ("fc1+gelu", 42348, 4096, 1024, {"act": 1}),
          ("fc2+ls+res", 42348, 1024, 4096, {"res": 1})]


def run():
    import torch
    from adaptersis_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for name, M, N, K, kw in SHAPES:
        x = (torch.rand(M, K, device=dev) * 2 - 1).half()
        w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).half()
        b = torch.rand(N, device=dev)
        extra = {}
        if kw.get("res"):
            extra = dict(out_f32=True, scale_n=torch.rand(N, device=dev), res=torch.rand(M, N, device=dev))
        if kw.get("act"):
            extra = dict(act=ops.ACT_GELU)
        o = torch.empty(M, N, device=dev, dtype=torch.float32 if kw.get("res") else torch.float16)
        f = lambda: ops.gemm(x, w, out=o, bias_n=b, **extra)
        for _ in range(20):
            f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100):
            f()
        e.record()
        torch.cuda.synchronize()
        out[name] = s.elapsed_time(e) / 100 * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        res = {}
        variants = (("full", {}), ("mainloop", {"ASIS_GEMM_NOEPI": "1"}), ("nostore", {"ASIS_GEMM_NOEPI": "8"}),
                    ("nores", {"ASIS_GEMM_NOEPI": "32"}), ("nostore_nores", {"ASIS_GEMM_NOEPI": "40"}))
        for tag, env in variants:
            r = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True)
            res[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        for name, M, N, K, _ in SHAPES:
            a, b = res["full"][name], res["mainloop"][name]
            if name != "fc2+ls+res":  # the lab variants exist for the default (K < 2048) form only
                print(f"{name:12s}   lab (us): no global stores {res['nostore'][name]:.1f}, no residual fetch {res['nores'][name]:.1f}, "
                      f"neither {res['nostore_nores'][name]:.1f}")
            print(f"{name:12s} M={M} N={N} K={K}: full {a:7.1f} us ({2.0 * M * N * K / a / 1e6:6.0f} TFLOP/s)  main loop {b:7.1f} us "
                  f"({2.0 * M * N * K / b / 1e6:6.0f})  epilogue share {100 * (a - b) / a:4.1f} %")
