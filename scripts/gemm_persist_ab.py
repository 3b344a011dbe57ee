"""Persistent dense GEMM (ASIS_GEMM_PERSIST) against the default forms on the stacked ViT-L shapes: whole kernel and main loop."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("gemm_epi", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_epi.py"))
ge = importlib.util.module_from_spec(spec); spec.loader.exec_module(ge)
if __name__ == "__main__":
    res = {}
    for tag, env in (("default", {}), ("default main", {"ASIS_GEMM_NOEPI": "1"}), ("persist", {"ASIS_GEMM_PERSIST": sys.argv[1] if len(sys.argv) > 1 else "2"}),
                     ("persist main", {"ASIS_GEMM_PERSIST": sys.argv[1] if len(sys.argv) > 1 else "2", "ASIS_GEMM_NOEPI": "1"})):
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_epi.py"), "child"],
                           env={**os.environ, **env}, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(tag, "FAILED", r.stderr[-800:]); continue
        res[tag] = json.loads(lines[-1])
    for name, M, N, K, _ in ge.SHAPES:
        print(f"{name:12s} " + "  ".join(f"{t}: {res[t][name]:7.1f} us ({2.0 * M * N * K / res[t][name] / 1e6:5.0f} TF)" for t in res))
