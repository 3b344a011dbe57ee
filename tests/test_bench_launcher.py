"""CPU: ``python bench.py --gpus N`` (the driver's command shape) starts N ranks by itself — rendezvous on 127.0.0.1, one
JSON line from rank 0, exit code relayed — and refuses a --gpus / WORLD_SIZE mismatch.  (The ranks' GPU work is replaced by
a gloo all-reduce through ASIS_BENCH_RANKCHECK: there is no GPU in the build container.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=600)


def test_bench_self_launches_n_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"ASIS_BENCH_RANKCHECK": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j == {"rankcheck": True, "n_gpus": 2, "n_ranks_seen": 2, "sum": 2.0}


def test_bench_refuses_world_size_mismatch():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
