"""Build libasis_hip.so (gfx950) in-tree with hipcc.

    python -m adaptersis_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  Objects go to ``adaptersis_amd/csrc/build/`` and the
shared library to ``adaptersis_amd/lib/libasis_hip.so`` (git-ignored, but it travels to the GPU
box with the gpurun snapshot).  Only sources newer than their object are recompiled.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libasis_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build adaptersis_amd)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "asis_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


# per-file additions to FLAGS
# attention.hip: beside MFMAs a packed f32 op costs several single ones (MI355X_MICROARCH.md): keep the softmax arithmetic of the
# folded kernel unpacked
FILE_FLAGS = {"attention.hip": ("-fno-slp-vectorize",), "attn_bwd_pipe.hip": ("-fno-slp-vectorize",)}


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    cmd = [_hipcc(), *FLAGS, *FILE_FLAGS.get(os.path.basename(src), ()), "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Idempotent and safe to call from several ranks at once: an exclusive file lock serialises the builders (the first
    one compiles, the others find everything up to date)."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    dep_t = _deps_mtime()
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_t):
            todo.append(src)
    if todo:
        workers = min(len(todo), max(1, (os.cpu_count() or 4) // 2))
        with cf.ThreadPoolExecutor(workers) as ex:
            list(ex.map(lambda s: _compile(s, verbose), todo))
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build_library(a.force, a.verbose))


if __name__ == "__main__":
    sys.exit(main())
