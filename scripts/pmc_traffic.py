"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, see MI355X_MICROARCH.md "HBM" and
"rocprofv3 PMC slots") of `bench.py` into per-kernel HBM traffic per launch and write profiles/<round>_pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 ...
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units and the gfx950 correction follow the guide: both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of
wide coalesced streaming reads at 64 bytes, so the read side is doubled (all big readers here use 16-byte-per-lane
global loads or LDS-DMA); WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gemm_sources_sha256():
    """hash of the dense-GEMM kernel sources the traffic figure belongs to (bench.py refuses a figure whose hash differs from the
    tree it runs in: the GPU box has no .git to ask)"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "adaptersis_amd", "csrc", "gemm*"))):
        if f.endswith((".h", ".hip")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").strip()


def main():
    fetch, write, out = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else None          # git commit of the profiled build
    command = sys.argv[5] if len(sys.argv) > 5 else None
    F, Wr = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    rows = {}
    for k in F:
        f = F[k]
        w = Wr.get(k, [0.0])
        rows[short(k)] = {
            "launches": len(f),
            "fetch_kib_raw_avg": sum(f) / len(f),
            "write_kib_avg": sum(w) / len(w),
            "hbm_bytes_per_launch": (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0,
            "hbm_bytes_total": (2.0 * sum(f) + sum(w)) * 1024.0,
        }
    rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_total"]))
    json.dump({"note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per the guide's gfx950 correction; averages over all "
                       "launches of the kernel in the profiled bench.py run", "commit": commit, "command": command,
               "gemm_sources_sha256": gemm_sources_sha256(),
               "kernels": rows}, open(out, "w"), indent=1)
    for k, v in list(rows.items())[:25]:
        print(f"{v['launches']:6d}  {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  {v['hbm_bytes_total'] / 1e9:8.2f} GB  {k[:110]}")


if __name__ == "__main__":
    main()
