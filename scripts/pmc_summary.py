"""Per-kernel averages of every counter found in a rocprofv3 --pmc output directory (one or more passes):
    python scripts/pmc_summary.py gpurun_out/pmc_x [substring filter] > profiles/rNN_x.txt
Columns: kernel | dispatches | avg us | counter = average per dispatch ...  (SQ_* tick in quad-cycles except
SQ_VALU_MFMA_BUSY_CYCLES, MI355X_MICROARCH.md)"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").strip()
    n = re.sub(r"^_ZN\d+_GLOBAL__N_1(\d+_GLOBAL__N_1)?\d+", "", n)
    return n[:70]


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    ns = defaultdict(dict)
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if filt and filt not in k:
                continue
            c = r["Counter_Name"]
            acc[k][c] += float(r["Counter_Value"])
            disp[k][c].add((f, r["Dispatch_Id"]))
            ns[k][(f, r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k in sorted(acc, key=lambda k: -sum(ns[k].values())):
        t = sum(ns[k].values()) / max(len(ns[k]), 1) / 1e3
        parts = [f"{c}={acc[k][c] / max(len(disp[k][c]), 1):.4g}" for c in sorted(acc[k])]
        print(f"{k:<72} n={len(ns[k]):4d} avg {t:8.1f} us  " + "  ".join(parts))


if __name__ == "__main__":
    main()
