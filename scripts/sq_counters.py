"""Per-kernel SQ wave-cycle breakdown from one rocprofv3 PMC pass (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ_WAIT_ANY = wave
parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and sum to
SQ_WAVE_CYCLES; SQ counters tick in quad-cycles):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \\
        --output-format csv -d gpurun_out/r04/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --single-stream
    python scripts/sq_counters.py gpurun_out/r04/pmc_sq profiles/r04_sq_counters.txt"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").strip()
    m = re.search(r"(gemm_p8_kernel|gemm_big_kernel|attn_fwd_pipe_kernel|wgrad_dense_big_kernel|wgrad_kernel|smallcout_\w+_kernel|msda_\w+_kernel)(.*)", n)
    return (m.group(1) + m.group(2))[:100] if m else n[:100]


def main():
    d, out = sys.argv[1:3]
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    ns = defaultdict(float)
    seen = set()
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
                ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = sorted(acc.items(), key=lambda kv: -ns[kv[0]])
    with open(out, "w") as f:
        f.write("# kernel | dispatches | avg us (PMC pass) | of SQ_WAVE_CYCLES: issuing / issue-stalled / parked | VALU instructions per dispatch (M)\n")
        for k, c in rows[:16]:
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            if wc <= 0:
                continue
            f.write(f"{k:<102} {cnt[k]:5d} {ns[k] / cnt[k] / 1e3:9.1f}   {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.2f} / "
                    f"{c.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f} / {c.get('SQ_WAIT_ANY', 0) / wc:5.2f}   {c.get('SQ_INSTS_VALU', 0) / cnt[k] / 1e6:9.1f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
