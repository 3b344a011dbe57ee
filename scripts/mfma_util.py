"""Per-kernel effective clock and matrix-pipe utilisation from ONE rocprofv3 PMC pass of `bench.py --single-stream`:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r04/pmc_mfma -- python3 bench.py ...
    python scripts/mfma_util.py gpurun_out/r04/pmc_mfma profiles/r04_mfma_util.txt

MI355X_MICROARCH.md ("DVFS give-back", "s_memtime tick vs SQ PMC units"): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
effective shader clock of a dispatch is GRBM_GUI_ACTIVE / 8 / wall time (reads high on dispatches under ~0.3 ms);
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs (16 per v_mfma_f32_16x16x32_f16), so
utilisation in CYCLES = busy / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).  The nominal 2.5 PFLOP/s peak assumes 2.4 GHz: a
kernel at clock f and utilisation u delivers u * f / 2.4 of it."""
import csv
import glob
import re
import sys
from collections import defaultdict

CUS, SIMDS, XCDS = 256, 4, 8


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").strip()
    m = re.search(r"(gemm_p8_kernel|gemm_big_kernel|attn_fwd_pipe_kernel|wgrad_dense_big_kernel|wgrad_kernel|smallcout_\w+_kernel)(.*)", n)
    return (m.group(1) + m.group(2))[:110] if m else n[:110]


def main():
    d, out = sys.argv[1:3]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], r["Dispatch_Id"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_ns"] = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])]
    per = defaultdict(list)
    for (name, _), c in acc.items():
        if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        gui, busy, ns = sum(c["GRBM_GUI_ACTIVE"]), sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), c["_ns"][0]
        if gui <= 0 or ns <= 0:
            continue
        per[short(name)].append((ns, gui / XCDS / ns, busy / (gui / XCDS * CUS * SIMDS), busy))
    rows = sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1]))
    with open(out, "w") as f:
        f.write("# kernel | dispatches | avg us (PMC pass: serialised, slower than the trace) | effective clock GHz | matrix-pipe busy "
                "fraction (cycles) | busy cycles per dispatch\n")
        for name, v in rows[:24]:
            n = len(v)
            f.write(f"{name:<112} {n:5d} {sum(x[0] for x in v) / n / 1e3:9.1f} {sum(x[1] for x in v) / n:6.3f} "
                    f"{sum(x[2] for x in v) / n:6.3f} {sum(x[3] for x in v) / n:14.0f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
