"""rocprofv3 (ROCm 7.2 default output = rocpd sqlite) -> the per-kernel summary CSV kept under profiles/:
    python scripts/kernel_stats_from_db.py gpurun_out/prof_x/<host>/<pid>_results.db profiles/rNN_name_kernel_stats.csv [steps]
Columns as `--stats` prints them (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs)."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
rows = c.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                 f"from kernels group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(100.0 * r[2] / tot, 3), int(r[4]), int(r[5])])
print(f"{len(rows)} kernels, {tot / 1e6:.2f} ms of kernel time -> {out}")
