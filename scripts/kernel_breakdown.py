"""per-kernel (and per-grid) time of the last n steps of a rocprofv3 --kernel-trace database: python scripts/kernel_breakdown.py <results.db> [n]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = db.execute("select name,start,end,grid_x,grid_y,grid_z from kernels order by start").fetchall()
    sgd = [i for i, r in enumerate(rows) if "sgd" in r[0]]
    # several buckets -> several sgd launches per step: a step ends at the LAST sgd launch of a run of consecutive ones
    ends = [i for j, i in enumerate(sgd) if j + 1 == len(sgd) or not all("sgd" in rows[k][0] or "nonfinite" in rows[k][0] for k in range(i + 1, sgd[j + 1] + 1))]
    if len(ends) < n + 1:
        print("not enough steps:", len(ends)); return
    lo, hi = ends[-n - 1], ends[-1]
    seg = rows[lo + 1:hi + 1]
    wall = (rows[hi][2] - rows[lo][2]) / 1e6 / n
    agg = {}
    for r in seg:
        nm = re.sub(r"_ZN12_GLOBAL__N_1(12_GLOBAL__N_1)?\d+", "", r[0]).replace("(anonymous namespace)::", "").replace("void ", "")
        key = (nm[:60], r[3], r[4], r[5])
        a = agg.setdefault(key, [0, 0]); a[0] += r[2] - r[1]; a[1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"{n} steps: wall {wall:.2f} ms/step, kernel time {tot / 1e6 / n:.2f} ms/step, {sum(v[1] for v in agg.values()) / n:.0f} launches/step")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
        print(f"{k[0]:60s} grid {k[1]:>8}x{k[2]}x{k[3]:<3} {v[1] / n:6.1f}/step avg {v[0] / v[1] / 1e3:8.1f} us {v[0] / 1e6 / n:7.2f} ms/step {100 * v[0] / tot:5.1f}%")


if __name__ == "__main__":
    main()
