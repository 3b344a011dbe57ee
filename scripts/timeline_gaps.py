"""Idle time and concurrency of the step from a rocprofv3 --kernel-trace database (results.db):
   python scripts/timeline_gaps.py <results.db> [n_steps] [first_step]
(bench.py's default run: warmup 3 + 10 timed steps, THEN a single-stream pass of the same steps for the roofline launch times —
first_step 4, n_steps 8 looks at the timed region).  The step boundary is the guarded-SGD launch; for the last n steps: wall time, union-busy time (any kernel running), idle time,
per-queue busy time, and the largest idle gaps with the kernels on either side."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0][:48]


def main():
    db = sqlite3.connect(sys.argv[1])
    nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    sgd = [i for i, r in enumerate(rows) if "sgd" in r[0]]
    if len(sgd) < nlast + 1:
        print("not enough steps", len(sgd)); return
    first = int(sys.argv[3]) if len(sys.argv) > 3 else len(sgd) - nlast - 1
    lo, hi = sgd[first], sgd[first + nlast]
    seg = rows[lo + 1:hi + 1]
    t0, t1 = rows[lo][2], rows[hi][2]
    wall = (t1 - t0) / 1e6
    # union
    busy, cur_s, cur_e, gaps, prev = 0, None, None, [], None
    for r in sorted(seg, key=lambda r: r[1]):
        s, e = max(r[1], t0), r[2]
        if cur_e is None:
            cur_s, cur_e, last = s, e, r
            if s > t0: gaps.append((s - t0, "step start", r[0]))
        elif s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last[0], r[0]))
            cur_s, cur_e, last = s, e, r
        elif e > cur_e:
            cur_e, last = e, r
    busy += cur_e - cur_s
    ksum = sum(r[2] - r[1] for r in seg)
    print(f"{nlast} steps: wall {wall / nlast:.2f} ms/step, some kernel running {busy / 1e6 / nlast:.2f} ms/step, idle {(wall - busy / 1e6) / nlast:.2f} ms/step, "
          f"kernel-time sum {ksum / 1e6 / nlast:.2f} ms/step (concurrency {ksum / busy:.2f})")
    perq = {}
    for r in seg:
        perq.setdefault((r[3], r[4]), [0, 0])
        perq[(r[3], r[4])][0] += r[2] - r[1]
        perq[(r[3], r[4])][1] += 1
    for k, v in sorted(perq.items(), key=lambda kv: -kv[1][0]):
        print(f"  queue {k[0]} stream {k[1]}: {v[0] / 1e6 / nlast:7.2f} ms/step in {v[1] / nlast:6.1f} launches/step")
    gaps.sort(reverse=True)
    tot = sum(g[0] for g in gaps)
    print(f"idle gaps: {len(gaps) / nlast:.0f}/step, total {tot / 1e6 / nlast:.2f} ms/step; by size:")
    for lim in (2e3, 5e3, 1e4, 2e4, 5e4, 1e9):
        sel = [g[0] for g in gaps if g[0] < lim]
        print(f"   < {lim / 1e3:8.0f} us: {len(sel) / nlast:7.1f}/step {sum(sel) / 1e6 / nlast:6.2f} ms/step")
    print("largest:")
    for g in gaps[:25]:
        print(f"   {g[0] / 1e3:8.1f} us  after {short(g[1]):48s} before {short(g[2])}")
    # by the kernel that follows the gap
    agg = {}
    for g in gaps:
        a = agg.setdefault(short(g[2]), [0, 0]); a[0] += g[0]; a[1] += 1
    print("idle by the kernel that follows:")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:20]:
        print(f"   {v[0] / 1e6 / nlast:6.2f} ms/step  {v[1] / nlast:6.1f}/step  avg {v[0] / v[1] / 1e3:6.1f} us  {k}")


if __name__ == "__main__":
    main()
