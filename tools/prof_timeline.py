#!/usr/bin/env python3
"""Per-kernel duration AND the gap to the previous kernel from a rocprofv3 --kernel-trace database (rocpd sqlite):
for every kernel name: calls, mean/min duration, mean gap (start - previous end) — the inter-kernel boundary cost.
usage: tools/prof_timeline.py <results.db> [--last N] [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in names else None
if view is None:
    print("tables/views:", names)
    sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
if "--schema" in sys.argv:
    print(cols)
rows = list(db.execute(f"select name, start, end from {view} order by start"))
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else len(rows)
rows = rows[-last:]
agg = {}
prev_end = None
for name, st, en in rows:
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0, 0])
    a[0] += 1
    a[1] += (en - st) / 1e3
    a[2] = min(a[2], (en - st) / 1e3)
    if prev_end is not None and st - prev_end < 50_000:      # ignore host-side stalls (> 50 us)
        a[3] += (st - prev_end) / 1e3
        a[4] += 1
    prev_end = en
lines = [f"{'kernel':<70} {'calls':>7} {'avg_us':>8} {'min_us':>8} {'gap_before_us':>14} {'total_us':>10}"]
for name, (n, tot, mn, gap, ng) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"{name[:70]:<70} {n:>7} {tot / n:>8.2f} {mn:>8.2f} {(gap / ng if ng else 0):>14.2f} {tot:>10.1f}")
span = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum(a[1] for a in agg.values())
lines.append(f"span {span:.1f} us, kernel-busy {busy:.1f} us ({100 * busy / span:.1f} %), {len(rows)} dispatches")
txt = "\n".join(lines)
print(txt)
outs = [a for a in sys.argv[2:] if a.endswith(".txt")]
if outs:
    open(outs[0], "w").write(txt + "\n")
