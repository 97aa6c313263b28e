#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as a per-kernel table.
usage: tools/prof_summary.py gpurun_out/<dir>/<name>_results.db [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
lines = [f"{'kernel':<86} {'calls':>8} {'total_us':>12} {'avg_us':>9} {'%':>6}"]
for name, calls, total, avg, pct in rows:
    lines.append(f"{name[:86]:<86} {calls:>8} {total:>12.1f} {avg:>9.2f} {pct:>6.2f}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
