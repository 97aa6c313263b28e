#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc database.  usage: tools/pmc_kernel.py <results.db> [substr]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if sub and sub not in k:
        continue
    a = agg.setdefault(k[:60], {})
    a[c] = a.get(c, 0.0) + float(v)
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {v:16.0f}")
