#!/usr/bin/env python3
"""print a rocprofv3 --stats kernel csv as ms per step: python tools/show_stats.py <csv> <steps> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel time {tot / 1e6 / steps:.3f} ms/step")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step x{int(r['Calls']) / steps:6.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'].replace('void ', '')[:110]}")
