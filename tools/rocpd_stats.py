#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this turns its kernel-dispatch table into the per-kernel
summary `--stats` prints (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev) as CSV.
    python tools/rocpd_stats.py gpurun_out/x/prof/t_results.db profiles/r02_train_kernel_stats.csv [--last N-th fraction]"""
import csv
import math
import sqlite3
import sys
import subprocess


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from '{ks}'")}
    rows = cur.execute(f"select kernel_id, start, end from '{kd}' order by start").fetchall()
    if len(sys.argv) > 3 and sys.argv[3] == "--skip-frac":          # drop the leading fraction (warm-up, table building)
        rows = rows[int(len(rows) * float(sys.argv[4])):]
    agg = {}
    for kid, s, e in rows:
        agg.setdefault(names[kid], []).append(e - s)
    dm = demangle([n[:-3] if n.endswith(".kd") else n for n in agg])
    tot = sum(sum(v) for v in agg.values())
    out = []
    for n, v in agg.items():
        k = n[:-3] if n.endswith(".kd") else n
        m = sum(v) / len(v)
        sd = math.sqrt(sum((x - m) ** 2 for x in v) / len(v))
        out.append((dm.get(k, k), len(v), sum(v), round(m, 3), round(100.0 * sum(v) / tot, 2), min(v), max(v), round(sd, 3)))
    out.sort(key=lambda r: -r[2])
    w = csv.writer(open(sys.argv[2], "w"), quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(out)
    for r in out[:45]:
        print(f"{r[4]:6.2f}%  {r[2] / 1e6:9.3f} ms  x{r[1]:5d}  avg {r[3] / 1e3:9.1f} us  {r[0][:150]}")
    print(f"total kernel time {tot / 1e6:.2f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
