#!/usr/bin/env python3
"""Per-kernel summary (the columns of rocprofv3 --stats) from a --kernel-trace csv, dropping the leading fraction of the dispatches
(model building, warm-up steps): python tools/trace_stats.py <dir> <out.csv> [skip_fraction | step:<kernel substring>:<n>]
step:adam_kernel:5 keeps the dispatches of the last 5 steps, a step ending with its (single) launch of that kernel."""
import csv, glob, math, re, sys


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    arg = sys.argv[3] if len(sys.argv) > 3 else "0"
    if arg.startswith("step:"):
        _, key, n = arg.split(":")
        ends = [i for i, r in enumerate(rows) if key in r[2]]
        n = int(n)
        assert len(ends) > n, (len(ends), n)
        skip = (ends[-n - 1] + 1) / len(rows)
        rows = rows[ends[-n - 1] + 1:ends[-1] + 1]
    else:
        skip = float(arg)
        rows = rows[int(len(rows) * skip):]
    agg = {}
    for s, e, n in rows:
        agg.setdefault(n, []).append(e - s)
    tot = sum(sum(v) for v in agg.values())
    out = []
    for n, v in agg.items():
        m = sum(v) / len(v)
        out.append((n, len(v), sum(v), round(m, 3), round(100.0 * sum(v) / tot, 2), min(v), max(v), round(math.sqrt(sum((x - m) ** 2 for x in v) / len(v)), 3)))
    out.sort(key=lambda r: -r[2])
    w = csv.writer(open(sys.argv[2], "w"), quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(out)
    for r in out[:40]:
        print(f"{r[4]:6.2f}%  {r[2] / 1e6:9.3f} ms  x{r[1]:5d}  avg {r[3] / 1e3:9.1f} us  {re.sub(r'^void ', '', r[0])[:140]}")
    print(f"total kernel time {tot / 1e6:.2f} ms over {len(rows)} dispatches (leading {skip:.0%} dropped)")


if __name__ == "__main__":
    main()
