#!/usr/bin/env python3
"""Launch ORDER of the kernels of the last step in a rocprofv3 --kernel-trace csv: who launches what between the hand-written
kernels (copyBuffer / fill / cast launches dispatched by torch are attributed to the call site that follows or precedes them).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x/tr -- python3 bench.py --steps 1 --warmup 1 --graph 0 ...
    python tools/trace_order.py gpurun_out/x/tr [n_last]"""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    n = n.replace("mhe::conv::", "").replace("mhe::", "").replace("at::native::", "at::")
    return n[:110]


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rows = rows[-n_last:]
    prev, cnt = None, 0
    for s, e, n in rows:
        print(f"{(e - s) / 1e3:8.1f} us  {short(n)}")


if __name__ == "__main__":
    main()
