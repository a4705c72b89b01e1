#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of bench.py) into
HBM bytes per launch per kernel.  Units and gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md
section HBM: the counters are in KiB; FETCH_SIZE reports exactly half of the bytes of wide coalesced
streaming reads on gfx950, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 ... (and WRITE_SIZE)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic.json
The summary records the sha1 of the convolution sources it was collected on; bench.py reports `traffic` only while they match.
"""
import collections
import csv
import hashlib
import os
import glob
import json
import re
import sys


def agg(d):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            out[k][0] += float(r["Counter_Value"])
            out[k][1] += 1
    return out


def kernel_sources_sha1():
    """sha1 over every kernel source of the library (csrc/*.hip, csrc/*.h, sorted by name): the counters describe THESE kernels"""
    cs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mhentropy_amd", "csrc")
    names = sorted(f for f in os.listdir(cs) if f.endswith(".hip") or f.endswith(".h"))
    return hashlib.sha1(b"".join(open(os.path.join(cs, f), "rb").read() for f in names)).hexdigest()


def main():
    fetch, write, dst = agg(sys.argv[1]), agg(sys.argv[2]), sys.argv[3]
    res = {}
    for k in fetch:
        nf, nw = fetch[k][1], max(write[k][1], 1)
        rd = fetch[k][0] / nf * 1024 * 2          # KiB -> B, x2 gfx950 correction
        wr = write[k][0] / nw * 1024
        res[k] = {"launches_profiled": nf, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr)}
    sha = kernel_sources_sha1()
    cmd = sys.argv[4] if len(sys.argv) > 4 else ("python3 bench.py --steps 2 --warmup 1 --dtype bf16 --no-cpu-baseline --graph 0 --train-steps 0 "
                                                 "--no-glow-variant")        # the command the two passes profiled (the caller says which)
    json.dump({"source_sha1": sha, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- " + cmd,
               "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), counters in KiB", "kernels": res},
              open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"])[:8]:
        print(f"{k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  x{v['launches_profiled']}")


if __name__ == "__main__":
    main()
