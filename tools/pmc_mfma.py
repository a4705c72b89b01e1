#!/usr/bin/env python3
"""Summarise one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) of bench.py into per-kernel MFMA utilisation and wave-state shares.

Units per /opt/skills/guides/MI355X_MICROARCH.md (cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles of
MFMA-pipe occupancy summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles summed over waves.  mfma_util = MFMA busy cycles / (elapsed shader cycles x 1024 SIMDs),
elapsed shader cycles = GRBM_GUI_ACTIVE / 8.

    python tools/pmc_mfma.py gpurun_out/pmc_mfma profiles/r01_pmc_mfma.json
"""
import collections, csv, glob, json, re, sys

SIMDS = 256 * 4


def main():
    src, dst = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(src + "/*/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key); n[k] += 1
    out = {}
    for k, c in acc.items():
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0 or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        out[k] = {"launches": n[k], "elapsed_shader_cycles_per_launch": round(cyc / n[k]),
                  "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS), 4),
                  "wave_share_waiting(s_waitcnt/barrier)": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                  "wave_share_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                  "wave_share_issue_stalled_on_lds": round(c.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
                  "wave_share_issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python bench.py --steps 2 --warmup 1 "
                         "--no-cpu-baseline --graph 0 --train-steps 1",
               "kernels": out}, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["elapsed_shader_cycles_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k[:72]:72s} x{v['launches']:4d} mfma_util {v['mfma_util']:.3f} wait {v['wave_share_waiting(s_waitcnt/barrier)']:.2f} "
              f"stall {v['wave_share_issue_stalled']:.2f} (lds {v['wave_share_issue_stalled_on_lds']:.2f}) issue {v['wave_share_issuing']:.2f}")


if __name__ == "__main__":
    main()
