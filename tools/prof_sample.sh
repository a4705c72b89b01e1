#!/bin/bash
# kernel-level profile of the reference iteration's metrics pass (tools/sample_bench.py): bash tools/prof_sample.sh <tag>
set -e
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/r5/prof_sample_$tag
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $GRAFT_REPO_ROOT/tools/sample_bench.py > $out.log 2>&1
tail -1 $out.log
python3 - "$out/s_kernel_stats.csv" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mhe::" in r["Name"]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
