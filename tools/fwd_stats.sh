# per-step kernel times of the forward bench (graph off so that kernels are traced one by one): bash tools/fwd_stats.sh  -> gpurun_out/fwd_stats.txt
O=gpurun_out/fwd_stats
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --graph 0 --train-steps 0 --no-glow-variant > $R/$O/log.txt 2>&1
cd $R
python tools/trace_stats.py $O/t $O/step.csv step:randn_kernel:5 > /dev/null
python - <<'PY' > gpurun_out/fwd_stats.txt
import csv
rows = list(csv.DictReader(open("gpurun_out/fwd_stats/step.csv")))
tot = 0
for r in rows:
    c = int(r["Calls"]) / 5; a = float(r["AverageNs"]) / 1e3; tot += c * a
    print("%8.1f us  x%5.1f %8.1f us  %s" % (c * a, c, a, r["Name"][:100]))
print("sum", tot)
PY
head -40 gpurun_out/fwd_stats.txt
find $O -name "*.csv" -size +3M -delete
