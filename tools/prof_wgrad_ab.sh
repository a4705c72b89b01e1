set -e
O=gpurun_out/r5/wg
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
ITERS=4 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/multi -- python3 $R/tools/train_bench.py > $R/$O/multi.log 2>&1
ITERS=4 MHE_WGRAD_MULTI=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/single -- python3 $R/tools/train_bench.py > $R/$O/single.log 2>&1
cd $R
python tools/trace_stats.py $O/multi $O/multi_stats.csv step:adam_kernel:4 | tail -1
python tools/trace_stats.py $O/single $O/single_stats.csv step:adam_kernel:4 | tail -1
find $O -name "*kernel_trace.csv" -size +8M -delete
