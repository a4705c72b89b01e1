# HBM traffic of the train step by the counters, with and without the Gram-statistics reverse of conv3 (separate FETCH_SIZE / WRITE_SIZE passes):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/refresh_train_traffic_r03.sh'
set -e
O=gpurun_out/r3t
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  MHE_CONV3_FOLD=$f MHE_TRAIN_RECOMPUTE=$f ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/fetch$f -- python3 $R/tools/train_bench.py > $R/$O/f$f.log 2>&1
  MHE_CONV3_FOLD=$f MHE_TRAIN_RECOMPUTE=$f ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/write$f -- python3 $R/tools/train_bench.py > $R/$O/w$f.log 2>&1
  MHE_CONV3_FOLD=$f MHE_TRAIN_RECOMPUTE=$f ITERS=5 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr$f -- python3 $R/tools/train_bench.py > $R/$O/t$f.log 2>&1
done
cd $R
for f in 1 0; do
  python tools/pmc_traffic.py $O/fetch$f $O/write$f $O/train_traffic_fold$f.json
  python tools/trace_stats.py $O/tr$f $O/train_stats_fold$f.csv step:adam_kernel:5 > /dev/null
done
find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O
