# Round-5 evidence in one gpurun call: bench line, rocprofv3 kernel statistics of the forward step and of the train step, HBM traffic
# (separate FETCH_SIZE / WRITE_SIZE passes) and MFMA / wave-state counters for the forward AND the train step.
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/refresh_profiles_r05.sh'
# then copy gpurun_out/r5p/* summaries into profiles/ (the script prints the cp lines).
set -e
O=gpurun_out/r5p
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
FW="--steps 5 --warmup 2 --no-cpu-baseline --graph 0 --train-steps 0 --no-glow-variant"
PM="--steps 2 --warmup 1 --no-cpu-baseline --graph 0 --train-steps 0 --no-glow-variant"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o s -- python3 $R/bench.py $FW > $R/$O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py $PM > $R/$O/pf.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py $PM > $R/$O/pw.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_mfma -- python3 $R/bench.py $PM > $R/$O/pm.log 2>&1
# train step: kernel statistics over 5 eager steps (warm-up dispatches dropped by --skip-frac) and the MFMA counters over 1 step
ITERS=5 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/train_tr -- python3 $R/tools/train_bench.py > $R/$O/train.log 2>&1
ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_mfma_train -- python3 $R/tools/train_bench.py > $R/$O/pmt.log 2>&1
# HBM traffic of the TRAIN step per kernel (bench.py's train_step.roofline.traffic): the same two passes over one eager step
ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch_train -- python3 $R/tools/train_bench.py > $R/$O/pft.log 2>&1
ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write_train -- python3 $R/tools/train_bench.py > $R/$O/pwt.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
python tools/pmc_traffic.py $O/pmc_fetch_train $O/pmc_write_train $O/train_pmc_traffic.json "ITERS=1 python3 tools/train_bench.py (one eager TrainStep.step at config C2 after two warm-up steps)"
cp $O/train_pmc_traffic.json profiles/r05_train_pmc_traffic.json
# the headline line last: it reads profiles/r05_pmc_traffic.json for roofline.traffic, which must come from these very sources
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python tools/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma.json
python tools/pmc_mfma.py $O/pmc_mfma_train $O/pmc_mfma_train.json
python tools/trace_stats.py $O/train_tr $O/train_kernel_stats.csv step:adam_kernel:5 > $O/train_stats.txt; tail -1 $O/train_stats.txt
# the forward's kernels PER TIMED STEP (a step ends with its single elbo_reduce launch): nothing of the model's construction in it
python tools/trace_stats.py $O/stats $O/step_kernel_stats.csv step:elbo_reduce_kernel:5 > $O/step_stats.txt; tail -1 $O/step_stats.txt
timeout -k 10 300 python bench.py --workload c1 --no-glow-variant > $O/bench_c1.json 2> $O/bench_c1.err || true
find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O
echo "cp $O/bench.json profiles/r05_bench_c2_bf16.json; cp $O/stats/s_kernel_stats.csv profiles/r05_c2_bf16_kernel_stats.csv; cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json; cp $O/pmc_mfma.json profiles/r05_pmc_mfma.json; cp $O/pmc_mfma_train.json profiles/r05_pmc_mfma_train.json; cp $O/train_kernel_stats.csv profiles/r05_train_kernel_stats.csv; cp $O/step_kernel_stats.csv profiles/r05_c2_bf16_step_kernel_stats.csv; cp $O/bench_c1.json profiles/r05_bench_c1_f32.json; cp $O/train_pmc_traffic.json profiles/r05_train_pmc_traffic.json"
# the Glow branch's head + flow train step (from the trunk feature on) per kernel
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/glow -o glow -- python3 $R/tools/glow_train_prof.py > $R/$O/glow.log 2>&1 || true
cd $R
cp $O/glow/glow_kernel_stats.csv profiles/r05_glow_head_train_kernel_stats.csv 2>/dev/null || true
# the metrics pass of the reference's iteration (sample(N=[200,200]) + MHEntLoss from the conditioning feature on) per kernel
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/sample -o s -- python3 $R/tools/sample_bench.py > $R/$O/sample.log 2>&1 || true
cd $R
cp $O/sample/s_kernel_stats.csv profiles/r05_metrics_pass_kernel_stats.csv 2>/dev/null || true
tail -1 $O/sample.log || true
