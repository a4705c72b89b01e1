set -e
mkdir -p gpurun_out/r2k
R=$PWD
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2k/test.log 2>&1 || true
tail -3 gpurun_out/r2k/test.log
timeout -k 10 400 python bench.py > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 2 --warmup 1 --dtype bf16 --no-cpu-baseline --graph 0 --train-steps 0 --no-glow-variant"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2k/stats -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --graph 0 --train-steps 0 --no-glow-variant > $R/gpurun_out/r2k/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2k/pmc_fetch -- python3 $R/bench.py $BARGS > $R/gpurun_out/r2k/pf.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2k/pmc_write -- python3 $R/bench.py $BARGS > $R/gpurun_out/r2k/pw.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r2k/pmc_mfma -- python3 $R/bench.py $BARGS > $R/gpurun_out/r2k/pm.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/r2k/pmc_fetch gpurun_out/r2k/pmc_write gpurun_out/r2k/pmc_traffic.json
python tools/pmc_mfma.py gpurun_out/r2k/pmc_mfma gpurun_out/r2k/pmc_mfma.json
ls gpurun_out/r2k/stats | head; du -sh gpurun_out/r2k
find gpurun_out/r2k -name "*counter_collection.csv" -size +5M -delete; find gpurun_out/r2k -name "*kernel_trace.csv" -path "*pmc*" -delete
