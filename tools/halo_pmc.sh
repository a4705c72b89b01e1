# counters of conv_halo_kernel on tools/halo_check.py's launches: bash tools/halo_pmc.sh   (-> gpurun_out/halo_pmc.txt)
O=gpurun_out/halo_pmc
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/a -- python3 $R/tools/halo_check.py 256 > $R/$O/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/$O/b -- python3 $R/tools/halo_check.py 256 > $R/$O/b.log 2>&1
cd $R
python - <<'PY' > gpurun_out/halo_pmc.txt
import csv, glob, collections
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/halo_pmc/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "halo" not in k and "conv_kernel" not in k and "p8" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"): n[k] += 1
    for k, v in acc.items():
        print(d, k, n[k], {c: f"{x / max(n[k],1):.4g}" for c, x in v.items()})
PY
cat gpurun_out/halo_pmc.txt
find $O -name "*.csv" -size +2M -delete
