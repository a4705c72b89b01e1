# A/B of one environment switch on the forward bench (and optionally the train step): bash tools/ab_env.sh VAR "0 1" [train]
# two rounds per value, same box.  Output: gpurun_out/ab_<VAR>.log
V=$1; VALS=$2; TR=${3:-0}
mkdir -p gpurun_out
L=gpurun_out/ab_$V.log
: > $L
for r in 1 2; do for v in $VALS; do
  env $V=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-glow-variant --train-steps $TR > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
  python - $V $v >> $L <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
t=d.get('train_step') or {}
print(sys.argv[1], sys.argv[2], 'fwd', d['ms_per_step'], 'train', t.get('ms_per_step'))
PY
done; done
cat $L
