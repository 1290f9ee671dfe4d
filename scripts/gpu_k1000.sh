#!/bin/bash
export TMPDIR=/tmp
for d in 1 0; do
AMDKGE_TILE_DIRECT=$d timeout 300 python bench.py --model ComplEx --k 1000 --steps 28 --warmup 5 --no-cpu-baseline --no-eval 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('direct=$d ComplEx k=1000', round(d['ms_per_step'],4), d['roofline']['frac'], d['phases_ms'])"
done
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/k1000 -o r -- python $OLDPWD/bench.py --model ComplEx --k 1000 --steps 28 --warmup 5 --reps 1 --no-cpu-baseline --no-eval > /dev/null 2>&1; cd $OLDPWD
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/k1000/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
