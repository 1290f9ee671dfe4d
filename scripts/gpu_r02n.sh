#!/bin/bash
# last check of the round's final commit: full GPU suite, smoke, headline line and the two models whose kernels changed after r02m
set -u
O=gpurun_out/r02n; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for m in TransE DistMult RotatE; do timeout 300 python bench.py --model $m --no-cpu-baseline >> $O/models.jsonl 2>> $O/models.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02n/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        if isinstance(d,dict) and 'ms_per_step' in d: print(f.split('/')[-1], d['config']['workload'][28:62], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'eval', round(d.get('eval',{}).get('ranks_per_s',0)))
PY
