#!/bin/bash
# round 3 checkpoint: full GPU suite, smoke, headline bench, C5 both modes, models
set -u
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log | grep -E "FAILED|passed|failed|rc="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench.err
for mode in lazy dense; do timeout 600 python bench.py --config C5 --no-cpu-baseline --no-eval --optimizer-mode $mode --steps 6 --warmup 2 --reps 3 >> $O/c5.jsonl 2>> $O/bench.err; done
for m in TransE DistMult RotatE HolE; do timeout 300 python bench.py --model $m --no-cpu-baseline >> $O/models.jsonl 2>> $O/bench.err; done
for c in C1 C3 C4; do timeout 300 python bench.py --config $c --no-cpu-baseline >> $O/configs.jsonl 2>> $O/bench.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03m/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        if isinstance(d,dict) and 'ms_per_step' in d:
            ev=d.get('eval') or {}
            print(f.split('/')[-1], d['config']['workload'][:58], d['config']['optimizer_mode'], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'eval', round(ev.get('ranks_per_s',0)), 'ms', round(ev.get('ms',0),3), 'idx', round(ev.get('filter_index_ms',0),1), (ev.get('screening') or {}).get('fraction'))
PY
tail -3 $O/bench.err
