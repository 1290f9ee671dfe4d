#!/bin/bash
# round-2 closing measurements (after the single-pass RotatE kernels): full GPU suite, smoke, headline + rocprof stats + PMC passes, the five models at the C2 shape,
# BASELINE configs C1 / C3 / C4 / C5 (dense and lazy) on one GPU, k = 350, zipf, 2-rank gloo runs of the multi-rank paths
set -u
O=gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -14 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
bash scripts/profile_bench.sh r02m > $O/profile.log 2>&1
for m in ComplEx DistMult HolE TransE RotatE; do timeout 300 python bench.py --model $m --no-cpu-baseline >> $O/models.jsonl 2>> $O/models.err; done
for m in ComplEx DistMult TransE RotatE; do timeout 300 python bench.py --model $m --k 350 --no-cpu-baseline >> $O/k350.jsonl 2>> $O/models.err; done
for c in C1 C3 C4; do timeout 400 python bench.py --config $c --no-cpu-baseline >> $O/configs.jsonl 2>> $O/configs.err; done
timeout 400 python bench.py --config C4 --optimizer-mode lazy --no-cpu-baseline >> $O/configs.jsonl 2>> $O/configs.err
for mode in dense lazy; do timeout 900 python bench.py --config C5 --optimizer-mode $mode --steps 20 --warmup 3 >> $O/configs.jsonl 2>> $O/configs.err; done
timeout 300 python bench.py --popularity zipf --no-cpu-baseline >> $O/zipf.jsonl 2>> $O/configs.err
timeout 300 python bench.py --deterministic --no-cpu-baseline --no-eval >> $O/det.jsonl 2>> $O/configs.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02m/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        print(f.split('/')[-1], d['metric'][-42:], d['config'].get('optimizer_mode'), 'n', d['n_gpus'], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), {k: round(v,3) for k,v in d.get('phases_ms',{}).items()}, 'eval', round(d.get('eval',{}).get('ranks_per_s',0)))
PY
grep -l "Error\|Traceback" $O/*.err | head
