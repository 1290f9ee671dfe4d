#!/bin/bash
# round 3, first full check: GPU suite, smoke, headline line with the new timed region, self-launched 2-rank gloo runs
set -u
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_20.json 2>> $O/bench_c2.err
# bench.py --gpus 2 WITHOUT a launcher: must start two ranks itself (gloo: both on the one GPU of this box)
for cfg in "" "--config C4" "--config C5 --ents-per-gpu 200000 --batch 8192"; do
  AMDKGE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --reps 3 --no-cpu-baseline --no-eval $cfg >> $O/gloo2_selflaunch.jsonl 2>> $O/gloo2.err
  echo "rc=$?" >> $O/gloo2.err
done
# nccl backend with --gpus 2 on a one-GPU box must fail loudly
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/nccl2.out 2>&1; echo "rc=$?" >> $O/nccl2.out; tail -2 $O/nccl2.out
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        if isinstance(d,dict) and 'ms_per_step' in d: print(f.split('/')[-1], d['config']['workload'][:60], 'n_gpus', d['n_gpus'], 'ms', round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['ms_per_step_max'],4), 'frac', round(d['roofline']['frac'],3), 'eval', round(d.get('eval',{}).get('ranks_per_s',0)), d['phases_ms'])
PY
tail -5 $O/gloo2.err
