#!/bin/bash
# round-4 closing measurements: full GPU suite, smoke, headline + rocprof stats + PMC passes, the five models at the C2 shape, k = 350,
# BASELINE configs C1 / C3 / C4 / one GPU's C5 shard (dense and lazy, with kernel stats and FETCH / WRITE passes), zipf,
# deterministic mode, 2-rank gloo self-launches of the multi-rank presets.   usage: scripts/gpu_r04_close.sh TAG
set -u
TAG=${1:-r04z}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -14 $O/pytest.log | grep -E "FAILED|passed|failed|rc="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_c2_driver_flags.json 2>> $O/bench_c2.err
bash scripts/profile_bench.sh $TAG > $O/profile.log 2>&1
for m in ComplEx DistMult HolE TransE RotatE; do timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --also none >> $O/models.jsonl 2>> $O/models.err; done
for m in ComplEx DistMult TransE RotatE; do timeout 300 python bench.py --model $m --k 350 --no-cpu-baseline --trained-eval --also none >> $O/k350.jsonl 2>> $O/models.err; done
for c in C1 C3 C4; do timeout 400 python bench.py --config $c --no-cpu-baseline --trained-eval >> $O/configs.jsonl 2>> $O/configs.err; done
timeout 400 python bench.py --config C4 --optimizer-mode lazy --no-cpu-baseline >> $O/configs.jsonl 2>> $O/configs.err
timeout 300 python bench.py --popularity zipf --no-cpu-baseline >> $O/zipf.jsonl 2>> $O/configs.err
timeout 300 python bench.py --deterministic --no-cpu-baseline --no-eval --also none >> $O/det.jsonl 2>> $O/configs.err
timeout 300 python bench.py --deterministic --model TransE --no-cpu-baseline --no-eval --also none >> $O/det.jsonl 2>> $O/configs.err
AMDKGE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-eval >> $O/rccl_world1.jsonl 2>> $O/configs.err
AMDKGE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --config C4 --steps 20 --warmup 5 --no-cpu-baseline --no-eval >> $O/rccl_world1.jsonl 2>> $O/configs.err
for extra in "" "--config C4" "--config C5 --ents-per-gpu 400000 --batch 8192"; do
  AMDKGE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 $extra --steps 20 --warmup 3 --no-cpu-baseline --no-eval >> $O/gloo2.jsonl 2>> $O/gloo2.err
done
bash scripts/gpu_c5_prof.sh $TAG > $O/c5_prof.log 2>&1; tail -12 $O/c5_prof.log
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        if isinstance(d,dict) and 'ms_per_step' in d:
            ev=d.get('eval') or {}
            et=d.get('eval_trained_like') or {}
            print(f.split('/')[-1], d['config']['workload'][:50], d['config'].get('optimizer_mode'), 'n', d['n_gpus'], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'eval', round(ev.get('ranks_per_s',0)), 'ms', round(ev.get('ms',0),3), 'exact', round(((ev.get('exact_fp32_kernel_alone') or {}).get('ms') or 0),3),
                  '| trained-like', round(et.get('ranks_per_s',0)), 'ms', round(et.get('ms',0),3), 'plain', round(((et.get('exact_fp32_kernel_alone') or {}).get('ms') or 0),3), 'mrr', round(et.get('mrr',0),3))
PY
grep -l "Error\|Traceback" $O/*.err | head
