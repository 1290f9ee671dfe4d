#!/bin/bash
# round-2 GPU check A: full GPU test-suite, then the headline bench and the k % 4 != 0 shapes next to their padded twins
set -u
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for cfg in "ComplEx 350" "ComplEx 352" "TransE 350" "TransE 352" "RotatE 350" "RotatE 352" "DistMult 350" "DistMult 352"; do
  set -- $cfg
  timeout 300 python bench.py --model $1 --k $2 --no-cpu-baseline --steps 140 >> $O/bench_k350.jsonl 2>> $O/bench_k350.err
done
timeout 300 python bench.py --model TransE --k 50 --eta 5 --loss pairwise --no-cpu-baseline --steps 140 >> $O/bench_c1.jsonl 2>> $O/bench_c1.err
AMDKGE_TRAIN_PATH=tiled timeout 300 python bench.py --model TransE --k 50 --eta 5 --loss pairwise --no-cpu-baseline --steps 140 >> $O/bench_c1.jsonl 2>> $O/bench_c1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02a/bench*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        print(f.split('/')[-1], d['metric'][-40:], 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'eval', d.get('eval',{}).get('ranks_per_s'))
PY
