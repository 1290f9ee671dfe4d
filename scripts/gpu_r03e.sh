#!/bin/bash
set -u
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tile_direct.py tests/test_gpu_kernels.py tests/test_gpu_lazy.py tests/test_gpu_deterministic.py tests/test_gpu_fullsize.py tests/test_gpu_learning.py tests/test_gpu_discovery.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
for mode in lazy dense; do
  timeout 600 python bench.py --config C5 --no-cpu-baseline --no-eval --optimizer-mode $mode --steps 6 --warmup 2 --reps 3 >> $O/c5.jsonl 2>> $O/c5.err
done
timeout 300 python bench.py --config C5 --dataset synth-c5-small --no-cpu-baseline --no-eval --optimizer-mode dense --steps 10 --warmup 2 --reps 3 >> $O/c5small.jsonl 2>> $O/c5.err
timeout 300 python bench.py --model ComplEx --k 1000 --no-cpu-baseline --no-eval --steps 28 --warmup 5 >> $O/wide.jsonl 2>> $O/c5.err
timeout 300 python bench.py --steps 56 --warmup 10 --no-cpu-baseline > $O/c2.json 2>> $O/c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03e/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        if isinstance(d,dict) and 'ms_per_step' in d: print(f.split('/')[-1], d['config']['workload'][:70], d['config']['optimizer_mode'], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), d.get('eval',{}).get('ranks_per_s'), d.get('eval',{}).get('filter_index_ms'))
PY
tail -3 $O/c5.err
