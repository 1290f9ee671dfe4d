#!/bin/bash
set -u
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tile_direct.py tests/test_gpu_kernels.py tests/test_gpu_learning.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log | grep -E "FAILED|passed|failed"
for d in 1 0; do
for mode in lazy dense; do
  AMDKGE_TILE_DIRECT=$d timeout 600 python bench.py --config C5 --no-cpu-baseline --no-eval --optimizer-mode $mode --steps 6 --warmup 2 --reps 3 >> $O/c5_direct$d.jsonl 2>> $O/c5.err
done
AMDKGE_TILE_DIRECT=$d timeout 300 python bench.py --config C5 --dataset synth-c5-small --no-cpu-baseline --no-eval --optimizer-mode dense --steps 10 --warmup 2 --reps 3 >> $O/c5small_direct$d.jsonl 2>> $O/c5.err
AMDKGE_TILE_DIRECT=$d timeout 300 python bench.py --config C5 --dataset synth-c5-small --no-cpu-baseline --no-eval --optimizer-mode lazy --steps 10 --warmup 2 --reps 3 >> $O/c5small_direct$d.jsonl 2>> $O/c5.err
AMDKGE_TILE_DIRECT=$d timeout 300 python bench.py --model ComplEx --k 1000 --no-cpu-baseline --no-eval --steps 28 --warmup 5 >> $O/wide_direct$d.jsonl 2>> $O/c5.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03f/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        if isinstance(d,dict) and 'ms_per_step' in d: print(f.split('/')[-1], d['config']['workload'][:62], d['config']['optimizer_mode'], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3))
PY
tail -3 $O/c5.err
