#!/bin/bash
set -u
O=gpurun_out/r02f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -20
timeout 300 python bench.py --no-cpu-baseline --no-eval --deterministic > $O/bench_det.json 2> $O/bench_det.err
timeout 300 python bench.py --no-cpu-baseline --no-eval --deterministic --model TransE > $O/bench_det_transe.json 2>> $O/bench_det.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02f/bench*.json')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        print(f.split('/')[-1], d['metric'][-45:], 'det', d['config'].get('deterministic'), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3))
PY
tail -3 $O/bench_det.err
