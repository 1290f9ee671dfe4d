#!/bin/bash
# development: kernel + model tests, then the headline / C1 / TransE bench lines
set -u
O=gpurun_out/quick; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_lazy.py tests/test_gpu_deterministic.py tests/test_golden.py tests/test_gpu_session.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -10
for a in "--config C2" "--config C1" "--model TransE" "--model RotatE" "--config C3" "$@"; do
  timeout 200 python bench.py $a --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$a', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), d['phases_ms'], 'eval', round(d.get('eval',{}).get('ranks_per_s',0)))"
done
