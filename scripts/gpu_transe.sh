#!/bin/bash
# development: TransE single-pass forward kernel -- parity tests, then bench lines + kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/transe; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lazy.py tests/test_gpu_learning.py tests/test_gpu_deterministic.py -m gpu -q -x -k "TransE or lazy or learn" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -10
for a in "--model TransE" "--model TransE --k 352" "--model TransE --k 100" "--config C1 --deterministic" "--model TransE --loss pairwise"; do
  timeout 200 python bench.py $a --no-cpu-baseline --no-eval --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$a', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), d['phases_ms'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $ROOT/bench.py --no-cpu-baseline --no-eval --steps 100 --warmup 10 --model TransE > $O/prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-160
