#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/transe; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lazy.py tests/test_gpu_learning.py tests/test_gpu_deterministic.py tests/test_gpu_session.py tests/test_gpu_shard_kernels.py -m gpu -q -k "tiled or lazy or learn or determin or session or shard" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -10
bash scripts/gpu_variants.sh default "--model TransE" "--model TransE --k 352" "--model TransE --k 100" "--config C1 --deterministic" "--config C1" "--model RotatE" "--config C2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $ROOT/bench.py --no-cpu-baseline --no-eval --steps 100 --warmup 10 --model TransE > $O/prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -4 $f | cut -d, -f1-4 | cut -c1-120
