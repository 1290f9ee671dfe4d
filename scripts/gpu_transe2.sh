#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/transe; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lazy.py tests/test_gpu_learning.py -m gpu -q -k "tiled or lazy or learn" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -10
bash scripts/gpu_variants.sh default,pf10,pf12,w6,w5pf8 "--model TransE" "--model DistMult" "--config C3" "--config C2"
