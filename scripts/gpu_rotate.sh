#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/rotate; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lazy.py tests/test_gpu_learning.py tests/test_gpu_deterministic.py tests/test_gpu_session.py tests/test_gpu_shard_kernels.py tests/test_gpu_model.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -10
bash scripts/gpu_prof_lib.sh default "--model RotatE" "--config C5 --ents-per-gpu 200000 --steps 8 --warmup 2"
bash scripts/gpu_variants.sh default "--model RotatE --k 352" "--model TransE" "--config C2"
