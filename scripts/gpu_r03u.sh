#!/bin/bash
set -u
O=$PWD/gpurun_out/r03u; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rank_screen.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|^E  |FAILED" $O/pytest.log | tail -12
bash scripts/gpu_variants.sh unroll16 2>&1 | grep -E "rank_screen_kernel|two sides"
