#!/bin/bash
set -u
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_learning.py tests/test_gpu_discovery.py tests/test_gpu_kernels.py tests/test_gpu_lazy.py tests/test_gpu_session.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
./scripts/gpu_c5_prof.sh r03a
