#!/bin/bash
# evaluate(): the screening-pass / full-size / model tests, then eval lines of five configurations
set -u
O=$PWD/gpurun_out/eval_check; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|^E  |FAILED" $O/pytest.log | tail -12
for c in "" "--steps 56 --warmup 10" "--config C3" "--config C4" "--model DistMult" "--k 350"; do
timeout 300 python bench.py $c --no-cpu-baseline 2>>$O/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); e=d['eval']; print('$c', round(d['ms_per_step'],4), round(e['ranks_per_s']/1e6,2), round(e['ms'],3), e['screening']['rechecked_pairs_per_side'], round(e['exact_fp32_kernel_alone']['ms'],3), e['exact_fp32_kernel_alone']['ranks_identical_to_screened'])"
done
python scripts/eval_profile.py 2>&1 | grep "evaluate() call"
