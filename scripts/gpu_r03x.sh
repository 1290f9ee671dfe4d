#!/bin/bash
set -u
O=$PWD/gpurun_out/r03x; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|^E  |FAILED" $O/pytest.log | tail -12
python scripts/screen_diag.py 2>&1 | grep -v amdgpu.ids
bash scripts/gpu_variants.sh 2>&1 | grep -E "rank_screen_kernel|two sides|rank_recheck"
timeout 300 python bench.py --steps 56 --warmup 10 --no-cpu-baseline 2>>$O/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); e=d['eval']; print('C2-56', round(d['ms_per_step'],4), e['ranks_per_s'], e['ms'], e['screening']['rechecked_pairs_per_side'], e['exact_fp32_kernel_alone'])"
timeout 300 python bench.py --no-cpu-baseline 2>>$O/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); e=d['eval']; print('C2 default', round(d['ms_per_step'],4), e['ranks_per_s'], e['ms'], e['screening']['rechecked_pairs_per_side'], e['exact_fp32_kernel_alone'])"
python scripts/eval_profile.py 2>&1 | grep "evaluate() call"
