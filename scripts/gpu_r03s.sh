#!/bin/bash
set -u
O=$PWD/gpurun_out/r03s; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rank_screen.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|^E  |FAILED" $O/pytest.log | tail -12
PYTHONPATH=$R timeout 200 python scripts/filter_index_timing.py 2>&1 | tail -4
timeout 300 python bench.py --steps 56 --warmup 10 --no-cpu-baseline 2>>$O/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('C2', round(d['ms_per_step'],4), d['eval'])"
timeout 300 python bench.py --no-cpu-baseline 2>>$O/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('C2 default', round(d['ms_per_step'],4), d['eval'])"
