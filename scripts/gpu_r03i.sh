#!/bin/bash
set -u
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rank_screen.py -m gpu -q -x -s > $O/pytest_screen.log 2>&1; echo "rc=$?" >> $O/pytest_screen.log; grep -E "rechecked|passed|failed|Error|assert" $O/pytest_screen.log | tail -30
timeout 300 python bench.py --steps 28 --warmup 5 --no-cpu-baseline > $O/c2.json 2>> $O/err.log
python -c "
import json; d=json.load(open('$O/c2.json')); print('C2 eval', d['eval'])"
for lib in b1200 b2400 b4800; do
  export AMDKGE_LIB=$PWD/build/$lib/libamdkge.so
  for cfg in "--config C5 --optimizer-mode lazy --steps 5 --warmup 2 --reps 2" "--config C5 --optimizer-mode dense --steps 5 --warmup 2 --reps 2" "--config C5 --dataset synth-c5-small --optimizer-mode dense --steps 10 --warmup 2 --reps 2" "--model ComplEx --k 1000 --steps 28 --warmup 5"; do
    timeout 600 python bench.py $cfg --no-cpu-baseline --no-eval 2>>$O/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib | $cfg |', round(d['ms_per_step'],3))" | tee -a $O/variants.txt
  done
done
unset AMDKGE_LIB
tail -3 $O/err.log
