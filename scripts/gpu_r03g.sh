#!/bin/bash
# development: direct tile pass variants at one GPU's C5 shard (build/<name>/libamdkge.so)
set -u
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
for lib in default pre0 rows4 pre0rows4 w4; do
  if [ "$lib" != default ]; then export AMDKGE_LIB=$PWD/build/$lib/libamdkge.so; else unset AMDKGE_LIB; fi
  for mode in lazy dense; do
    timeout 600 python bench.py --config C5 --no-cpu-baseline --no-eval --optimizer-mode $mode --steps 5 --warmup 2 --reps 2 2>>$O/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib $mode', round(d['ms_per_step'],3))" | tee -a $O/variants.txt
  done
done
