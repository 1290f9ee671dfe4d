#!/bin/bash
# development: tile-size sweep of the direct tile pass (build/<name>/libamdkge.so)
set -u
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
for lib in b1200 b2400 b4800 b2400p2; do
  export AMDKGE_LIB=$PWD/build/$lib/libamdkge.so
  for cfg in "--config C5 --optimizer-mode lazy --steps 5 --warmup 2 --reps 2" "--config C5 --optimizer-mode dense --steps 5 --warmup 2 --reps 2" "--config C5 --dataset synth-c5-small --optimizer-mode dense --steps 10 --warmup 2 --reps 2" "--model ComplEx --k 1000 --steps 28 --warmup 5"; do
    timeout 600 python bench.py $cfg --no-cpu-baseline --no-eval 2>>$O/err.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib | $cfg |', round(d['ms_per_step'],3))" | tee -a $O/variants.txt
  done
done
