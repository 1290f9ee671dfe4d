#!/bin/bash
# development: compare library variants (build/<name>/libamdkge.so) on a few bench shapes
set -u
for lib in default "$@"; do
  if [ "$lib" != default ]; then export AMDKGE_LIB=$PWD/build/$lib/libamdkge.so; else unset AMDKGE_LIB; fi
  for a in "--model TransE" "--model DistMult" "--model TransE --k 100" "--config C1" "--config C2"; do
    timeout 200 python bench.py $a --no-cpu-baseline --no-eval --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib | $a |', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), round(d['phases_ms']['kernels'],4))"
  done
done
