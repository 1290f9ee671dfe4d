#!/bin/bash
# development: compare library variants (build/<name>/libamdkge.so) on bench shapes: $1 = "lib1,lib2", rest = bench flag strings
set -u
IFS=',' read -ra LIBS <<< "$1"; shift
for lib in "${LIBS[@]}"; do
  if [ "$lib" != default ]; then export AMDKGE_LIB=$PWD/build/$lib/libamdkge.so; else unset AMDKGE_LIB; fi
  for a in "$@"; do
    timeout 300 python bench.py $a --no-cpu-baseline --no-eval --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib | $a |', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), round(d['phases_ms']['kernels'],4))"
  done
done
