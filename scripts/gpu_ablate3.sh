#!/bin/bash
# development: ablation build under AMDKGE_DEBUG masks on arbitrary bench flags: $1 = bench flags, rest = masks
set -u
export AMDKGE_LIB=$PWD/build/ablate/libamdkge.so
flags=$1; shift
for dbg in 0 "$@"; do
  AMDKGE_DEBUG=$dbg timeout 200 python bench.py $flags --no-cpu-baseline --no-eval --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$flags dbg=$dbg', round(d['ms_per_step'],4), d['phases_ms'])"
done
