#!/bin/bash
# development: ablation build (make EXTRA=-DKGE_ABLATE OUT=build/ablate/libamdkge.so) under AMDKGE_DEBUG masks
set -u
O=gpurun_out/ablate; mkdir -p $O
export AMDKGE_LIB=$PWD/build/ablate/libamdkge.so
for cfg in "C2" "C1"; do
 for dbg in 0 128 $@; do
  AMDKGE_DEBUG=$dbg timeout 200 python bench.py --config $cfg --no-cpu-baseline --no-eval --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg dbg=$dbg', round(d['ms_per_step'],4), d['phases_ms'])"
 done
done
