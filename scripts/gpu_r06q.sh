#!/bin/bash
# round 6: (1) the multi-GPU first-run script, dry run: 8 ranks over gloo on the ONE GPU of this box; (2) SQ counters of the shipped screening kernel
set -u
O=gpurun_out/${1:-r06q}; mkdir -p $O
export TMPDIR=/tmp
( time AMDKGE_BENCH_BACKEND=gloo timeout 1500 python scripts/multi_gpu_first_run.py --gpus 8 --same-device --timeout 280 --out $O/multi_gpu_dry_run_gloo8.jsonl ) > $O/dry_run.log 2>&1; tail -12 $O/dry_run.log | cut -c1-400
bash scripts/gpu_screen_quick.sh 4 ${1:-r06q} 2>&1 | tail -2
bash scripts/gpu_screen_quick.sh 1 ${1:-r06q} 2>&1 | tail -1
