#!/bin/bash
# kernel-trace stats of the non-headline shapes (TransE / RotatE at the C2 shape, C1, zipf)
set -u
TAG=${1:-r02g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o r -- python $ROOT/bench.py --no-cpu-baseline --no-eval --steps 100 --warmup 10 "$@" > $O/$name.json 2> $O/$name.err; 
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1); echo "== $name"; head -6 $f | cut -c1-150; }
run transe --model TransE
run rotate --model RotatE
run c1 --config C1
run c1tiled --config C1 --deterministic
run zipf --popularity zipf
