#!/bin/bash
# development: per-kernel durations of the ablation build under AMDKGE_DEBUG masks: $1 = bench flags, rest = masks
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export AMDKGE_LIB=$ROOT/build/ablate/libamdkge.so
O=$ROOT/gpurun_out/ablate_prof; mkdir -p $O
flags=$1; shift
tag=$(echo $flags | tr -c "A-Za-z0-9" "_")
cd /tmp && export TMPDIR=/tmp
for dbg in "$@"; do
  AMDKGE_DEBUG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}d$dbg -o r -- python $ROOT/bench.py $flags --no-cpu-baseline --no-eval --steps 100 --warmup 10 > $O/${tag}d$dbg.json 2> $O/${tag}d$dbg.err
  f=$(find $O/${tag}d$dbg -name "*kernel_stats.csv" | head -1); echo "== $flags dbg=$dbg"; python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]: print("  ", r["Name"][:56], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
