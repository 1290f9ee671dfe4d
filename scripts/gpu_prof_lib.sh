#!/bin/bash
# development: per-kernel durations of a library variant (build/<name>/libamdkge.so, or "default"): $1 = names "a,b", rest = bench flag strings
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/prof_lib; mkdir -p $O
IFS=',' read -ra LIBS <<< "$1"; shift
cd /tmp && export TMPDIR=/tmp
for lib in "${LIBS[@]}"; do
  if [ "$lib" != default ]; then export AMDKGE_LIB=$ROOT/build/$lib/libamdkge.so; else unset AMDKGE_LIB; fi
  for flags in "$@"; do
    tag=${lib}_$(echo $flags | tr -c "A-Za-z0-9" "_")
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o r -- python $ROOT/bench.py $flags --no-cpu-baseline --no-eval --steps 100 --warmup 10 > $O/$tag.json 2> $O/$tag.err
    f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); echo "== $lib | $flags"; python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]: print("  ", r["Name"][:56], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
  done
done
