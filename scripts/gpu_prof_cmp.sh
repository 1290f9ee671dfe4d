#!/bin/bash
# development: per-kernel durations (rocprofv3 stats) of bench shapes under library variants: $1 = variants "a,b", rest = one bench flag string per shape
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/cmp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IFS=',' read -ra LIBS <<< "$1"; shift
for lib in "${LIBS[@]}"; do
  if [ "$lib" != default ]; then export AMDKGE_LIB=$ROOT/build/$lib/libamdkge.so; else unset AMDKGE_LIB; fi
  i=0
  for a in "$@"; do
    i=$((i+1)); d=$O/${lib}_$i; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $ROOT/bench.py --no-cpu-baseline --no-eval --steps 100 --warmup 10 $a > $d.json 2> $d.err
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    echo "== $lib | $a"; python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    if 'kge' in r['Name']: print('   ', r['Name'][:64], round(float(r['AverageNs'])/1e3,1),'us x',r['Calls'])
PY
  done
done
