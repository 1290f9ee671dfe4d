#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export AMDKGE_LIB=$ROOT/build/ablate/libamdkge.so
cd /tmp; export TMPDIR=/tmp
for dbg in 0 256 512 768; do
 for a in "--model TransE" "--model RotatE"; do
  d=$ROOT/gpurun_out/abl2/${dbg}_$(echo $a | tr -d ' -'); rm -rf $d; mkdir -p $d
  AMDKGE_DEBUG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $ROOT/bench.py --no-cpu-baseline --no-eval --steps 60 --warmup 10 $a > $d.json 2> $d.err
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== dbg=$dbg $a"; python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    if 'kge' in r['Name']: print('   ', r['Name'][:56], round(float(r['AverageNs'])/1e3,1),'us')
PY
 done
done
