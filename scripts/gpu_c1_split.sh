#!/bin/bash
# kernel split of BASELINE configs[0] (TransE k=50 eta=5 pairwise) on both train paths.   usage: scripts/gpu_c1_split.sh TAG
TAG=${1:-c1}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for path in atomic tiled; do
  AMDKGE_TRAIN_PATH=$path timeout 200 python bench.py --config C1 --no-cpu-baseline --no-eval --also none >> $O/c1_paths.jsonl 2>> $O/err.log
  (cd /tmp && AMDKGE_TRAIN_PATH=$path timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$path -o r -- python $GRAFT_REPO_ROOT/bench.py --config C1 --no-cpu-baseline --no-eval --also none --steps 50 --warmup 5 > /dev/null 2>> $GRAFT_REPO_ROOT/$O/err.log)
  f=$(find $O/prof_$path -name "*kernel_stats.csv" | head -1)
  echo "== $path"; [ -n "$f" ] && { head -8 "$f" | cut -c1-160; cp "$f" $O/c1_${path}_kernel_stats.csv; }
  rm -rf $O/prof_$path
done
python - <<PY
import json
for l in open('$O/c1_paths.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:60], round(d['ms_per_step'],4), round(d['roofline']['frac'],3), d.get('phases_ms'))
PY
