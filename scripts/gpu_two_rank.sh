#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/two_rank; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in rows dp; do
  rm -rf $O/$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o r -- python $ROOT/scripts/two_rank_profile.py $mode > $O/$mode.log 2>&1
  tail -1 $O/$mode.log
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-110 | head -24
done
