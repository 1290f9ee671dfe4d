#!/bin/bash
# development: per-kernel times of scripts/screen_time.py for the default library and the named build_variants/
R=$PWD; export TMPDIR=/tmp; cd /tmp
for v in "" $@; do
  if [ -n "$v" ]; then export AMDKGE_LIB=$R/build_variants/$v/libamdkge.so; fi
  O=$R/gpurun_out/var_${v:-default}; rm -rf $O; mkdir -p $O
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python $R/scripts/screen_time.py 2>$O/err.log | tail -1; tail -2 $O/err.log | cut -c1-200
  python - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(s in r["Name"] for s in ("rank_screen_kernel", "rank_recheck", "rank_filter")): print("  ${v:-default}", r["Name"].split("(")[0][:40], "avg us", round(float(r["AverageNs"])/1e3, 1))
PY
done
