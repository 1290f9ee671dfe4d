#!/bin/bash
# development: for the shipped library ("base") and each build_variants/<name>[:K]: screen_time.py's line (serial evaluate-shaped timing,
# recheck stats, ranks crc32 -- the same ranks whatever the kernel) and the screening kernel's own duration under rocprofv3 --kernel-trace.
# <name>:K runs that library with AMDKGE_SCREEN_KERNEL=K.   usage: VARS="scrr_o1 scrs:5" gpu_screen_variants.sh <tag>
set -u
O=gpurun_out/${1:-r06x}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in base ${VARS:-}; do
  n=${v%%:*}; k=${v#*:}; [ "$k" = "$v" ] && k=""
  if [ $n = base ]; then unset AMDKGE_LIB; else export AMDKGE_LIB=$R/build_variants/$n/libamdkge.so; fi
  if [ -n "$k" ]; then export AMDKGE_SCREEN_KERNEL=$k; else unset AMDKGE_SCREEN_KERNEL; fi
  echo "$v: $(timeout 200 python $R/scripts/screen_time.py 2>&1 | tail -1)" | tee -a $R/$O/variants.txt
  P=/tmp/trace_$n; rm -rf $P; mkdir -p $P
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > $P/out.log 2> $P/err.log || tail -3 $P/err.log
  python - "$v" "$n" <<PY
import csv, glob, sys
for g in glob.glob("/tmp/trace_"+sys.argv[2]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel" in r["Name"] or "recheck_kernel<false>" in r["Name"] or "rank_count_mfma" in r["Name"]:
            line = "  %-10s %-60s calls %4s avg_us %9.1f" % (sys.argv[1], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3)
            print(line); open("$R/$O/variants.txt", "a").write(line + "\n")
PY
done
cd $R
