#!/bin/bash
# development: the screening kernel's own duration (rocprofv3 kernel trace) for the ablation builds of rank_screen_kernel_r
set -u
O=gpurun_out/${1:-r06e_ablate}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for n in base ${ABL:-scrr_ab1 scrr_ab2}; do
  if [ $n = base ]; then unset AMDKGE_LIB; else export AMDKGE_LIB=$R/build_variants/$n/libamdkge.so; fi
  P=/tmp/trace_$n; rm -rf $P; mkdir -p $P
  AMDKGE_SCREEN_KERNEL=${SK:-4} timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > $P/out.log 2> $P/err.log || tail -3 $P/err.log
  python - "$n" <<PY
import csv, glob, sys
for g in glob.glob("/tmp/trace_"+sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel" in r["Name"] or "recheck_kernel<false>" in r["Name"] or "rank_count_mfma" in r["Name"]:
            line = "ablate %-5s %-60s calls %4s avg_us %9.1f" % (sys.argv[1], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3)
            print(line); open("$R/$O/ablate.txt", "a").write(line + "\n")
PY
done
cd $R
