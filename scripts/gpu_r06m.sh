#!/bin/bash
# which screening kernels the bench's evaluation runs, and for how long (kernel trace of a short bench run)
set -u
O=gpurun_out/${1:-r06m}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in 1 4; do
P=/tmp/trace_b$v; rm -rf $P; mkdir -p $P
AMDKGE_SCREEN_KERNEL=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python $R/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --also none > $P/out.log 2> $P/err.log || tail -3 $P/err.log
python - "$v" <<PY
import csv, glob, sys
for g in glob.glob("/tmp/trace_b"+sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(g)))
    print("screen kernel env", sys.argv[1])
    for r in rows[:16]: print("  %-80s calls %5s avg_us %9.1f  %5s%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("$R/$O/bench_kernel_stats_v"+sys.argv[1]+".csv", "w").write(open(g).read())
PY
done
cd $R
