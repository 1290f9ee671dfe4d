#!/bin/bash
# round 6: kernel r with the integer per-(row, tile) thresholds: identity tests first, then timing / counters / kernel trace
set -u
O=gpurun_out/${1:-r06f}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in 1 4; do
  AMDKGE_SCREEN_KERNEL=$v timeout 200 python scripts/screen_time.py 2>&1 | tail -1 | tee -a $O/screen_time.txt
done
AMDKGE_SCREEN_KERNEL=4 timeout 900 python -m pytest tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "screen or bit_identical" > $O/pytest_screen_r.log 2>&1; echo "rc=$?" >> $O/pytest_screen_r.log; tail -5 $O/pytest_screen_r.log | cut -c1-300
cd /tmp
P=/tmp/trace_r; rm -rf $P; mkdir -p $P
AMDKGE_SCREEN_KERNEL=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > $P/out.log 2> $P/err.log || tail -3 $P/err.log
python - <<PY
import csv, glob
for g in glob.glob("/tmp/trace_r/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(g)))
    for r in rows[:9]: print("  %-70s calls %5s avg_us %9.1f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("$R/$O/kernel_stats_r.csv", "w").write(open(g).read())
PY
cd $R
bash scripts/gpu_screen_quick.sh 4 ${1:-r06f} 2>&1 | tail -1
