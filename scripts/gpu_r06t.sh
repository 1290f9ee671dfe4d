#!/bin/bash
# round 6: the paired-wave screening kernel (AMDKGE_SCREEN_KERNEL=5) against kernel r: ranks crc, identity tests, kernel trace, counters
set -u
O=gpurun_out/${1:-r06t}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in 4 5; do
  AMDKGE_SCREEN_KERNEL=$v timeout 200 python scripts/screen_time.py 2>&1 | tail -1 | tee -a $O/screen_time.txt
done
AMDKGE_SCREEN_KERNEL=5 timeout 900 python -m pytest tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "screen or bit_identical" > $O/pytest_screen_p.log 2>&1; echo "rc=$?" >> $O/pytest_screen_p.log; tail -5 $O/pytest_screen_p.log | cut -c1-300
cd /tmp
P=/tmp/trace_p; rm -rf $P; mkdir -p $P
AMDKGE_SCREEN_KERNEL=5 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > $P/out.log 2> $P/err.log || tail -3 $P/err.log
python - <<PY
import csv, glob
for g in glob.glob("/tmp/trace_p/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(g)))
    for r in rows[:8]: print("  %-70s calls %5s avg_us %9.1f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("$R/$O/kernel_stats_p.csv", "w").write(open(g).read())
PY
cd $R
sed -i 's/"rank_screen_kernel_r" in r\["Kernel_Name"\]/("rank_screen_kernel_r" in r["Kernel_Name"] or "rank_screen_kernel_p" in r["Kernel_Name"])/' scripts/gpu_screen_quick.sh
bash scripts/gpu_screen_quick.sh 5 ${1:-r06t} 2>&1 | tail -1
