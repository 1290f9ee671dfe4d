#!/bin/bash
# round 6, third lease: (1) the self-adversarial -inf coefficient fix (tests/test_gpu_nonfinite.py), (2) the register-resident-query
# screening kernel (AMDKGE_SCREEN_KERNEL=4, kge_rank_screen_r.h) against the shipped one: identity tests, serial timing, per-kernel
# trace, SQ counters.
set -u
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_nonfinite.py -q -p no:cacheprovider > $O/pytest_nonfinite.log 2>&1; echo "rc=$?" >> $O/pytest_nonfinite.log; grep -E "^FAILED| passed| failed" $O/pytest_nonfinite.log | cut -c1-200 | tail -20
for v in 1 4; do
  AMDKGE_SCREEN_KERNEL=$v timeout 200 python scripts/screen_time.py 2>&1 | tail -1 | tee -a $O/screen_time.txt
done
AMDKGE_SCREEN_KERNEL=4 timeout 900 python -m pytest tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "screen or bit_identical" > $O/pytest_screen_r.log 2>&1; echo "rc=$?" >> $O/pytest_screen_r.log; tail -4 $O/pytest_screen_r.log | cut -c1-400
cd /tmp
for v in 1 4; do
  P=$R/$O/trace_v$v; rm -rf $P; mkdir -p $P
  AMDKGE_SCREEN_KERNEL=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > /dev/null 2> $P/err.log || tail -3 $P/err.log
  python - "$v" <<PY
import csv, glob, sys
for g in glob.glob("$R/$O/trace_v"+sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(g)))
    print("kernel", sys.argv[1])
    for r in rows[:12]: print("  %-70s calls %5s avg_us %9.1f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("$R/$O/kernel_stats_v"+sys.argv[1]+".csv", "w").write(open(g).read())
PY
  find $P -name "*.csv" -size +1M -delete
done
for v in 1 4; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
    i=$((i+1)); P=$R/$O/pmc_screen_v$v/p$i; rm -rf $P; mkdir -p $P
    AMDKGE_SCREEN_KERNEL=$v timeout 150 rocprofv3 --pmc $set --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > /dev/null 2> $P/err.log || tail -3 $P/err.log
  done
  python - "$v" <<PY
import csv, glob, collections, json, sys
acc = collections.defaultdict(list)
for g in glob.glob("$R/$O/pmc_screen_v"+sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in acc.items()}
if m:
    m["mfma_util"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024) if m.get("GRBM_GUI_ACTIVE") else None
    print("screen kernel", sys.argv[1], json.dumps({k: round(v, 4) if k == "mfma_util" and v else round(v) for k, v in sorted(m.items())}))
    json.dump({"kernel_variant": sys.argv[1], "mean_per_launch": m}, open("$R/$O/pmc_screen_v"+sys.argv[1]+".json", "w"), indent=1)
PY
  find $R/$O/pmc_screen_v$v -name "*.csv" -size +1M -delete
done
cd $R
