#!/bin/bash
# quick look at one screening kernel variant: serial timing (+ ranks crc32) and the two SQ counter passes.  usage: gpu_screen_quick.sh <variant> <tag>
set -u
V=${1:-4}; TAG=${2:-quick}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
AMDKGE_SCREEN_KERNEL=$V timeout 200 python scripts/screen_time.py 2>&1 | tail -1 | tee -a $O/screen_time.txt
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
  i=$((i+1)); P=$R/$O/pmc_v$V/p$i; rm -rf $P; mkdir -p $P
  AMDKGE_SCREEN_KERNEL=$V timeout 150 rocprofv3 --pmc $set --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > /dev/null 2> $P/err.log || tail -3 $P/err.log
done
python - "$V" <<PY
import csv, glob, collections, json, sys
acc = collections.defaultdict(list)
for g in glob.glob("$R/$O/pmc_v"+sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel_r" in r["Kernel_Name"] or ("rank_screen_kernel_v1(" in r["Kernel_Name"] and sys.argv[1] == "1"): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in acc.items()}
if m:
    m["mfma_util"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024) if m.get("GRBM_GUI_ACTIVE") else None
    print("screen kernel", sys.argv[1], json.dumps({k: round(v, 4) if k == "mfma_util" and v else round(v) for k, v in sorted(m.items())}))
    json.dump({"kernel_variant": sys.argv[1], "mean_per_launch": m}, open("$R/$O/pmc_screen_v"+sys.argv[1]+".json", "w"), indent=1)
PY
find $R/$O -name "*.csv" -size +1M -delete
cd $R
