#!/bin/bash
# development: issue-level SQ / SQC counters of the screening kernel for one library.  usage: gpu_screen_pmc2.sh <variant|base> <tag>
set -u
V=${1:-base}; TAG=${2:-pmc2}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
if [ $V != base ]; then export AMDKGE_LIB=$R/build_variants/$V/libamdkge.so; fi
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_I8 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU2"; do
  i=$((i+1)); P=/tmp/pmc2_$V/p$i; rm -rf $P; mkdir -p $P
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > /dev/null 2> $P/err.log || tail -3 $P/err.log
done
python - "$V" <<PY
import csv, glob, collections, json, sys
acc = collections.defaultdict(list)
for g in glob.glob("/tmp/pmc2_"+sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel_r" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in sorted(acc.items())}
print(json.dumps({k: round(v) for k, v in m.items()}))
json.dump({"library": sys.argv[1], "mean_per_launch": m}, open("$R/$O/pmc2_"+sys.argv[1]+".json", "w"), indent=1)
PY
cd $R
