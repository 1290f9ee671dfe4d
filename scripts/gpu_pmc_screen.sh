#!/bin/bash
# development: SQ counters of the screening kernel (scripts/screen_time.py), several passes
R=$PWD; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1)); O=$R/gpurun_out/pmc_screen/p$i; rm -rf $O; mkdir -p $O
  timeout 100 rocprofv3 --pmc $set --output-format csv -d $O -o r -- python $R/scripts/screen_time.py > /dev/null 2> $O/err.log || tail -3 $O/err.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for g in glob.glob("$R/gpurun_out/pmc_screen/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:28s} {sum(acc[k])/len(acc[k]):16.0f}")
PY
