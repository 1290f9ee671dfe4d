#!/bin/bash
set -u
O=$PWD/gpurun_out/r03l; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/a -o r -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --reps 1 > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/b -o r -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --reps 1 > /dev/null 2> $O/b.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM --output-format csv -d $O/c -o r -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --reps 1 > /dev/null 2> $O/c.err
python - <<PY
import csv,glob,collections
for sub in ("a","b","c"):
    f=glob.glob("$O/"+sub+"/**/*counter_collection.csv",recursive=True)
    if not f: print("no csv",sub, open("$O/"+sub+".err").read()[-400:]); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0]
        if "rank_screen_kernel" in k or "rank_count_mfma_pipe" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k[:40], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
