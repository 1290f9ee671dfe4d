#!/bin/bash
# development: SQ counters of the train kernels for a bench shape ("$@" = bench flags)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/a -o r -- python $ROOT/bench.py --no-cpu-baseline --no-eval --steps 10 --warmup 2 "$@" > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM --output-format csv -d $O/b -o r -- python $ROOT/bench.py --no-cpu-baseline --no-eval --steps 10 --warmup 2 "$@" > /dev/null 2> $O/b.err
python - <<PY
import csv,glob,collections
for sub in ("a","b"):
    f=glob.glob("$O/"+sub+"/**/*counter_collection.csv",recursive=True)
    if not f: print("no csv",sub); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0]
        if "train_fwdbwd" in k or "tile_backward" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k[:60], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
