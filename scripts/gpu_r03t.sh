#!/bin/bash
# kernel breakdown of evaluate() with the screening pass (rocprofv3 kernel trace) + the screen kernel's SQ counters
set -u
R=$PWD; O=$R/gpurun_out/r03t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 56 --warmup 10 --reps 1 --phase-steps 1 --no-cpu-baseline > $O/bench.json 2> $O/stats.err
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o r -- python $R/bench.py --steps 56 --warmup 10 --reps 1 --phase-steps 1 --no-cpu-baseline > /dev/null 2> $O/mfma.err
python - <<PY
import csv, glob, collections
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "rank" in n or "filter" in n or float(r["Percentage"]) > 2:
        print(f'{n.split("(")[0][:80]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
g = glob.glob("$O/mfma/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(g)):
    k = r["Kernel_Name"].split("(")[0]
    if "rank" in k: acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c, x in v.items()})
PY
