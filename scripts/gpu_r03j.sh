#!/bin/bash
set -u
O=$PWD/gpurun_out/r03j; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 4 --warmup 2 --reps 1 --no-cpu-baseline > $O/bench.json 2> $O/err.log
python - <<PY
import csv,glob
f=glob.glob("$O/stats/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'rank' in r['Name'] or 'filter' in r['Name'] or 'DeviceRadix' in r['Name'] or 'scan' in r['Name'].lower():
        print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
