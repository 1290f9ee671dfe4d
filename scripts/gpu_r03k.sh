#!/bin/bash
set -u
O=$PWD/gpurun_out/r03k; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rank_screen.py -m gpu -q -x -k nothing > $O/pytest_screen.log 2>&1; echo "rc=$?" >> $O/pytest_screen.log; grep -E "passed|failed|Error|assert" $O/pytest_screen.log | tail -10
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 56 --warmup 10 --reps 1 --no-cpu-baseline > $O/bench.json 2> $O/err.log
python - <<PY
import csv,glob,json
f=glob.glob("$O/stats/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'rank' in r['Name']:
        print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
print(json.load(open("$O/bench.json"))['eval'])
PY
import sys,json; d=json.loads(sys.stdin.readline()); print('C3 eval', d['eval'])"
