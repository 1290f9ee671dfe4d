#!/bin/bash
set -u
O=$PWD/gpurun_out/r03A; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_session.py tests/test_gpu_discovery.py tests/test_golden.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|^E  |FAILED" $O/pytest.log | tail -12
timeout 300 python bench.py --steps 56 --warmup 10 --no-cpu-baseline 2>>$O/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); e=d['eval']; print('C2-56', round(d['ms_per_step'],4), e['ranks_per_s'], e['ms'], e['screening']['rechecked_pairs_per_side'], e['exact_fp32_kernel_alone'])"
timeout 300 python bench.py --no-cpu-baseline 2>>$O/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); e=d['eval']; print('C2 default', round(d['ms_per_step'],4), e['ranks_per_s'], e['ms'], e['screening']['rechecked_pairs_per_side'], e['exact_fp32_kernel_alone'])"
python scripts/eval_profile.py 2>&1 | grep "evaluate() call"
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $OLDPWD/bench.py --steps 56 --warmup 10 --reps 1 --phase-steps 1 --no-cpu-baseline > /dev/null 2>&1; cd $OLDPWD
python - <<PY
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(s in r["Name"] for s in ("rank_", "filter_")): print(r["Name"].split("(")[0][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["MinNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1))
PY
