#!/bin/bash
set -u
O=gpurun_out/${1:-r06r}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py tests/test_gpu_session.py -q -p no:cacheprovider -k "screen or bit_identical or rank" > $O/pytest_screen.log 2>&1; echo "rc=$?" >> $O/pytest_screen.log; tail -3 $O/pytest_screen.log | cut -c1-300
AMDKGE_SCREEN_KERNEL=4 timeout 200 python scripts/screen_time.py 2>&1 | tail -1 | tee -a $O/screen_time.txt
for cfg in "" "--config C3"; do
  timeout 300 python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --also none 2>> $O/bench.err | grep '^{' | tail -1 > $O/b.json
  python - "$cfg" <<PY
import json,sys
d=json.load(open("$O/b.json")); e=d["eval"]
line="default %-12s eval ms %.3f (means %s) ranks/s %.2f M identical: %s" % (sys.argv[1] or "C2", e["ms"], [round(x,3) for x in e.get("ms_mean_before_and_after_the_exact_path")], e["ranks_per_s"]/1e6, e["exact_fp32_kernel_alone"]["ranks_identical_to_screened"])
print(line); open("$O/eval_lines.txt","a").write(line+"\n")
PY
done
bash scripts/gpu_screen_quick.sh 4 ${1:-r06r} 2>&1 | tail -1
