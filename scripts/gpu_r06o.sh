#!/bin/bash
set -u
O=gpurun_out/${1:-r06o}; mkdir -p $O
for combo in "1:2:64" "4:2:64" "4:1:64" "4:2:12" "4:2:8" "4:1:12" "1:1:64"; do
  v=${combo%%:*}; rest=${combo#*:}; lanes=${rest%%:*}; run=${rest#*:}
  AMDKGE_SCREEN_RUN=$run AMDKGE_EVAL_LANES=$lanes AMDKGE_SCREEN_KERNEL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --also none 2>> $O/bench.err | grep '^{' | tail -1 > $O/b.json
  python - "$v" "$lanes" "$run" <<PY
import json,sys
d=json.load(open("$O/b.json")); e=d["eval"]
line="screen kernel %s lanes %s run<=%s C2 eval ms %.3f (means %s) ranks/s %.2f M identical: %s" % (sys.argv[1], sys.argv[2], sys.argv[3], e["ms"], [round(x,3) for x in e.get("ms_mean_before_and_after_the_exact_path")], e["ranks_per_s"]/1e6, e["exact_fp32_kernel_alone"]["ranks_identical_to_screened"])
print(line); open("$O/eval_lines.txt","a").write(line+"\n")
PY
done
