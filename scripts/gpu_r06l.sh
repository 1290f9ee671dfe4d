#!/bin/bash
# round 6: the bench's evaluation line (filtered, both sides) with screening kernel v1 / r and the sides in two lanes / one after the other
set -u
O=gpurun_out/${1:-r06l}; mkdir -p $O
for v in 1 4; do for lanes in 2 1; do for cfg in "" "--config C3"; do
  AMDKGE_EVAL_LANES=$lanes AMDKGE_SCREEN_KERNEL=$v timeout 300 python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --also none 2>> $O/bench.err | grep '^{' | tail -1 > $O/b.json
  python - "$v" "$lanes" "$cfg" <<PY
import json,sys
d=json.load(open("$O/b.json")); e=d["eval"]
line="screen kernel %s lanes %s %-12s eval ms %.3f (means %s) ranks/s %.2f M identical to exact: %s  recheck %s" % (sys.argv[1], sys.argv[2], sys.argv[3] or "C2", e["ms"], e.get("ms_mean_before_and_after_the_exact_path"), e["ranks_per_s"]/1e6, e["exact_fp32_kernel_alone"]["ranks_identical_to_screened"], e.get("screening",{}).get("rechecked_pairs"))
print(line); open("$O/eval_lines.txt","a").write(line+"\n")
PY
done; done; done
