#!/bin/bash
# round 4: screening kernel at three waves per SIMD (launch bounds only: 168 VGPRs + 208 B of scratch) against the default; early-exit overhead check
set -u
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $ROOT
for lib in default screen3w; do
  if [ "$lib" != default ]; then export AMDKGE_LIB=$ROOT/build_variants/$lib/libamdkge.so; else unset AMDKGE_LIB; fi
  for flags in "" "--config C3" "--model DistMult" "--k 350"; do
    timeout 300 python bench.py $flags --no-cpu-baseline --also none --steps 20 --warmup 5 > $O/tmp.json 2>> $O/err.log
    python - "$lib" "$flags" <<PY
import json, sys
d = json.load(open("$O/tmp.json")); ev = d["eval"]; ex = ev.get("exact_fp32_kernel_alone") or {}
print("screen", sys.argv[1], "|", sys.argv[2], "| eval ms", round(ev["ms"], 3), "ranks/s", round(ev["ranks_per_s"]), "identical", ex.get("ranks_identical_to_screened"), "trained-like ms", round((d.get("eval_trained_like") or {}).get("ms", 0), 3))
open("$O/screen_ab.jsonl", "a").write(json.dumps({"lib": sys.argv[1], "flags": sys.argv[2], "eval_ms": ev["ms"], "ranks_per_s": ev["ranks_per_s"], "identical": ex.get("ranks_identical_to_screened")}) + "\n")
PY
  done
done
unset AMDKGE_LIB
for m in TransE RotatE; do timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --steps 20 --warmup 5 >> $O/dist_models.jsonl 2>> $O/err.log; done
python - <<PY
import json
for line in open("$O/dist_models.jsonl"):
    d = json.loads(line)
    for key in ("eval", "eval_trained_like"):
        ev = d.get(key) or {}; ex = ev.get("exact_fp32_kernel_alone") or {}
        print(d["config"]["workload"][26:54], key[:12], "ranks/s", round(ev.get("ranks_per_s", 0)), "ms", round(ev.get("ms", 0), 3), "| plain ms", round(ex.get("ms", 0), 3), "same", ex.get("ranks_identical_to_screened"))
PY
grep -v amdgpu.ids $O/err.log | tail -3
