#!/bin/bash
# round 4, fourth GPU call: bitwise fits against the ordered oracle (all update rules, both integer-gradient losses, C1 at full size),
# early exit after the retune (tests + bench), re-barred rule cases.      usage: scripts/gpu_r04d.sh TAG
set -u
TAG=${1:-r04d}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_learning.py tests/test_gpu_fullsize.py tests/test_gpu_rank_early.py tests/test_gpu_tile_direct.py tests/test_gpu_session.py -m gpu -q -s -k "not mean_mrr" --durations=6 > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
grep -h "vs ordered oracle\|passed\|failed\|FAILED\|rc=" $O/pytest_new.log | cut -c1-330 | head -60
grep -h -B2 -A14 "Error\b" $O/pytest_new.log | cut -c1-400 | head -80
echo "== distance models"
for m in TransE RotatE; do
  timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --steps 50 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
  for cfg in "1,1,1,12" "1,2,2,12" "1,2,1,24" "1,4,2,12"; do
    AMDKGE_RANK_EARLY=$cfg timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --steps 10 --warmup 3 >> $O/dist_models_sweep.jsonl 2>> $O/dist_models.err
  done
done
timeout 300 python bench.py --config C1 --no-cpu-baseline --trained-eval --steps 50 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
timeout 300 python bench.py --model RotatE --k 350 --no-cpu-baseline --trained-eval --steps 20 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
timeout 300 python bench.py --model TransE --k 350 --no-cpu-baseline --trained-eval --steps 20 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/dist_models*.jsonl")):
    for line in open(f):
        try: d = json.loads(line)
        except Exception: continue
        for key in ("eval", "eval_trained_like"):
            ev = d.get(key) or {}
            ex = ev.get("exact_fp32_kernel_alone") or {}
            print(f.split("/")[-1][:18], d["config"]["workload"][26:54], key[:12], "ranks/s", round(ev.get("ranks_per_s", 0)), "ms", round(ev.get("ms", 0), 3), "| plain ms", round(ex.get("ms", 0), 3), "same", ex.get("ranks_identical_to_screened"),
                  "| handed over", (ev.get("screening") or {}).get("fraction"), "mrr", round(ev.get("mrr", ev.get("mrr_untrained_tables", 0)), 4))
PY
grep -v "amdgpu.ids" $O/dist_models.err | tail -5
