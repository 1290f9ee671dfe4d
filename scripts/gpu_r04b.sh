#!/bin/bash
# round 4, second GPU call: the early exit of the distance models (tests + bench lines on untrained and trained-like tables, setting
# sweep), the un-skipped RotatE optimizer-rule cases, the many-seed MRR means against the regenerated golden, TransE / pairwise
# bit for bit against the ordered oracle.      usage: scripts/gpu_r04b.sh TAG
set -u
TAG=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_rank_early.py tests/test_gpu_tile_direct.py tests/test_gpu_learning.py tests/test_gpu_fullsize.py tests/test_gpu_session.py -m gpu -q -s --durations=8 > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
grep -h "mean MRR over seeds\|vs ordered oracle\|^early exit\|passed\|failed\|FAILED\|rc=" $O/pytest_new.log | cut -c1-600 | head -80
grep -h -B2 -A12 "Error\|assert " $O/pytest_new.log | head -150
for m in TransE RotatE; do
  timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --steps 50 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
  for cfg in "1,2,1,6" "1,8,4,6" "1,4,2,3" "1,4,2,12"; do
    AMDKGE_RANK_EARLY=$cfg timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --steps 20 --warmup 5 >> $O/dist_models_sweep.jsonl 2>> $O/dist_models.err
  done
done
timeout 300 python bench.py --model TransE --k 50 --eta 5 --loss pairwise --no-cpu-baseline --trained-eval --steps 50 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
timeout 300 python bench.py --model RotatE --k 350 --no-cpu-baseline --trained-eval --steps 20 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
# narrow rows: does a row that is a whole number of 128-byte lines (k = 64) beat k = 50 (208-byte rows) on the atomic path?
for kk in 48 50 64; do timeout 200 python bench.py --model TransE --k $kk --eta 5 --loss pairwise --no-cpu-baseline --no-eval --steps 200 --warmup 20 >> $O/narrow.jsonl 2>> $O/dist_models.err; done
python - <<PY2
import json
for line in open("$O/narrow.jsonl"):
    d = json.loads(line); print("narrow", d["config"]["workload"][26:60], "ms/step", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3))
PY2
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/dist_models*.jsonl")):
    for line in open(f):
        try: d = json.loads(line)
        except Exception: continue
        for key in ("eval", "eval_trained_like"):
            ev = d.get(key) or {}
            ex = ev.get("exact_fp32_kernel_alone") or {}
            print(f.split("/")[-1], d["config"]["workload"][26:58], key, "ranks/s", round(ev.get("ranks_per_s", 0)), "ms", round(ev.get("ms", 0), 3), "| plain ms", round(ex.get("ms", 0), 3), "identical", ex.get("ranks_identical_to_screened"),
                  "| handed over", (ev.get("screening") or {}).get("fraction"), "mrr", ev.get("mrr", ev.get("mrr_untrained_tables")))
PY
grep -v "amdgpu.ids" $O/dist_models.err | tail -5
