#!/bin/bash
# round 4, fifth GPU call: TransE / nll deterministic vs the ordered oracle (diagnostic + test), early exit after the second retune.
set -u
TAG=${1:-r04e}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $ROOT
timeout 200 python scripts/diag_det_nll.py > $O/diag_det_nll.jsonl 2> $O/diag_det_nll.err; cat $O/diag_det_nll.jsonl | cut -c1-600; tail -3 $O/diag_det_nll.err
timeout 400 python -m pytest tests/test_gpu_learning.py tests/test_gpu_deterministic.py tests/test_gpu_rank_early.py tests/test_gpu_kernels.py -m gpu -q -s -k "nll_deterministic or two_runs or early or tiled" --durations=4 > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
grep -h "vs ordered oracle\|passed\|failed\|FAILED\|rc=" $O/pytest_new.log | cut -c1-330 | head -30
grep -h -B2 -A14 "Error\b" $O/pytest_new.log | cut -c1-400 | head -60
for m in TransE RotatE; do timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --steps 50 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err; done
timeout 300 python bench.py --deterministic --no-cpu-baseline --no-eval --also none >> $O/det.jsonl 2>> $O/dist_models.err
timeout 300 python bench.py --deterministic --model TransE --loss nll --no-cpu-baseline --no-eval --also none >> $O/det.jsonl 2>> $O/dist_models.err
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
for line in open("$O/det.jsonl"):
    d = json.loads(line); print("det", d["config"]["workload"][26:70], "ms/step", round(d["ms_per_step"], 4))
PY
grep -v "amdgpu.ids" $O/dist_models.err | tail -5
