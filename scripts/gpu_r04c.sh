#!/bin/bash
# round 4, third GPU call: early exit with the device-side probe (tests, bench on untrained / planted-fitted tables, settings), the row-
# sharded session group, the re-barred RotatE rule cases + their lazy-mode diagnostic, the pipelined tile flush against the serial
# build (C2 / C3 / C4 / zipf / k = 350 / one GPU's C5 shard).      usage: scripts/gpu_r04c.sh TAG
set -u
TAG=${1:-r04c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_rank_early.py tests/test_gpu_session.py tests/test_gpu_tile_direct.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_lazy.py tests/test_gpu_deterministic.py -m gpu -q -s --durations=6 > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
grep -h "^probe \|passed\|failed\|FAILED\|rc=" $O/pytest_new.log | cut -c1-300 | head -60
grep -h -B2 -A14 "Error\b" $O/pytest_new.log | cut -c1-400 | head -120
timeout 200 python -m pytest tests/test_gpu_learning.py -m gpu -q -k "not mean_mrr" > $O/pytest_learning.log 2>&1; tail -3 $O/pytest_learning.log | cut -c1-300
echo "== lazy diag"; timeout 200 python scripts/diag_rotate_rules.py lazy > $O/diag_rotate_rules_lazy.jsonl 2> $O/diag_lazy.err; python - <<PY
import json
for line in open("$O/diag_rotate_rules_lazy.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    if "opt" in d: print(d["opt"], "t", d["t"], "inside", round(d["frac_inside"], 4), "rows_with_bad", d["rows_with_bad"], "by_history", d.get("bad_by_history"), "top", d["bad_rows_top"][:3])
PY
tail -2 $O/diag_lazy.err
echo "== distance models"
for m in TransE RotatE; do
  timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --steps 50 --warmup 5 >> $O/dist_models.jsonl 2>> $O/dist_models.err
  for cfg in "1,2,1,6" "1,8,4,6" "1,4,2,3" "1,4,2,12" "1,2,1,3"; do
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
echo "== flush A/B"
for lib in default flush_serial; do
  if [ "$lib" != default ]; then export AMDKGE_LIB=$ROOT/build_variants/$lib/libamdkge.so; else unset AMDKGE_LIB; fi
  for flags in "" "--config C3" "--config C4" "--config C4 --optimizer-mode lazy" "--popularity zipf" "--k 350" "--model DistMult" "--model TransE" "--model RotatE" "--dataset synth-c5-small --model RotatE --k 1000 --eta 64 --batch 65536"; do
    timeout 300 python bench.py $flags --no-cpu-baseline --no-eval --also none --steps 100 --warmup 10 > $O/flush_tmp.json 2>> $O/flush.err
    python - "$lib" "$flags" <<PY
import json, sys
try:
    d = json.load(open("$O/flush_tmp.json")); print("flush", sys.argv[1], "|", sys.argv[2], "| ms/step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "frac", round(d["roofline"]["frac"], 3))
    open("$O/flush_ab.jsonl", "a").write(json.dumps({"lib": sys.argv[1], "flags": sys.argv[2], "ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms"], "frac": d["roofline"]["frac"]}) + "\n")
except Exception as e: print("flush", sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done
unset AMDKGE_LIB
grep -v "amdgpu.ids" $O/flush.err | tail -5
