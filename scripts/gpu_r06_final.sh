#!/bin/bash
# round 6, the build after the closing run (variants g / p out of the library, C1 in the default line): the whole GPU suite, smoke, the driver's command
set -u
TAG=${1:-r06zz}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  AMDKGE_MARGIN_LOG=$PWD/$O/margins_run$i.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest_run$i.log 2>&1; echo "pytest rc=$?" >> $O/pytest_run$i.log
  grep -E "^FAILED| passed| failed|rc=" $O/pytest_run$i.log | tail -6
done
python scripts/margin_summary.py $O/margins_run1.jsonl $O/margins_run2.jsonl > $O/margins_summary.json 2> $O/margins_low.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench.err; tail -4 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench_driver_flags.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), (d["roofline"].get("traffic_source") or "")[:50])
e=d["eval"]; print("eval ms", e["ms"], "ranks/s", e["ranks_per_s"], "identical", e["exact_fp32_kernel_alone"]["ranks_identical_to_screened"], "util", e["roofline"]["int8_mfma_util"])
et=d.get("eval_trained_like") or {}; print("trained-like", et.get("ms"), et.get("ranks_per_s"))
for k,v in d.get("extra_configs",{}).items():
    ee=v.get("eval") or {}
    print(k, v.get("ms_per_step"), ee.get("ms"), ee.get("ranks_per_s"), v.get("wall_s"), v.get("error"))
print("dropin", {k: d["dropin"].get(k) for k in ("fit_epoch_ms","fit_epoch_over_28_steps","evaluate_call_ms","evaluate_call_cached_filter_ms")})
PY
