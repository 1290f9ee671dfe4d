#!/bin/bash
# round 6: kernel r as the default screening kernel of 13-slab rows: timing both ways, the whole GPU suite, the driver's bench line
set -u
O=gpurun_out/${1:-r06k}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in 1 4; do
  AMDKGE_SCREEN_KERNEL=$v timeout 200 python scripts/screen_time.py 2>&1 | tail -1 | tee -a $O/screen_time.txt
done
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log | cut -c1-300
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; tail -4 $O/bench_driver_flags.err
python - <<PY
import json
d=json.load(open("$O/bench_driver_flags.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"])
e=d["eval"]; print("eval ms", e["ms"], "ranks/s", e["ranks_per_s"], "identical", e["exact_fp32_kernel_alone"]["ranks_identical_to_screened"])
x=d.get("extra_configs",{})
for k,v in x.items():
    ee=v.get("eval") or {}
    print(k, v.get("ms_per_step"), ee.get("ms"), ee.get("ranks_per_s"))
print("dropin", json.dumps(d.get("dropin"))[:600])
PY
