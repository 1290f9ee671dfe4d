#!/bin/bash
# per-kernel durations of the train step of the other models at the C2 shape (rocprofv3 kernel trace)
R=$PWD; export TMPDIR=/tmp; cd /tmp
for m in DistMult TransE RotatE; do
  O=$R/gpurun_out/model_$m; rm -rf $O; mkdir -p $O
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python $R/bench.py --model $m --steps 140 --warmup 14 --reps 1 --phase-steps 1 --no-eval --no-cpu-baseline > $O/bench.json 2>/dev/null
  python - <<PY
import csv, glob, json
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
d = json.load(open("$O/bench.json"))
print("$m", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3))
for r in list(csv.DictReader(open(f)))[:4]: print("   ", r["Name"].split("(")[0][:70], r["Calls"], round(float(r["AverageNs"])/1e3, 1), r["Percentage"])
PY
done
