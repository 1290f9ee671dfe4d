#!/bin/bash
# kernel trace of one model's evaluate() lines (untrained + trained-like tables).   usage: scripts/gpu_eval_trace.sh TAG MODEL
TAG=${1:-evt}; M=${2:-TransE}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$M -o r -- python $GRAFT_REPO_ROOT/bench.py --model $M --no-cpu-baseline --trained-eval --also none --steps 20 --warmup 5 --reps 1 > $O/bench_$M.json 2> $O/err_$M.log
f=$(find $O/trace_$M -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last evaluation = the trained-like one: print the kernels of its last two sides
idx = [i for i, r in enumerate(rows) if "rank_early_merge" in r["Kernel_Name"] or "rank_screen_merge" in r["Kernel_Name"]]
if idx:
    i0 = max(0, idx[-2] - 40 if len(idx) > 1 else idx[-1] - 40)
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:idx[-1] + 6]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:100]}")
PY
rm -rf $O/trace_$M
