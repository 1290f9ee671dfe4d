#!/bin/bash
# One GPU's C5 shard (RotatE k=1000 eta=64 B=65536, 6.25 M rows = 50 GB table, 200 GB resident): bench lines, rocprofv3 kernel
# stats and the FETCH_SIZE / WRITE_SIZE counter passes (separate runs, as the guide prescribes), dense and touched-rows mode.
# usage: scripts/gpu_c5_prof.sh TAG [extra bench flags]
set -u
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/c5_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --config C5 --no-cpu-baseline --no-eval"
for mode in lazy dense; do
  timeout 900 $B --optimizer-mode $mode --steps 6 --warmup 2 --reps 3 "$@" >> $O/bench.jsonl 2>> $O/bench.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -o r -- $B --optimizer-mode $mode --steps 4 --warmup 1 --reps 1 --phase-steps 1 "$@" > /dev/null 2> $O/stats_$mode.err
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$mode -o r -- $B --optimizer-mode $mode --steps 2 --warmup 1 --reps 1 --phase-steps 1 "$@" > /dev/null 2> $O/fetch_$mode.err
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$mode -o r -- $B --optimizer-mode $mode --steps 2 --warmup 1 --reps 1 --phase-steps 1 "$@" > /dev/null 2> $O/write_$mode.err
done
python - <<PY
import csv, glob, json, collections
O = "$O"
out = {}
for mode in ("lazy", "dense"):
    res = {}
    f = glob.glob(f"{O}/stats_{mode}/**/*kernel_stats.csv", recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            if float(r["Percentage"]) > 0.5:
                res[r["Name"].split("(")[0][:90]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        g = glob.glob(f"{O}/{sub}_{mode}/**/*counter_collection.csv", recursive=True)
        if not g: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(g[0])):
            if r["Counter_Name"] == ctr: acc[r["Kernel_Name"].split("(")[0][:90]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if k in res: res[k][ctr + "_kb_mean"] = sum(v) / len(v)
    out[mode] = res
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
for mode, res in out.items():
    print(mode)
    for k, v in res.items():
        fb = 2 * 1024 * v.get("FETCH_SIZE_kb_mean", 0); wb = 1024 * v.get("WRITE_SIZE_kb_mean", 0)
        print(f"  {k[:70]:70s} calls {v['calls']:4d} avg {v['avg_us']:10.1f} us  {v['pct']:5.1f}%  fetch(x2) {fb/1e9:8.2f} GB write {wb/1e9:8.2f} GB  -> {(fb+wb)/max(v['avg_us'],1e-9)/1e6:6.2f} TB/s")
PY
tail -3 $O/bench.jsonl | cut -c1-400
