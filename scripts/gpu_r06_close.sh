#!/bin/bash
# round-6 closing run on the FINAL build (as in round 5): the whole GPU suite THREE times (no -x, no cache), all logs kept; smoke;
# the driver's bench command; rocprofv3 stats + PMC passes of the headline; the models / configs / deterministic / column-sharded lines.
# usage: scripts/gpu_r05_close.sh TAG
set -u
TAG=${1:-r06z}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  AMDKGE_MARGIN_LOG=$PWD/$O/margins_run$i.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest_run$i.log 2>&1; echo "pytest rc=$?" >> $O/pytest_run$i.log
  grep -E "^FAILED| passed| failed|rc=" $O/pytest_run$i.log | tail -6
done
python scripts/margin_summary.py $O/margins_run1.jsonl $O/margins_run2.jsonl $O/margins_run3.jsonl > $O/margins_summary.json 2> $O/margins_low.json; cat $O/margins_low.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench.err
timeout 600 python bench.py > $O/bench_c2.json 2>> $O/bench.err
bash scripts/profile_bench.sh $TAG > $O/profile.log 2>&1
for m in ComplEx DistMult HolE TransE RotatE; do timeout 300 python bench.py --model $m --no-cpu-baseline --trained-eval --also none >> $O/models.jsonl 2>> $O/models.err; done
for c in C1 C3 C4; do timeout 400 python bench.py --config $c --no-cpu-baseline --trained-eval --also none >> $O/configs.jsonl 2>> $O/configs.err; done
timeout 300 python bench.py --popularity zipf --no-cpu-baseline --also none >> $O/zipf.jsonl 2>> $O/configs.err
timeout 300 python bench.py --deterministic --no-cpu-baseline --no-eval --also none >> $O/det.jsonl 2>> $O/configs.err
timeout 300 python bench.py --deterministic --model TransE --no-cpu-baseline --no-eval --also none >> $O/det.jsonl 2>> $O/configs.err
for w in 8 4 2; do AMDKGE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --parallelism columns --cols-of $w --no-cpu-baseline --no-eval --also none >> $O/cols.jsonl 2>> $O/configs.err; done
R=$PWD
( cd /tmp; AMDKGE_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cols8_stats -o r -- python $R/bench.py --parallelism columns --cols-of 8 --no-cpu-baseline --no-eval --also none --steps 100 --warmup 10 > $R/$O/cols8_under_rocprof.json 2> $R/$O/cols8_stats.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$O/cols8_l2 -o r -- python $R/bench.py --parallelism columns --cols-of 8 --steps 10 --warmup 2 --no-cpu-baseline --no-eval --also none > /dev/null 2> $R/$O/cols8_l2.err )
python - <<PY
import csv,glob,collections,json
f=glob.glob("$O/cols8_stats/**/*kernel_stats.csv",recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:6]: print("  cols8", r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
g=glob.glob("$O/cols8_l2/**/*counter_collection.csv",recursive=True)
if g:
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(g[0])):
        k=r["Kernel_Name"].split("(")[0]
        if "cols_" in k or "tile_backward" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out={}
    for k,v in acc.items():
        h=sum(v.get("TCC_HIT_sum",[0]))/max(1,len(v.get("TCC_HIT_sum",[0]))); m=sum(v.get("TCC_MISS_sum",[0]))/max(1,len(v.get("TCC_MISS_sum",[0])))
        out[k]={"TCC_HIT_sum_per_launch":h,"TCC_MISS_sum_per_launch":m,"l2_hit_rate":h/(h+m) if h+m else None}
    json.dump({"command":"rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -- bench.py --parallelism columns --cols-of 8 (one rank of 8: 52-unit slices of 14 505 rows, 80 000 positives per step)","kernels":out},open("$O/cols8_l2_hit_rates.json","w"),indent=1)
    print(json.dumps(out,indent=1))
PY
AMDKGE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-eval --also none >> $O/rccl_world1.jsonl 2>> $O/configs.err
AMDKGE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-eval --also none >> $O/gloo2.jsonl 2>> $O/gloo2.err
AMDKGE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --parallelism columns --steps 20 --warmup 3 --no-cpu-baseline --no-eval --also none >> $O/gloo2.jsonl 2>> $O/gloo2.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        if isinstance(d,dict) and 'ms_per_step' in d:
            ev=d.get('eval') or {}
            et=d.get('eval_trained_like') or {}
            print(f.split('/')[-1], d['config']['workload'][:44], '|', d['config']['parallelism'][:14], 'n', d['n_gpus'], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'eval', round(ev.get('ranks_per_s',0)), 'ms', round(ev.get('ms',0),3), 'exact', round(((ev.get('exact_fp32_kernel_alone') or {}).get('ms') or 0),3),
                  '| trained-like', round(et.get('ranks_per_s',0)), 'ms', round(et.get('ms',0),3), 'plain', round(((et.get('exact_fp32_kernel_alone') or {}).get('ms') or 0),3))
PY
grep -l "Error\|Traceback" $O/*.err | head
find $O gpurun_out/prof_$TAG -name "*.csv" -size +3M -delete
