#!/bin/bash
# round 4, first GPU call: the new tests (many-seed MRR means, session layer through RCCL / the screening pass), the whole GPU
# suite, first contact with RCCL (forced single-rank process group: C2 replicated with every merge schedule, C4 and a cut-down C5
# row-sharded), 8-rank gloo dry runs, the driver's bench command, L2 hit-rate counters of F and T, the XCD-slice gather probe,
# the RotatE optimizer-rule diagnostic.      usage: scripts/gpu_r04a.sh TAG
set -u
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log | grep -E "FAILED|ERROR|passed|failed|rc=|s call"
grep -h "mean MRR over seeds\|session group of one" $O/pytest.log | head
timeout 200 python -m pytest tests/test_gpu_learning.py tests/test_gpu_session.py -m gpu -q -s -k "mean_mrr or through_rccl or screening_pass" > $O/pytest_new.log 2>&1; grep -h "mean MRR over seeds\|session group of one\|passed\|failed" $O/pytest_new.log | cut -c1-700
echo "== xcd slice probe"; timeout 120 scripts/xcd_slice_bench 2>&1 | tee $O/xcd_slice_bench.txt
echo "== rotate rules"; timeout 300 python scripts/diag_rotate_rules.py > $O/diag_rotate_rules.jsonl 2> $O/diag_rotate_rules.err; cut -c1-420 $O/diag_rotate_rules.jsonl | head -40; tail -3 $O/diag_rotate_rules.err
echo "== RCCL first contact (world 1, forced)"
for extra in "" "--config C4" "--config C4 --parallelism sharded-global" "--config C5 --ents-per-gpu 400000 --batch 8192"; do
  AMDKGE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 $extra --steps 20 --warmup 5 --no-cpu-baseline --no-eval >> $O/rccl_world1.jsonl 2>> $O/rccl_world1.err
done
for m in allreduce sharded; do for gth in alltoall alltoall+allgather native; do
  AMDKGE_DP_MERGE=$m AMDKGE_DP_GATHER=$gth AMDKGE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-eval >> $O/rccl_world1.jsonl 2>> $O/rccl_world1.err
done; done
echo "== gloo 8-rank dry runs"
for extra in "" "--config C4" "--config C5 --ents-per-gpu 100000 --batch 4096"; do
  AMDKGE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 $extra --steps 5 --warmup 2 --reps 2 --no-cpu-baseline --no-eval >> $O/gloo8.jsonl 2>> $O/gloo8.err
done
echo "== driver command"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
echo "== L2 counters"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' | cut -c1-1500 > $O/tcc_counters.txt
for pass in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | tr -c "A-Za-z0-9" "_")
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$tag -o r -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eval --also none > /dev/null 2> $O/pmc_$tag.err
done
cd $ROOT
python - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$O/pmc_*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f:
        print("no csv", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "train_fwdbwd" in k or "tile_backward" in k:
            acc[("F" if "train_fwdbwd" in k else "T")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, x in v.items():
            out.setdefault(k, {})[c] = sum(x) / len(x)
for k, v in out.items():
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
json.dump(out, open("$O/pmc_l2.json", "w"), indent=1)
for f in sorted(glob.glob("$O/*.json*")):
    for line in open(f):
        try: d = json.loads(line)
        except Exception: continue
        if isinstance(d, dict) and "ms_per_step" in d:
            ev = d.get("eval") or {}
            print(f.split("/")[-1], d["config"]["workload"][:44], "| par", d["config"]["parallelism"][:60], "| n", d["n_gpus"], "ms", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3),
                  "ranks", d["ranks"].get("backend"), d["ranks"].get("rccl_version"), d["ranks"].get("merge_schedule"), d["config"].get("merge_ms_per_step_measured"), "eval", round(ev.get("ranks_per_s", 0)))
PY
grep -l "Error\|Traceback" $O/*.err 2>/dev/null | head
