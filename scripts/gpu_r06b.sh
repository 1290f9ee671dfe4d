#!/bin/bash
# round 6, second lease: (1) the tests added since the first lease (threaded group rank, columns + EarlyStopping, staging), (2) the LDS-DMA
# screening kernel (AMDKGE_SCREEN_KERNEL=3) against the shipped one: identity tests, timing, SQ counters, (3) old (round 5) vs new library
# on the train step (cost of the NaN-faithful loss code), (4) C4's tile pass: counters + phase split with the ablation build,
# (5) the driver's default command with the new `dropin` / C5 objects.
set -u
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_nonfinite.py -q -p no:cacheprovider > $O/pytest_nonfinite.log 2>&1; echo "rc=$?" >> $O/pytest_nonfinite.log; grep -E "^FAILED| passed| failed" $O/pytest_nonfinite.log | cut -c1-200 | tail -40
timeout 900 python -m pytest tests/test_gpu_session.py tests/test_gpu_cols.py tests/test_gpu_deterministic.py tests/test_gpu_kernels.py -q -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log; tail -6 $O/pytest_new.log | cut -c1-400
# ---- (2) screening kernel variants
for v in 1 3; do
  AMDKGE_SCREEN_KERNEL=$v timeout 200 python scripts/screen_time.py 2>&1 | tail -1 | tee -a $O/screen_time.txt
done
AMDKGE_SCREEN_KERNEL=3 timeout 900 python -m pytest tests/test_gpu_rank_screen.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "screen or bit_identical" > $O/pytest_screen_v3.log 2>&1; echo "rc=$?" >> $O/pytest_screen_v3.log; tail -4 $O/pytest_screen_v3.log | cut -c1-400
for v in 1 3; do
  for cfg in "" "--config C3"; do
    AMDKGE_SCREEN_KERNEL=$v timeout 300 python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --also none 2>> $O/bench.err | grep '^{' | tail -1 > $O/b.json
    python - "$v" "$cfg" <<PY
import json,sys
d=json.load(open("$O/b.json")); e=d["eval"]; d["screen_kernel"]=sys.argv[1]
print("screen kernel", sys.argv[1], sys.argv[2] or "C2", "eval ms", round(e["ms"],3), "ranks/s", round(e["ranks_per_s"]/1e6,2), "M  identical to exact:", e["exact_fp32_kernel_alone"]["ranks_identical_to_screened"], " train ms", round(d["ms_per_step"],4))
open("$O/eval_lines.jsonl","a").write(json.dumps(d)+"\n")
PY
  done
done
cd /tmp
for v in 1 3; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
    i=$((i+1)); P=$R/$O/pmc_screen_v$v/p$i; rm -rf $P; mkdir -p $P
    AMDKGE_SCREEN_KERNEL=$v timeout 150 rocprofv3 --pmc $set --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > /dev/null 2> $P/err.log || tail -3 $P/err.log
  done
  python - "$v" <<PY
import csv, glob, collections, json, sys
acc = collections.defaultdict(list)
for g in glob.glob("$R/$O/pmc_screen_v"+sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in acc.items()}
if m:
    n_mfma = 22671360.0
    m["mfma_util"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024) if m.get("GRBM_GUI_ACTIVE") else None
    print("screen kernel", sys.argv[1], json.dumps({k: round(v, 4) if k == "mfma_util" and v else round(v) for k, v in sorted(m.items())}))
    json.dump({"kernel_variant": sys.argv[1], "mean_per_launch": m}, open("$R/$O/pmc_screen_v"+sys.argv[1]+".json", "w"), indent=1)
PY
  find $R/$O/pmc_screen_v$v -name "*.csv" -size +2M -delete
done
cd $R
# ---- (3) round-5 library vs this one on the train step
for run in "old:" "new:" "old:--model TransE" "new:--model TransE" "old:--model RotatE" "new:--model RotatE" "old:--model DistMult" "new:--model DistMult" "old:--config C4" "new:--config C4" "old:--config C1" "new:--config C1"; do
  lib=${run%%:*}; cfg=${run#*:}
  if [ $lib = old ]; then export AMDKGE_LIB=$R/build_variants/r05/libamdkge.so; else unset AMDKGE_LIB; fi
  timeout 120 python bench.py $cfg --no-cpu-baseline --no-eval --also none --reps 3 2>> $O/bench.err | grep '^{' | tail -1 > $O/b.json
  python - "$lib" "$cfg" <<PY
import json,sys
d=json.load(open("$O/b.json")); d["lib"]=sys.argv[1]; d["flags"]=sys.argv[2]
print(sys.argv[1], sys.argv[2] or "C2", "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],3))
open("$O/train_old_vs_new.jsonl","a").write(json.dumps(d)+"\n")
PY
done
unset AMDKGE_LIB
# ---- (4) C4's tile pass: phase split (ablation build: 1024 no flush, 4096 no bucket scan, 2048 no accumulator zeroing) and counters
export AMDKGE_LIB=$R/build_variants/ablate/libamdkge.so
for dbg in 0 1024 4096 5120 7168; do
  AMDKGE_DEBUG=$dbg timeout 120 python bench.py --config C4 --no-cpu-baseline --no-eval --also none --steps 100 --warmup 10 --reps 3 2>> $O/bench.err | grep '^{' | tail -1 > $O/b.json
  python - "$dbg" <<PY
import json,sys
d=json.load(open("$O/b.json")); print("C4 ablate dbg", sys.argv[1], "ms/step", round(d["ms_per_step"],4))
open("$O/c4_ablate.jsonl","a").write(json.dumps({"dbg": int(sys.argv[1]), "ms_per_step": d["ms_per_step"], "phases_ms": d["phases_ms"]})+"\n")
PY
done
unset AMDKGE_LIB
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "(TCP|TCC|TA|TD|SQ)_[A-Z0-9_]+" | sort -u > $R/$O/counter_names.txt; wc -l $R/$O/counter_names.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); P=$R/$O/pmc_c4/p$i; rm -rf $P; mkdir -p $P
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $P -o r -- python $R/bench.py --config C4 --no-cpu-baseline --no-eval --also none --steps 6 --warmup 2 --reps 1 --phase-steps 2 > /dev/null 2> $P/err.log || (echo "pass $i failed: $set"; tail -2 $P/err.log)
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for g in glob.glob("$R/$O/pmc_c4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        k = r["Kernel_Name"].split("(")[0]
        if "tile_backward" in k or "train_fwdbwd" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v)/len(v) for c, v in sorted(d.items())} for k, d in acc.items()}
json.dump(out, open("$R/$O/c4_tile_counters.json", "w"), indent=1)
for k, d in out.items(): print(k[:70], json.dumps({c: round(v) for c, v in d.items()}))
PY
find $R/$O/pmc_c4 -name "*.csv" -size +2M -delete
cd $R
# ---- (5) the driver's command
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; tail -4 $O/bench_driver_flags.err
python - <<PY
import json
d=json.load(open("$O/bench_driver_flags.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"])
print("dropin", json.dumps(d.get("dropin"))[:1500])
x=d.get("extra_configs",{})
print({k:(v.get("wall_s"), v.get("ms_per_step")) for k,v in x.items()})
print("C5", json.dumps(x.get("C5_one_gpu_shard"))[:1800])
PY
