#!/bin/bash
# development: L2 hit rate / fabric requests of the screening kernel.  usage: gpu_screen_tcc.sh <variant|base> <tag>
set -u
V=${1:-base}; TAG=${2:-tcc}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
if [ $V != base ]; then export AMDKGE_LIB=$R/build_variants/$V/libamdkge.so; fi
cd /tmp
P=/tmp/tcc_$V; rm -rf $P; mkdir -p $P
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $P -o r -- python $R/scripts/screen_time.py > /dev/null 2> $P/err.log || tail -3 $P/err.log
python - "$V" <<PY
import csv, glob, collections, json, sys
acc = collections.defaultdict(list)
for g in glob.glob("/tmp/tcc_"+sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        if "rank_screen_kernel_r" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in sorted(acc.items())}
if m.get("TCC_HIT_sum") is not None: m["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
print(json.dumps({k: (round(v, 4) if k == "l2_hit_rate" else round(v)) for k, v in m.items()}))
json.dump({"library": sys.argv[1], "mean_per_launch": m}, open("$R/$O/tcc_"+sys.argv[1]+".json", "w"), indent=1)
PY
cd $R
