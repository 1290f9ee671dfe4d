#!/bin/bash
# development: the filler probe's modes under the issue-level counters (calibration of SQ_VALU_MFMA_COEXEC_CYCLES / SQ_WAIT_INST_ANY)
set -u
O=gpurun_out/${1:-probe_pmc}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
hipcc --offload-arch=gfx950 -O3 $R/scripts/mfma_filler_probe.hip -o /tmp/probe || exit 1
cd /tmp
rm -rf /tmp/pp; mkdir -p /tmp/pp
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pp -o r -- /tmp/probe > $R/$O/probe_out.txt 2> /tmp/pp/err.log || tail -3 /tmp/pp/err.log
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(dict)
for g in glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        k = r["Kernel_Name"]; 
        acc[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = open("$R/$O/probe_pmc.txt", "w")
for k in sorted(acc, key=lambda s: int(s.split("<")[1].split(">")[0]) if "<" in s else 0):
    m = {c: max(v) for c, v in acc[k].items()}   # (the long launch of the two)
    line = "%-12s coexec/mfma_busy %.3f  wait_inst/wave %.3f  wait_any/wave %.3f  active_any/wave %.3f  valu_insts %d" % (k[:12], m["SQ_VALU_MFMA_COEXEC_CYCLES"] / max(1, m["SQ_VALU_MFMA_BUSY_CYCLES"]), m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"])
    print(line); out.write(line + "\n")
PY
cd $R
