#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command plus the two PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md counters table) on a short run.
# Outputs under gpurun_out/prof_$1/ ; scripts/summarise_profiles.py turns them into profiles/*.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $ROOT/bench.py --no-cpu-baseline --also none > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eval --also none > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eval --also none > /dev/null 2> $OUT/write.err
# MFMA utilisation of the 1-vs-all rank kernel (SQ counters in their own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o r -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also none > /dev/null 2> $OUT/mfma.err
ls -R $OUT | head -40
