#!/bin/bash
# round-2 GPU check B: full GPU suite (new: bit-identical ranks, learning parity, lazy mode, shard kernels), smoke,
# C2 headline + rocprof stats + PMC passes, C5 shard dense vs lazy, 2-rank (gloo, one GPU) runs of the C4 / cut-down C5 presets
set -u
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json
bash scripts/profile_bench.sh r02b > $O/profile.log 2>&1
for mode in dense lazy; do
  timeout 900 python bench.py --config C5 --optimizer-mode $mode --steps 20 --warmup 3 >> $O/bench_c5_1gpu.jsonl 2>> $O/bench_c5.err
done
timeout 600 python bench.py --config C4 --steps 100 --warmup 10 --no-cpu-baseline >> $O/bench_c4_1gpu.jsonl 2>> $O/bench_c4.err
timeout 600 python bench.py --config C4 --optimizer-mode lazy --steps 100 --warmup 10 --no-cpu-baseline >> $O/bench_c4_1gpu.jsonl 2>> $O/bench_c4.err
export AMDKGE_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config C4 --steps 20 --warmup 3 > $O/bench_c4_gloo2.json 2> $O/bench_c4_gloo2.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --config C5 --ents-per-gpu 1000000 --batch 16384 --steps 10 --warmup 2 > $O/bench_c5_gloo2.json 2> $O/bench_c5_gloo2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_c2_gloo2.json 2> $O/bench_c2_gloo2.err
unset AMDKGE_BENCH_BACKEND
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02b/bench*.json*')):
    for line in open(f):
        try: d=json.loads(line)
        except Exception: continue
        print(f.split('/')[-1], d['metric'][-45:], d['config'].get('optimizer_mode'), 'n', d['n_gpus'], 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), d.get('phases_ms'), 'eval', d.get('eval',{}).get('ranks_per_s'))
PY
tail -3 $O/*.err | tail -40
